"""GPU parity of the three f16x3 implicit-GEMM kernels behind dpc_conv_pack / dpc_conv_run (include/dpc.h) against an fp64
restatement of the operator, on shapes chosen to land in each kernel and each of its edge paths:

  * row panels (csrc/igemm_panel.hip): 1-tap K <= 512, N in {64, 128, 256, 384}; LayerNorm prologue, residual, virtual concat,
    K % 32 != 0 falls back; few-tap im2col form (stride-2 pixel-unshuffle conv);
  * pixel tiles (csrc/igemm_tile.hip): the stride-1 2 x 2-tap classes of ConvTranspose at 64 -> 64 and 128 -> 128 channels, plain
    and with the parity scatter (out_mode 2), every parity's tap offsets, images narrower than a tile;
  * LDS-tiled wide kernel (csrc/igemm_wide.hip): reductions of >= 24 chunks, 128- and 64-column tiles, split-K by N, K % 32 != 0;
  * image tiles (csrc/igemm_img.hip, r04): the 3 x 3 convolutions of the Burgers net's 4 x 32 / 2 x 16 / 1 x 8 levels;
  * the narrow kernel (csrc/igemm6.hip) for what neither takes (shared vector epilogue, fragment-order weight pack).

Every case has a RAGGED row count (M is not a multiple of 32 / 64 / 256) so the tail rows of the last tile are exercised, and is
run twice: on all images and on a prefix of them -- the rows both runs compute must be BIT-identical (tile shape, slice count and
kernel choice are functions of the operator's shape only, never of the batch: the property the sharded samplers rely on).

The operator (reference: nn.Conv2d / the parity classes of nn.ConvTranspose2d as model/burgers_1d/unet.py and
video_diffusion_pytorch_conv3d.py:159-163, 206-257 use them):
    out[img, i, j, n] = bias[n] + resid + sum_{a, b, c} x[img, i sh + a - ph, j sw + b - pw, c] w[n, c, a, b]      (zero outside)
Tolerance: 3e-6 of the output range, the conv op tests' bound (measured on these cases: <= 7.8e-7, profiles/r03_al_igemm_kernel_tests.log); the x6 mode (exact products, fp32 accumulation) of the same entry points to 2e-6.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def SH():
    from diffphycon_amd.model import surrogates_hip
    return surrogates_hip


def reference(x0, x1, w, bias, resid, ln, sh, sw, ph, pw, Ho, Wo):
    """fp64, channels-last.  x*: [images, H, W, C]; w [N, K, kh, kw]; ln = (stats [rows, 2], gamma [K]) on the 1-tap input."""
    x = x0.double() if x1 is None else torch.cat([x0.double(), x1.double()], -1)
    images, H, W, K = x.shape
    if ln is not None:
        st = ln[0].double().reshape(images, H, W, 2)
        x = (x - st[..., 0:1]) * st[..., 1:2] * ln[1].double()
    N, _, kh, kw = w.shape
    P = 8
    xp = torch.zeros(images, H + 2 * P + sh * Ho, W + 2 * P + sw * Wo, K, dtype=torch.float64)
    xp[:, P:P + H, P:P + W] = x
    out = bias.double().reshape(1, 1, 1, N).repeat(images, Ho, Wo, 1)
    for a in range(kh):
        for b in range(kw):
            r0, c0 = a - ph + P, b - pw + P
            xs = xp[:, r0:r0 + Ho * sh:sh, c0:c0 + Wo * sw:sw]
            out += torch.einsum("ihwc,nc->ihwn", xs, w[:, :, a, b].double())
    if resid is not None:
        out += resid.double().reshape(images, Ho, Wo, N)
    return out


# name, images, H, W, C0, C1, N, (kh, kw), (sh, sw), (ph, pw), resid, ln, parity (None = out_mode 0)
CASES = [
    # ---- row panels, 1 tap
    ("panel 64->64 +res", 3, 5, 7, 64, 0, 64, (1, 1), (1, 1), (0, 0), True, False, None),
    ("panel 128->128 +res", 3, 5, 7, 128, 0, 128, (1, 1), (1, 1), (0, 0), True, False, None),
    ("panel concat 64+64->64", 3, 5, 7, 64, 64, 64, (1, 1), (1, 1), (0, 0), False, False, None),
    ("panel 256->256 +res", 3, 5, 7, 256, 0, 256, (1, 1), (1, 1), (0, 0), True, False, None),
    ("panel LN 256->384", 3, 5, 7, 256, 0, 384, (1, 1), (1, 1), (0, 0), False, True, None),
    ("panel LN 64->384", 5, 3, 9, 64, 0, 384, (1, 1), (1, 1), (0, 0), False, True, None),
    ("panel 32-row 512->128", 3, 5, 7, 256, 256, 128, (1, 1), (1, 1), (0, 0), False, False, None),
    ("panel 32-row 512->256 +res", 2, 5, 7, 512, 0, 256, (1, 1), (1, 1), (0, 0), True, False, None),
    # ---- row panels, few taps (im2col rows); ConvTranspose parity classes scatter into [images][2H][2W][N]
    ("taps 2x2 64->64", 3, 5, 7, 64, 0, 64, (2, 2), (1, 1), (0, 0), False, False, None),
    ("taps 2x2 64->64 parity (1,0)", 3, 5, 7, 64, 0, 64, (2, 2), (1, 1), (1, 0), False, False, (1, 0)),
    ("taps 2x2 64->64 parity (0,1)", 3, 6, 5, 64, 0, 64, (2, 2), (1, 1), (0, 1), False, False, (0, 1)),
    ("taps 2x2 64->64 parity (0,0), 40 x 33 images (tile inside a row, across rows and images)", 2, 40, 33, 64, 0, 64, (2, 2), (1, 1), (0, 0), False, False, (0, 0)),
    ("taps 2x2 concat 64+64->128 parity (1,0)", 3, 9, 11, 64, 64, 128, (2, 2), (1, 1), (1, 0), False, False, (1, 0)),
    ("taps 2x2 128->128 parity (1,1) 32-row", 3, 5, 7, 128, 0, 128, (2, 2), (1, 1), (1, 1), False, False, (1, 1)),
    ("taps 2x2 s2 64->128 (pixel-unshuffle conv)", 3, 6, 10, 64, 0, 128, (2, 2), (2, 2), (0, 0), False, False, None),
    # ---- neither: K % 32 != 0 and N = 192 stay on the narrow kernel (shared epilogue, fragment-order pack)
    ("narrow 36->64 +res", 3, 5, 7, 36, 0, 64, (1, 1), (1, 1), (0, 0), True, False, None),
    ("narrow 64->192", 3, 5, 7, 64, 0, 192, (1, 1), (1, 1), (0, 0), False, False, None),
    ("narrow 3x3 64->64 (18 chunks)", 3, 5, 7, 64, 0, 64, (3, 3), (1, 1), (1, 1), True, False, None),
    # ---- image tiles (r04, csrc/igemm_img.hip): 3 x 3 on images at most 32 wide that are not 8 x 8-tileable, >= 24 (tap, chunk) iterations,
    #      N % 128 == 0: unique pixels staged once per 32-channel block; residual, concat, split-K by N, the 3-tap form of the H = 1 level,
    #      tiles that hold exactly one image (4 x 32), several images (2 x 16) and straddle images (4 x 13)
    ("img 3x3 256->256", 5, 4, 13, 256, 0, 256, (3, 3), (1, 1), (1, 1), True, False, None),
    ("img 3x3 concat 128+128->128", 5, 4, 13, 128, 128, 128, (3, 3), (1, 1), (1, 1), False, False, None),
    ("img 3x3 256->256, 4 x 32 images (one image per tile)", 3, 4, 32, 256, 0, 256, (3, 3), (1, 1), (1, 1), False, False, None),
    ("img 3x3 concat 256+128->128, 2 x 16 images", 7, 2, 16, 256, 128, 128, (3, 3), (1, 1), (1, 1), True, False, None),
    ("img 3x3 96->128 (3 blocks, the shortest reduction it takes)", 3, 3, 5, 96, 0, 128, (3, 3), (1, 1), (1, 1), False, False, None),
    ("img 3x3 concat 104+24->128 (a block straddles the sources)", 4, 4, 11, 104, 24, 128, (3, 3), (1, 1), (1, 1), False, False, None),
    # ---- wide: >= 24 chunks.  128-column tiles, 64-column tiles, split-K by N (512: 2 slices, 1024: 4), K % 32 != 0, concat
    ("wide 3x3 256->256 (40-wide images)", 2, 3, 40, 256, 0, 256, (3, 3), (1, 1), (1, 1), True, False, None),
    ("wide 3x3 concat 128+128->128 (40-wide images)", 2, 3, 40, 128, 128, 128, (3, 3), (1, 1), (1, 1), False, False, None),
    ("wide 3x3 100->64 (64-column tile, K % 32 != 0)", 5, 4, 13, 100, 0, 64, (3, 3), (1, 1), (1, 1), False, False, None),
    ("wide 4x4 s2 64->64 (down conv)", 3, 10, 14, 64, 0, 64, (4, 4), (2, 2), (1, 1), False, False, None),
    ("img 3x3 256->512 split 2", 5, 2, 9, 256, 0, 512, (3, 3), (1, 1), (1, 1), True, False, None),
    ("img 3x3 512->1024 split 4, 1 x 8 images (dead taps: 3-tap form)", 7, 1, 8, 512, 0, 1024, (3, 3), (1, 1), (1, 1), False, False, None),
    ("wide 3x3 256->512 split 2 (40-wide images)", 2, 2, 40, 256, 0, 512, (3, 3), (1, 1), (1, 1), True, False, None),
    ("wide 1x1 1024->512 split 2", 3, 5, 7, 1024, 0, 512, (1, 1), (1, 1), (0, 0), True, False, None),
]


def run_case(SH, dev, case, mode, images_used=None):
    name, images, H, W, C0, C1, N, (kh, kw), (sh, sw), (ph, pw), res, use_ln, par = case
    g = torch.Generator().manual_seed(sum(ord(ch) * (i + 1) for i, ch in enumerate(name)) % (1 << 31))
    K = C0 + C1
    w = torch.randn(N, K, kh, kw, generator=g) / (K * kh * kw) ** 0.5
    x0 = torch.randn(images, H, W, C0, generator=g)
    x1 = torch.randn(images, H, W, C1, generator=g) if C1 else None
    bias = torch.randn(N, generator=g)
    if par is None and (kh, kw) == (2, 2) and (sh, sw) == (2, 2):
        Ho, Wo = H // 2, W // 2
    elif (kh, kw) == (4, 4):
        Ho, Wo = H // 2, W // 2
    else:
        Ho, Wo = H, W
    resid = torch.randn(images, Ho, Wo, N, generator=g) if res else None
    ln = None
    if use_ln:
        mu = x0.mean(-1)
        inv = (x0.var(-1, unbiased=False) + 1e-5).rsqrt()
        ln = (torch.stack([mu, inv], -1).reshape(-1, 2).contiguous(), 1 + 0.1 * torch.randn(K, generator=g))
    n_img = images if images_used is None else images_used
    conv = SH._Conv(w.to(dev), sh=sh, sw=sw, ph=ph, pw=pw, mode=mode)
    a0 = x0[:n_img].reshape(-1, C0).contiguous().to(dev)
    a1 = x1[:n_img].reshape(-1, C1).contiguous().to(dev) if C1 else None
    rd = resid[:n_img].reshape(-1, N).contiguous().to(dev) if res else None
    lnd = (ln[0][:n_img * H * W].contiguous().to(dev), ln[1].to(dev)) if ln else None
    if par is None:
        out = conv(a0, n_img, H, W, a1=a1, bias=bias.to(dev), resid=rd, ln=lnd, Ho=Ho, Wo=Wo)
        got = out.reshape(n_img, Ho, Wo, N)
    else:
        out = torch.full((n_img * 4 * Ho * Wo, N), float("nan"), device=dev)
        conv(a0, n_img, H, W, a1=a1, bias=bias.to(dev), out=out, Ho=Ho, Wo=Wo, out_mode=2, par=par)
        full = out.reshape(n_img, 2 * Ho, 2 * Wo, N)
        got = full[:, par[0]::2, par[1]::2]
        rest = full.clone()
        rest[:, par[0]::2, par[1]::2] = 0
        assert torch.isnan(rest).sum() == torch.isnan(full).sum(), "parity class wrote outside its own pixels"
        assert not torch.isnan(got).any()
    torch.cuda.synchronize()
    ref = reference(x0, x1, w, bias, resid, ln, sh, sw, ph, pw, Ho, Wo) if images_used is None else None
    return got.cpu(), ref


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_igemm_f16x3_vs_fp64_and_batch_independence(case, dev, SH):
    got, ref = run_case(SH, dev, case, "f16x3")
    err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{case[0]}: {err:.2e} of the output range")
    assert torch.isfinite(got).all()
    assert err < 3e-6, (case[0], err)
    n_pref = max(1, case[1] - 2)
    part, _ = run_case(SH, dev, case, "f16x3", images_used=n_pref)
    assert torch.equal(part, got[:n_pref]), f"{case[0]}: rows depend on the batch (prefix of {n_pref} images differs)"


@pytest.mark.parametrize("case", [CASES[i] for i in (3, 4, 9, 16, 20, 21)], ids=lambda c: c[0])
def test_igemm_x6_vs_fp64(case, dev, SH):
    got, ref = run_case(SH, dev, case, "x6")
    err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-6, (case[0], err)
