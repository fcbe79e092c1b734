"""GPU parity of the operator-level C-ABI entry points (include/dpc.h) against the CPU oracle / plain torch fp32.

Run on the MI355X box:  python -m pytest tests -m gpu
Tolerances are stated per test (per-block fp32 rel 1e-5 / abs 1e-6 scaled by the reduction length).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def L():
    from diffphycon_amd import _lib
    return _lib


def to_cl(x):      # [B,C,F,H,W] -> [B,F,H,W,C]
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_cf(x):      # [B,F,H,W,C] -> [B,C,F,H,W]
    return x.permute(0, 4, 1, 2, 3).contiguous()


def relerr(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


CONV_CASES = [
    # B, F, H, W, Cin, Cout, k, stride, pad
    (2, 4, 16, 16, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 3, 8, 12, 16, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 5, 16, 16, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 4, 8, 8, 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 4, 16, 16, 64, 64, (1, 4, 4), (1, 2, 2), (0, 1, 1)),
    (2, 3, 8, 8, 128, 6, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, 2, 6, 10, 40, 72, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # shapes the big-tile f16x3 kernel takes (F % 8 == 0 / F % 4 == 0 for Cout > 64, H % 8 == 0, W % 8 == 0)
    (1, 8, 16, 16, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 8, 8, 16, 128, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 4, 16, 8, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 8, 8, 8, 48, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 16, 8, 8, 20, 40, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 20, 16, 16, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),         # partial frame tile (jellyfish: 20 frames)
    # Winograd F(2,3)-over-frames kernel (conv3w.hip: H % 8 == 0, W % 8 == 0, Cout % 64 == 0): partial tiles with an odd frame
    # pair / an odd last frame, one-tile planes, K not a multiple of 16, several samples and column tiles
    (1, 18, 8, 8, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 17, 8, 16, 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (3, 4, 8, 8, 20, 192, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 12, 24, 8, 72, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # ... with several 64-wide column tiles per plane tile
    (2, 18, 8, 8, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 17, 8, 16, 32, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 4, 16, 16, 128, 384, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # (1,3,3): the 2-D U-Net's 3x3 convolutions on the halo kernel, batch on the frame axis (any batch size, partial tiles)
    (1, 16, 16, 128, 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (3, 1, 8, 64, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (5, 1, 16, 24, 12, 72, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 3, 4, 8, 16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1)),            # H % 8 != 0: implicit-GEMM path
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_cl(case, dev, L):
    B, Fr, H, W, Ci, Co, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    x = torch.randn(B, Ci, Fr, H, W, generator=g)
    w = torch.randn(Co, Ci, *k, generator=g) / (Ci * k[0] * k[1] * k[2]) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv3d(x.double(), w.double(), b.double(), stride=s, padding=p).float()
    xd, wd, bd = to_cl(x).to(dev), w.to(dev).contiguous(), b.to(dev)
    out = torch.empty(to_cl(ref).shape, device=dev)
    nb = L.lib().dpc_conv_workspace_bytes(Ci, Co, k[0] * k[1] * k[2])
    ws = L.workspace(nb, dev)
    L.check(L.lib().dpc_conv3d_cl(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), B, Fr, H, W, Ci, Co, *k, *s, *p,
                                  C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    got = to_cf(out.cpu())
    # split operands (22 bits) + fp32 accumulation over K = Cin * taps terms vs an fp64 reference: measured 1.3e-7 .. 9.5e-7 of the
    # output range over these cases (profiles/r03_a_conv_errors.log); 3e-6 = 3 x the largest, inside SURVEY 8d's per-block 1e-5
    assert relerr(got, ref) < 3e-6, relerr(got, ref)


@pytest.mark.parametrize("shape", [(2, 32, 64, 64, 64, 64), (1, 32, 32, 32, 256, 128), (1, 64, 32, 32, 128, 128), (2, 20, 32, 32, 64, 128)])
def test_conv3d_winograd_at_full_layer_extents_sampled_vs_fp64(shape, dev, L):
    """The Winograd F(4,3) kernel at the REAL extents of the BASELINE configs' layers (S64 level 0: 32 x 64 x 64, 64 -> 64; a concatenated
    up-path layer 256 -> 128; S128's 64 frames; J128's 20 frames) -- too large for a full fp64 reference in the GPU suite, so the launch
    is full-size and the check is exact fp64 dot products at sampled output positions: 2048 random ones and every combination of the
    tile seams (frames 0, 3, 4, F-1; rows / columns 0, 7, 8, last) for three output channels.  Same 3e-6-of-range bound as the small
    cases (the persistent tile walk, the XCD-aware tile order and the deferred stores only show at sizes with many tiles per CU)."""
    B, Fr, H, W, Ci, Co = shape
    g = torch.Generator(device=dev).manual_seed(Ci + Co + Fr)
    x = torch.randn(B, Fr, H, W, Ci, device=dev, generator=g)                 # channels-last
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev, generator=g) / (Ci * 27) ** 0.5
    b = torch.randn(Co, device=dev, generator=g)
    out = torch.empty(B, Fr, H, W, Co, device=dev)
    ws = L.workspace(L.lib().dpc_conv_workspace_bytes(Ci, Co, 27), dev)
    L.check(L.lib().dpc_conv3d_cl(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(out), B, Fr, H, W, Ci, Co, 3, 3, 3, 1, 1, 1, 1, 1, 1,
                                  C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    rng = np.random.RandomState(Ci * 7 + Fr)
    pts = [(rng.randint(B), rng.randint(Fr), rng.randint(H), rng.randint(W), rng.randint(Co)) for _ in range(2048)]
    for f in sorted({0, 3, 4, Fr - 1}):
        for h in sorted({0, 7, 8, H - 1}):
            for ww in sorted({0, 7, 8, W - 1}):
                for n in (0, Co // 2 + 1, Co - 1):
                    pts.append((B - 1, f, h, ww, n))
    idx = torch.tensor(pts, device=dev)
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1, 1, 1))          # zero halo on f, h, w
    bi, fi, hi, wi, ni = idx.unbind(1)
    ref = b.double()[ni].clone()
    wd = w.double()
    for df in range(3):
        for dh in range(3):
            for dw in range(3):
                ref += (xp[bi, fi + df, hi + dh, wi + dw] * wd[ni, :, df, dh, dw]).sum(1)
    got = out[bi, fi, hi, wi, ni].double()
    scale = out.abs().max().double()
    err = ((got - ref).abs().max() / scale).item()
    print(f"conv3w4 {shape}: sampled error {err:.3e} of the output range at {len(pts)} points")
    assert err < 3e-6, (shape, err)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("scale", [1e-3, 1e-2, 3e-2, 100.0])
def test_conv3d_winograd_small_and_large_inputs(scale, dev, L):
    """The Winograd kernel splits a plain (un-normalised) input with a pre-scale of 2 only (F(4,3): |B^T d| <= 7 |d| must stay below
    65504 for |d| <= 4094, the range every f16x3 kernel guarantees), where the direct kernels use 2^4: for a tensor whose values are
    ALL of magnitude ~1e-3 the remainder plane sits in fp16's subnormals and carries 4-5 bits.  The 3e-6-of-range bound of the
    other cases holds from |x| ~ 1e-2 upwards; at 1e-3 the error stays inside SURVEY 8d's per-block 1e-5 (measured 9.5e-6 with
    F(4,3); F(2,3) with its pre-scale of 8 stayed below 3e-6 there).  Magnitude 100 checks the other end of the window."""
    B, Fr, H, W, Ci, Co = 1, 8, 16, 16, 64, 64
    g = torch.Generator().manual_seed(int(scale * 1000) + 5)
    x = torch.randn(B, Ci, Fr, H, W, generator=g) * scale
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
    b = torch.randn(Co, generator=g) * scale
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=1).float()
    xd, wd, bd = to_cl(x).to(dev), w.to(dev).contiguous(), b.to(dev)
    out = torch.empty(to_cl(ref).shape, device=dev)
    ws = L.workspace(L.lib().dpc_conv_workspace_bytes(Ci, Co, 27), dev)
    L.check(L.lib().dpc_conv3d_cl(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), B, Fr, H, W, Ci, Co, 3, 3, 3, 1, 1, 1, 1, 1, 1,
                                  C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    err = relerr(to_cf(out.cpu()), ref)
    print(f"conv3w input scale {scale:g}: relative error {err:.3e}")
    assert err < (1e-5 if scale < 1e-2 else 3e-6), (scale, err)


@pytest.mark.parametrize("seed", range(10))
def test_conv3d_winograd_random_shapes(seed, dev, L):
    """Random shapes inside the Winograd kernel's domain (conv3w.hip: 3x3x3, H % 8 == 0, W % 8 == 0, Cin % 32 == 0, Cout % 64 == 0, F % 4 == 0
    or F >= 16): several samples, fewer tiles than CUs as well as many, partial frame tiles, 1-4 column tiles, 2-8 channel chunks."""
    import random
    rnd = random.Random(1000 + seed)
    B = rnd.choice([1, 2, 3])
    Fr = rnd.choice([4, 8, 12, 16, 17, 18, 19, 20])
    H, W = 8 * rnd.choice([1, 2, 3]), 8 * rnd.choice([1, 2, 4])
    Ci, Co = 32 * rnd.choice([1, 2, 3, 4, 8]), 64 * rnd.choice([1, 2, 3, 4])
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, Fr, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=1).float()
    xd, wd, bd = to_cl(x).to(dev), w.to(dev).contiguous(), b.to(dev)
    out = torch.empty(to_cl(ref).shape, device=dev)
    nb = L.lib().dpc_conv_workspace_bytes(Ci, Co, 27)
    ws = L.workspace(nb, dev)
    L.check(L.lib().dpc_conv3d_cl(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), B, Fr, H, W, Ci, Co, 3, 3, 3, 1, 1, 1, 1, 1, 1,
                                  C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    got = to_cf(out.cpu())
    assert relerr(got, ref) < 2e-6 * (Ci * 27) ** 0.5 + 1e-6, ((B, Fr, H, W, Ci, Co), relerr(got, ref))


@pytest.mark.parametrize("shape", [(2, 4, 8, 8, 16, 16), (1, 3, 16, 16, 128, 128), (1, 2, 4, 6, 64, 32)])
def test_convtranspose3d_144(shape, dev, L):
    B, Fr, H, W, Ci, Co = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, Fr, H, W, generator=g)
    w = torch.randn(Ci, Co, 1, 4, 4, generator=g) / (Ci * 4) ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv_transpose3d(x.double(), w.double(), b.double(), stride=(1, 2, 2), padding=(0, 1, 1)).float()
    xd, wd, bd = to_cl(x).to(dev), w.to(dev).contiguous(), b.to(dev)
    out = torch.full(to_cl(ref).shape, float("nan"), device=dev)
    nb = 4 * L.lib().dpc_conv_workspace_bytes(Ci, Co, 4)
    ws = L.workspace(nb, dev)
    L.check(L.lib().dpc_convtranspose3d_144_cl(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), B, Fr, H, W, Ci, Co,
                                               C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    got = to_cf(out.cpu())
    assert torch.isfinite(got).all()
    assert relerr(got, ref) < 1e-5, relerr(got, ref)


@pytest.mark.parametrize("B,R,Cc,G,ss", [(2, 1024, 8, 8, True), (2, 4 * 16 * 16, 16, 8, False), (1, 32 * 64 * 64, 64, 8, True),
                                         (3, 2000, 256, 8, True), (2, 333, 64, 1, False)])
def test_groupnorm_silu(B, R, Cc, G, ss, dev, L):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, R, Cc, generator=g) * 2 + 0.7
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    scsh = torch.randn(B, 2 * Cc, generator=g) * 0.3 if ss else None
    xc = x.permute(0, 2, 1).double()                                   # [B,C,R]
    ref = F.group_norm(xc, G, gamma.double(), beta.double(), eps=1e-5)
    if ss:
        sc, sh = scsh.double().chunk(2, dim=1)
        ref = ref * (sc[:, :, None] + 1) + sh[:, :, None]
    ref = F.silu(ref).permute(0, 2, 1).float()
    xd = x.to(dev).contiguous()
    gd, bd, sd = gamma.to(dev), beta.to(dev), (scsh.to(dev) if ss else None)   # keep alive until the launch
    ws = L.workspace(L.lib().dpc_groupnorm_workspace_bytes(B, Cc), dev)
    L.check(L.lib().dpc_groupnorm_silu_cl(L.ptr(xd), L.ptr(gd), L.ptr(bd), L.ptr(sd), B, R, Cc, G,
                                          C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    got = xd.cpu()
    assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5), (got - ref).abs().max()


def _attn_ref(qkv, heads, rot, bias):
    """qkv [..., n, 3*heads*32] -> [..., n, heads*32], float64 reference of Attention.forward (…conv3d.py:311-351)."""
    from oracle import unet3d as O
    q, k, v = [z.reshape(*z.shape[:-1], heads, 32).transpose(-2, -3) for z in qkv.double().chunk(3, dim=-1)]
    q = q * 32 ** -0.5
    if rot:
        q, k = O.rotary(q.float()).double(), O.rotary(k.float()).double()
    sim = torch.einsum("...hid,...hjd->...hij", q, k)
    if bias is not None:
        sim = sim + bias.double()
    attn = sim.softmax(dim=-1)
    out = torch.einsum("...hij,...hjd->...hid", attn, v)
    return out.transpose(-2, -3).reshape(*qkv.shape[:-1], heads * 32).float()


@pytest.mark.parametrize("B,Fr,HW", [(2, 4, 16), (1, 32, 64), (2, 20, 12), (1, 64, 8), (1, 33, 5)])
def test_attention_core_temporal(B, Fr, HW, dev, L):
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import _rotary_tables
    heads = 4
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, Fr, HW, 3 * heads * 32, generator=g)           # rows ordered (b, f, pixel)
    bias = torch.randn(heads, Fr, Fr, generator=g)
    ref = _attn_ref(qkv.permute(0, 2, 1, 3), heads, True, bias).permute(0, 2, 1, 3).contiguous()
    cos, sin = _rotary_tables(Fr, 32)
    out = torch.empty(B, Fr, HW, heads * 32, device=dev)
    qd, cd, sd, bd = qkv.to(dev), cos.to(dev), sin.to(dev), bias.to(dev)
    L.check(L.lib().dpc_attention_core(L.ptr(qd), L.ptr(out), heads, Fr, B * HW, HW, Fr * HW, 1, HW,
                                       L.ptr(cd), L.ptr(sd), L.ptr(bd), L.stream()))
    got = out.cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=2e-5), (got - ref).abs().max()


@pytest.mark.parametrize("BF,N", [(3, 64), (2, 256), (2, 16), (1, 100)])
def test_attention_core_spatial(BF, N, dev, L):
    heads = 4
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(BF, N, 3 * heads * 32, generator=g) * 1.5
    ref = _attn_ref(qkv, heads, False, None)
    out = torch.empty(BF, N, heads * 32, device=dev)
    qd = qkv.to(dev)
    L.check(L.lib().dpc_attention_core(L.ptr(qd), L.ptr(out), heads, N, BF, 1, N, 0, 1, None, None, None, L.stream()))
    got = out.cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=2e-5), (got - ref).abs().max()


@pytest.mark.parametrize("imgs,N", [(3, 256), (2, 64), (1, 4096), (2, 50)])
def test_linear_attention_core(imgs, N, dev, L):
    heads = 4
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(imgs, N, 3 * heads * 32, generator=g) * 1.5
    q, k, v = [z.reshape(imgs, N, heads, 32).permute(0, 2, 3, 1).double() for z in qkv.chunk(3, dim=-1)]   # b h d n
    q = q.softmax(dim=-2) * 32 ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    ref = torch.einsum("bhde,bhdn->bhen", ctx, q).permute(0, 3, 1, 2).reshape(imgs, N, heads * 32).float()
    out = torch.empty(imgs, N, heads * 32, device=dev)
    ws = L.workspace(L.lib().dpc_linear_attention_workspace_bytes(imgs, heads), dev)
    qd = qkv.to(dev)
    L.check(L.lib().dpc_linear_attention_core(L.ptr(qd), L.ptr(out), heads, imgs, N,
                                              C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    got = out.cpu()
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-6), (got - ref).abs().max()


def test_burgers_fd_bit_exact(dev, L):
    from oracle import burgers as OB
    g = load_golden("burgers_fd")
    for u0, f, visc, T, dt, nt, key in ((g["u0"], g["f"], 0.01, 1.0, 1e-4, 10, "traj"),
                                        (g["u0b"], g["fb"], 0.02, 0.5, 5e-4, 4, "trajb")):
        N, nx = u0.shape
        out = torch.empty(N, nt + 1, nx, device=dev)
        ud, fd = torch.from_numpy(u0).to(dev), torch.from_numpy(f).to(dev)
        L.check(L.lib().dpc_burgers_fd(L.ptr(ud), L.ptr(fd), L.ptr(out), N, nx, nt, visc, T, dt, L.stream()))
        got = out.cpu().numpy()
        # integer index schedule + fp32 stencil: bit-exact against the oracle AND against the reference fixture
        assert np.array_equal(got, OB.burgers_numeric_solve_free(u0, f, visc, T, dt, nt))
        assert np.array_equal(got, g[key])
    # a batch that does not fill the last workgroup, synthetic inputs
    u0, f = OB.synthetic_inputs(50, 128, 10, seed=3)
    out = torch.empty(50, 11, 128, device=dev)
    ud, fd = torch.from_numpy(u0).to(dev), torch.from_numpy(f).to(dev)
    L.check(L.lib().dpc_burgers_fd(L.ptr(ud), L.ptr(fd), L.ptr(out), 50, 128, 10, 0.01, 1.0, 1e-4, L.stream()))
    assert np.array_equal(out.cpu().numpy(), OB.burgers_numeric_solve_free(u0, f, 0.01, 1.0, 1e-4, 10))


def test_philox_normal_sharding_invariance(dev, L):
    per = 4 * 6 * 16 * 16 + 3
    full = torch.empty(8, per, device=dev)
    L.check(L.lib().dpc_philox_normal(L.ptr(full), 8, per, 1234, 0, 5, L.stream()))
    part = torch.empty(3, per, device=dev)
    L.check(L.lib().dpc_philox_normal(L.ptr(part), 3, per, 1234, 4, 5, L.stream()))
    assert torch.equal(full[4:7], part)                       # trajectory 4..6 drawn identically on "another rank"
    other = torch.empty(8, per, device=dev)
    L.check(L.lib().dpc_philox_normal(L.ptr(other), 8, per, 1234, 0, 6, L.stream()))
    assert not torch.equal(full, other)
    big = torch.empty(64, 65536, device=dev)
    L.check(L.lib().dpc_philox_normal(L.ptr(big), 64, 65536, 7, 0, 0, L.stream()))
    assert abs(big.mean().item()) < 2e-3 and abs(big.std().item() - 1) < 2e-3
    assert abs((big ** 4).mean().item() - 3) < 0.05


def test_hardware_fp16_saturation_of_the_operand_split(dev):
    """The fused attention kernels (csrc/tattn3.hip, lattn3.hip) dropped the software clamp in front of the fp16 operand conversion in
    r04 and set MODE.FP16_OVFL instead (csrc/f16x3.h: hw_sat_enable): an overflowing conversion must SATURATE at +-65504, never give inf.
    In range the split is the exact 22-bit one: hi + lo == x to 2^-22 |x|, hi = fp16(x)."""
    from diffphycon_amd import _lib as L
    x = torch.tensor([0.0, 1.0, -3.25, 1234.567, 65504.0, 65519.9, 70000.0, -1.0e5, 1.0e6, -3.0e38, 4093.999, 6.1e-5], device=dev)
    out = torch.empty(2 * x.numel(), device=dev)
    L.check(L.lib().dpc_selftest_fp16_clamp(L.ptr(x), L.ptr(out), x.numel(), L.stream()))
    hi, lo = out[0::2].cpu(), out[1::2].cpu()
    xc = x.cpu()
    assert torch.isfinite(out).all(), out                         # saturation, not inf
    big = xc.abs() > 65504
    assert torch.equal(hi[big], torch.sign(xc[big]) * 65504.0)
    assert torch.equal(hi[~big], xc[~big].half().float())
    assert ((hi + lo)[~big] - xc[~big]).abs().max() <= 2.0 ** -21 * xc[~big].abs().max()
    assert torch.equal(lo[big], (xc[big] - hi[big]).clamp(-65504, 65504).half().float())
