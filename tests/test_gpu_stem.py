"""GPU parity of the 7x7x7 stem convolution (init_conv of Unet3D_with_Conv3D, video_diffusion_pytorch_conv3d.py:392:
nn.Conv3d(channels, dim, 7, padding 3)) behind dpc_stem_pack / dpc_stem_run (include/dpc.h) against torch's conv3d in fp64.

The default f16x3 kernel (csrc/stem7x6.hip: stem7p_kernel) packs one CHANNEL PAIR x 8 w-taps per MFMA k-step; the cases cover every
pair count (C = 1 .. 8, odd C = a half-empty last pair), a channel slice of a wider state tensor (the reference-layout
[B][F][ctot][H][W] state the samplers pass), extents that are not multiples of the 4 x 4 x 8 output tile and fewer frames / rows
than the 7-tap window.  Tolerance: 3e-6 of the output range (the conv op tests' f16x3 bound); 2e-6 for the exact x6 mode.
Batch independence: a prefix of the batch gives bit-identical rows.
"""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


def stem(x, w, bias, c_off, mode, dev):
    from diffphycon_amd import _lib
    L = _lib.lib()
    B, F, ctot, H, W = x.shape
    N, Cin = w.shape[:2]
    h = C.c_void_p()
    wd, xd, bd = w.contiguous().to(dev), x.contiguous().to(dev), bias.to(dev)
    _lib.check(L.dpc_stem_pack(_lib.ptr(wd), N, Cin, 7, mode.encode(), C.byref(h), _lib.stream()))
    out = torch.full((B, F, H, W, N), float("nan"), device=dev)
    try:
        _lib.check(L.dpc_stem_run(h, _lib.ptr(xd), ctot, c_off, _lib.ptr(bd), _lib.ptr(out), B, F, H, W, _lib.stream()))
        torch.cuda.synchronize()
    finally:
        L.dpc_stem_free(h)
    return out.cpu()


# name, B, F, H, W, ctot, c_off, C, N
CASES = [
    ("C=6 of 6 (joint smoke denoiser)", 3, 8, 16, 16, 6, 0, 6, 64),
    ("C=2 slice of 6 (prior smoke denoiser)", 3, 8, 16, 16, 6, 3, 2, 64),
    ("C=1", 2, 5, 9, 11, 1, 0, 1, 64),
    ("C=3 of 4, ragged extents", 2, 6, 10, 13, 4, 1, 3, 64),
    ("C=4", 2, 4, 8, 8, 4, 0, 4, 64),
    ("C=5, fewer frames and rows than taps", 2, 3, 5, 20, 5, 0, 5, 64),
    ("C=7 (jellyfish joint denoiser), N=32", 2, 4, 12, 12, 7, 0, 7, 32),
    ("C=8, N=128", 2, 4, 8, 16, 8, 0, 8, 128),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("mode", ["f16x3", "x6"])
def test_stem_vs_fp64(case, mode, dev):
    name, B, F, H, W, ctot, c_off, Cin, N = case
    g = torch.Generator().manual_seed(1234 + 17 * Cin + H)
    x = torch.randn(B, F, ctot, H, W, generator=g)
    w = torch.randn(N, Cin, 7, 7, 7, generator=g) / (Cin * 343) ** 0.5
    bias = torch.randn(N, generator=g)
    got = stem(x, w, bias, c_off, mode, dev)
    xs = x[:, :, c_off:c_off + Cin].permute(0, 2, 1, 3, 4).double()              # [B, C, F, H, W]
    ref = torch.nn.functional.conv3d(xs, w.double(), bias.double(), padding=3).permute(0, 2, 3, 4, 1)
    assert torch.isfinite(got).all()
    err = ((got.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{name} [{mode}]: {err:.2e} of the output range")
    assert err < (3e-6 if mode == "f16x3" else 2e-6), (name, mode, err)
    part = stem(x[:max(1, B - 1)], w, bias, c_off, mode, dev)
    assert torch.equal(part, got[:max(1, B - 1)]), f"{name}: rows depend on the batch"
