"""Implicit-GEMM micro-bench on the GPU box (through the C ABI: dpc_conv_pack / dpc_conv_run), the shapes of both workloads:
the Burgers U-Net's GEMM-shaped deep levels (3x3 on 4x32 / 2x16 / 1x8 images, batch 256) and the smoke U-Net's 1x1x1 projections.
Each shape: f16x3 (default) against the exact x6 mode (max error relative to the output range), median of `reps` event-timed runs.
  gpurun -- 'python tools/bench_igemm.py [burgers|smoke|flat|all] [reps]'
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffphycon_amd.model import surrogates_hip as SH

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20

# (name, images, H, W, C0, C1, N, k, resid)
BURGERS = [
    ("b 4x32 256->256 3x3", 256, 4, 32, 256, 0, 256, 3, False),
    ("b 4x32 512->256 3x3 (concat)", 256, 4, 32, 256, 256, 256, 3, False),
    ("b 2x16 512->512 3x3", 256, 2, 16, 512, 0, 512, 3, False),
    ("b 2x16 1024->512 3x3 (concat)", 256, 2, 16, 512, 512, 512, 3, False),
    ("b 1x8 1024->1024 3x3", 256, 1, 8, 1024, 0, 1024, 3, False),
    ("b 1x8 512->1024 3x3", 256, 1, 8, 512, 0, 1024, 3, False),
    ("b 1x8 1024->1024 1x1 +res", 256, 1, 8, 1024, 0, 1024, 1, True),
    ("b 2x16 1024->512 1x1", 256, 2, 16, 512, 512, 512, 1, False),
    ("b 8x64 128->384 1x1", 256, 8, 64, 128, 0, 384, 1, False),
]
FLAT = [   # the (1,3,3) halo-kernel shapes (H % 8 == 0, W % 8 == 0): Burgers levels 0 / 1, the jellyfish surrogates at J128
    ("b 16x128 64->64 3x3", 256, 16, 128, 64, 0, 64, 3, False),
    ("b 16x128 128->64 3x3 (concat)", 256, 16, 128, 64, 64, 64, 3, False),
    ("b 8x64 64->64 3x3", 256, 8, 64, 64, 0, 64, 3, False),
    ("b 8x64 128->64 3x3 (concat)", 256, 8, 64, 64, 64, 64, 3, False),
    ("j 128x128 64->64 3x3", 320, 128, 128, 64, 0, 64, 3, False),
    ("j 64x64 128->128 3x3", 320, 64, 64, 128, 0, 128, 3, False),
    ("j 32x32 256->256 3x3", 320, 32, 32, 256, 0, 256, 3, False),
    ("j 16x16 512->512 3x3", 320, 16, 16, 512, 0, 512, 3, False),
]
SMOKE = [
    ("s 16x16 256->256 1x1 +res", 512, 16, 16, 256, 0, 256, 1, True),
    ("s 16x16 256->384 1x1", 512, 16, 16, 256, 0, 384, 1, False),
    ("s 16x16 256->384 1x1 LN", 512, 16, 16, 256, 0, 384, 1, "ln"),
    ("s 16x16 128->256 1x1 +res", 512, 16, 16, 128, 0, 256, 1, True),
    ("s 32x32 128->128 1x1 +res", 512, 32, 32, 128, 0, 128, 1, True),
    ("s 32x32 256->128 1x1 (concat)", 512, 32, 32, 128, 128, 128, 1, False),
    ("s 64x64 128->64 1x1 (concat)", 512, 64, 64, 64, 64, 64, 1, False),
    ("s 64x64 64->64 1x1 +res", 512, 64, 64, 64, 0, 64, 1, True),
    ("s 64x64 64->384 1x1", 512, 64, 64, 64, 0, 384, 1, False),
    ("s 32x32 64->64 2x2 (ConvT class)", 512, 32, 32, 64, 0, 64, 2, False),
    ("s 16x16 128->128 2x2 (ConvT class)", 512, 16, 16, 128, 0, 128, 2, False),
    ("s 64x64->32x32 64->64 4x4 s2", 512, 64, 64, 64, 0, 64, 4, False),
]
shapes = (BURGERS if which in ("burgers", "all") else []) + (SMOKE if which in ("smoke", "all") else []) + (FLAT if which in ("flat", "all") else [])
g = torch.Generator(device="cpu").manual_seed(0)
for name, images, H, W, C0, C1, N, k, res in shapes:
    K = C0 + C1
    w = (torch.randn(N, K, k, k, generator=g) / (K * k * k) ** 0.5).to(dev)
    a0 = torch.randn(images * H * W, C0, generator=g).to(dev)
    a1 = torch.randn(images * H * W, C1, generator=g).to(dev) if C1 else None
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(images * (H // 2 if k == 4 else H) * (W // 2 if k == 4 else W), N, generator=g).to(dev) if res is True else None
    ln = None
    if res == "ln":
        mu = a0.mean(1); inv = (a0.var(1, unbiased=False) + 1e-5).rsqrt()
        ln = (torch.stack([mu, inv], 1).contiguous(), (1 + 0.1 * torch.randn(K, generator=g)).to(dev))
    kw = dict(sh=2, sw=2, ph=1, pw=1) if k == 4 else {}
    Ho, Wo = (H // 2, W // 2) if k == 4 else (H, W)
    c3 = SH._Conv(w, mode="f16x3", **kw)
    c6 = SH._Conv(w, mode="x6", **kw)
    out3 = c3(a0, images, H, W, a1=a1, bias=bias, resid=resid, ln=ln, Ho=Ho, Wo=Wo)
    out6 = c6(a0, images, H, W, a1=a1, bias=bias, resid=resid, ln=ln, Ho=Ho, Wo=Wo)
    torch.cuda.synchronize()
    err = float((out3 - out6).abs().max() / out6.abs().max())
    out3b = c3(a0, images, H, W, a1=a1, bias=bias, resid=resid, ln=ln, Ho=Ho, Wo=Wo)
    same = bool(torch.equal(out3, out3b))
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c3(a0, images, H, W, a1=a1, bias=bias, resid=resid, out=out3, ln=ln, Ho=Ho, Wo=Wo)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    M = images * Ho * Wo
    flop = 2.0 * M * N * K * k * k
    byts = 4.0 * (M * K + M * N * (2 if res is True else 1) + K * k * k * N)
    print(f"{name:34s} M={M:7d} K={K * k * k:5d} N={N:4d}  {us:8.1f} us  {flop / us * 1e-6:7.1f} TF/s  {byts / us * 1e-6:6.2f} TB/s  "
          f"err {err:.2e}  repeatable {same}", flush=True)
