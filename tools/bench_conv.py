"""Micro-benchmark of the 3x3x3 convolution kernels at the shapes of the S64 U-Net (micro-batch 8).
    python tools/bench_conv.py [reps]      env DPC_CONV_MODE=f32|x6, DPC_CONV3X6_BDIRECT=0|1"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 8          # micro-batch (bench.py runs 32 since r03)
dev = torch.device("cuda:0")
L = _lib.lib()
SHAPES = [  # B, F, H, W, Cin, Cout   (count per U-Net forward in parentheses)
    (MB, 32, 64, 64, 64, 64),      # level 0 blocks (x7)
    (MB, 32, 64, 64, 128, 64),     # ups level 0 block1 with concat / final (x2)
    (MB, 32, 32, 32, 64, 128),     # level 1 first block
    (MB, 32, 32, 32, 128, 128),    # level 1 (x5)
    (MB, 32, 32, 32, 256, 128),    # ups level 1 concat
    (MB, 32, 16, 16, 128, 256),    # level 2 first
    (MB, 32, 16, 16, 256, 256),    # level 2 / mid (x7)
    (MB, 32, 16, 16, 512, 256),    # ups level 2 concat
]
tot = 0.0
for (B, Fr, H, W, Ci, Co) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(Ci * 1000 + Co + H)
    x = torch.randn(B, Fr, H, W, Ci, device=dev, generator=g)
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev, generator=g) / (Ci * 27) ** 0.5
    b = torch.randn(Co, device=dev, generator=g)
    out = torch.empty(B, Fr, H, W, Co, device=dev)
    ws = _lib.workspace(L.dpc_conv_workspace_bytes(Ci, Co, 27) * 4, dev)
    args = (_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, Fr, H, W, Ci, Co, 3, 3, 3, 1, 1, 1, 1, 1, 1,
            C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream())
    _lib.check(L.dpc_conv3d_cl(*args))
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(reps):
        _lib.check(L.dpc_conv3d_cl(*args))
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    k = [v for n, v in prof.items() if n.startswith("conv3")][0]
    ms = k["total_ms"] / k["launches"]
    fl = 2.0 * B * Fr * H * W * Co * 27 * Ci
    chk = int(out.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff      # bit-level checksum of the output
    print(f"{B}x{Fr}x{H}x{W} {Ci:4d}->{Co:4d}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s  chk {chk:012x}")
    tot += ms
print(f"sum {tot:.3f} ms")
