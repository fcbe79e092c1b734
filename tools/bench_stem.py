"""Micro-benchmark of the 7x7x7 stem (dpc_stem_pack / dpc_stem_run) at the S64 shapes: joint denoiser C = 6, prior denoiser C = 2
(a channel slice of the 6-channel state), 64 x 64 x 32 frames.      python tools/bench_stem.py [reps] [micro-batch]
A/B of the channel-pair kernel against the slot forms: DPC_DEBUG=1 DPC_STEM_PAIRS=0 python tools/bench_stem.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
L = _lib.lib()
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(MB, 32, 6, 64, 64, device=dev, generator=g)
out = torch.empty(MB, 32, 64, 64, 64, device=dev)
for name, c_off, Cin in (("joint C=6", 0, 6), ("prior C=2", 3, 2), ("C=4", 0, 4)):
    w = torch.randn(64, Cin, 7, 7, 7, device=dev, generator=g) / (Cin * 343) ** 0.5
    b = torch.randn(64, device=dev, generator=g)
    h = C.c_void_p()
    _lib.check(L.dpc_stem_pack(_lib.ptr(w), 64, Cin, 7, b"", C.byref(h), _lib.stream()))
    run = lambda: _lib.check(L.dpc_stem_run(h, _lib.ptr(x), 6, c_off, _lib.ptr(b), _lib.ptr(out), MB, 32, 64, 64, _lib.stream()))
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * MB * 32 * 64 * 64 * 64 * 343 * Cin
    chk = int(out.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff
    print(f"{name}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s  chk {chk:012x}")
    L.dpc_stem_free(h)
