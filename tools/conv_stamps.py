"""Pipeline-stage cycle stamps (s_memtime) of the big-tile conv kernel: mean cycles per stage over all waves.
Needs a library built with -DDPC_CONV_STAMPS (the stamps overwrite the start of the output):
    DPC_EXTRA_FLAGS=-DDPC_CONV_STAMPS python -m diffphycon_amd.build && python tools/conv_stamps.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
for (B, Fr, H, W, Ci, Co) in [(8, 32, 64, 64, 64, 64), (8, 32, 32, 32, 128, 128), (8, 32, 16, 16, 256, 256)]:
    x = torch.randn(B, Fr, H, W, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3, 3, device=dev) / (Ci * 27) ** 0.5
    b = torch.randn(Co, device=dev)
    out = torch.empty(B, Fr, H, W, Co, device=dev)
    ws = _lib.workspace(L.dpc_conv_workspace_bytes(Ci, Co, 27) * 4, dev)
    args = (_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, Fr, H, W, Ci, Co, 3, 3, 3, 1, 1, 1, 1, 1, 1,
            C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream())
    for _ in range(2):
        _lib.check(L.dpc_conv3d_cl(*args))
    torch.cuda.synchronize()
    wide = Co > 64
    tf = 4 if wide else 8
    nwg = B * (Fr // tf) * (H // 8) * (W // 8) * (Co // (128 if wide else 64))
    names = ["prologue"] + [f"{nm}{c}" for c in range(2) for nm in ("tap0_", "halo_issue", "taps1-11_", "prepare", "taps12-26_", "handover")]
    n = len(names) + 1
    t = out.flatten()[: nwg * 4 * 16].view(nwg * 4, 16)[:, 1:n].double()
    mean = t.mean(0).tolist()
    print(f"{Ci}->{Co} @{H}: ideal = {24 * 32} cycles/tap;", "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, mean)))
