"""Pipeline-stage cycle stamps (s_memtime) of the big-tile conv kernel + the per-CU workgroup timeline.
Needs a library built with -DDPC_CONV_STAMPS (the stamps overwrite the start of the output):
    python tools/build_variant.py stamps -DDPC_CONV_STAMPS && DPC_LIB=diffphycon_amd/lib/libdpc_stamps.so python tools/conv_stamps.py"""
import ctypes as C
import os
import sys
from collections import defaultdict

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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.dpc_conv3d_cl(*args))
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) * 1e3
    wide = Co > 64
    tf = 4 if wide else 8
    nwg = B * (Fr // tf) * (H // 8) * (W // 8) * (Co // (128 if wide else 64))
    names = ["prologue"] + [f"{nm}{c}" for c in range(2) for nm in ("tap0_", "halo_issue", "taps1-11_", "prepare", "taps12-26_", "handover")]
    rec = out.flatten()[: nwg * 4 * 32].view(nwg * 4, 32)
    mean = rec[:, 1:len(names) + 1].double().mean(0).tolist()
    epi = rec[:, 15].double().mean().item()
    print(f"{Ci}->{Co} @{H}: ideal = {24 * 32} cycles/tap;", "  ".join(f"{nm} {v:.0f}" for nm, v in zip(names, mean)), f" epilogue {epi:.0f}")
    raw = rec[:, 16:22].contiguous().view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff
    raw = raw[0::4]                                                  # wave 0 of every workgroup
    start = raw[:, 0] | (raw[:, 1] << 32)
    end = raw[:, 2] | (raw[:, 3] << 32)
    cu = (raw[:, 5] << 32) | (raw[:, 4] & 0xffffff00)                # (XCC_ID, HW_ID without wave / SIMD bits)
    per = defaultdict(list)
    for s_, e_, c_ in zip(start, end, cu):
        per[int(c_)].append((int(s_), int(e_)))
    gaps, durs = [], []
    for v in per.values():
        v.sort()
        durs += [e_ - s_ for s_, e_ in v]
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    t0, t1 = int(start.min()), int(end.max())
    import statistics as st
    print(f"   launch {launch_us:.1f} us by events; {nwg} workgroups on {len(per)} CUs ({nwg / max(len(per), 1):.2f} per CU); "
          f"first start -> last end {t1 - t0} ticks = {(t1 - t0) / launch_us:.1f} ticks/us; workgroup duration mean "
          f"{st.mean(durs):.0f} ticks, gap between consecutive workgroups on a CU mean {st.mean(gaps) if gaps else 0:.0f} "
          f"median {st.median(gaps) if gaps else 0:.0f} max {max(gaps) if gaps else 0} ticks")
