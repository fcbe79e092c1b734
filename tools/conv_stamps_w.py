"""Cycle stamps (s_memtime) of the Winograd conv kernel (conv3w.hip): for the first two tiles of every persistent workgroup, per
MFMA wave the time in taps / barrier waits / the three epilogue phases of each frame pair, per loader wave the time to produce a
chunk and its wait at the barrier; and the effective shader clock of the launch.
    python tools/build_variant.py stamps -DDPC_CONV_STAMPS && DPC_LIB=diffphycon_amd/lib/libdpc_stamps.so python tools/conv_stamps_w.py"""
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
    for _ in range(3):
        _lib.check(L.dpc_conv3d_cl(*args))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.dpc_conv3d_cl(*args))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nwg = 256
    kch = Ci // 16
    rec = out.flatten()[: nwg * 8 * 32].view(nwg, 8, 32)
    raw = rec[:, :, 28:32].contiguous().view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff
    dur = ((raw[..., 2] | (raw[..., 3] << 32)) - (raw[..., 0] | (raw[..., 1] << 32)))
    mf = rec[:, :4, :27].double().mean((0, 1)).tolist()
    ld = rec[:, 4:, :27].double().mean(0)        # per loader wave
    names = ["first_barrier"]
    for j in range(2):
        for kc in range(kch):
            names += [f"t{j}c{kc}_taps", f"t{j}c{kc}_wait"]
        for pr in range(2):
            names += [f"t{j}p{pr}_write+bar", f"t{j}p{pr}_loader_reads"]
    names = names[:27]
    print(f"{Ci}->{Co} @{H}: launch {us:.1f} us; MFMA wave lifetime mean {dur[:, :4].mean():.0f} cycles -> {dur[:, :4].mean() / us / 1e3:.3f} GHz "
          f"effective; ideal taps/chunk = {9 * 24 * 32}; tiles per workgroup {B * (Fr // 4) * (H // 8) * (W // 8) * (Co // 64) / 256:.1f}")
    print("   MFMA: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(names, mf)))
    lnames = ["first_produce+bar"]
    for s in range(kch):
        lnames += [f"s{s}_request", f"s{s}_landed", f"s{s}_finish", f"s{s}_wait"]
    lnames += ["ep_barriers+gather", "ep_emit0", "ep_emit1"]
    for s in range(kch, 2 * kch):
        lnames += [f"s{s}_request", f"s{s}_landed", f"s{s}_finish", f"s{s}_wait"]
    for wv in range(4):
        print(f"   loader wave {wv}: " + "  ".join(f"{n} {v:.0f}" for n, v in zip(lnames[:27], ld[wv].tolist())))
