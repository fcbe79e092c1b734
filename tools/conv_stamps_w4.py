"""Phase totals (s_memtime) of the Winograd F(4,3) conv kernel (csrc/conv3w4.hip): per MFMA wave the shader cycles in taps / accumulator
drain / chunk barrier / epilogue, per loader wave in request / halo wait / finish / deferred stores / chunk barrier / epilogue, summed
over one launch, and the effective shader clock.
    python tools/build_variant.py stamps -DDPC_CONV_STAMPS && DPC_LIB=diffphycon_amd/lib/libdpc_stamps.so python tools/conv_stamps_w4.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for (B, Fr, H, W, Ci, Co) in [(MB, 32, 64, 64, 64, 64), (MB, 32, 64, 64, 128, 64), (MB, 32, 32, 32, 128, 128), (MB, 32, 16, 16, 256, 256), (MB, 32, 16, 16, 512, 256)]:
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
    tiles = B * (Fr // 4) * (H // 8) * (W // 8) * (Co // 64) / nwg
    rec = out.flatten()[: nwg * 8 * 8].view(nwg, 8, 8).double().cpu()
    life = rec[:, :, 7].mean().item()
    print(f"{Ci}->{Co} @{H}: launch {us:.1f} us, wave lifetime {life:.0f} cycles = {life / us / 1e3:.3f} GHz; {tiles:.0f} tiles x {kch} chunks per "
          f"workgroup; ideal MFMA issue per chunk {162 * 32}")
    per_chunk = tiles * kch
    m = rec[:, :4, :].mean(0)
    for wv in range(4):
        v = m[wv]
        print(f"   MFMA wave {wv}: per chunk taps {v[0] / per_chunk:.0f}  drain {v[1] / per_chunk:.0f}  barrier {v[2] / per_chunk:.0f} | per tile "
              f"epilogue write {v[3] / tiles:.0f}  epilogue barriers {v[4] / tiles:.0f}")
    l = rec[:, 4:, :].mean(0)
    for wv in range(4):
        v = l[wv]
        print(f"   loader wave {wv}: per chunk request {v[0] / per_chunk:.0f}  halo wait {v[1] / per_chunk:.0f}  finish {v[2] / per_chunk:.0f}  "
              f"stores {v[3] / per_chunk:.0f}  barrier {v[4] / per_chunk:.0f} | per tile epilogue {v[5] / tiles:.0f}")
