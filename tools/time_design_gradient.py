#!/usr/bin/env python
"""Time of the jellyfish design gradient (force_fn: boundary updater forward -> ForceUnet forward -> both backward) at the
shape inference_2d_jellyfish.py runs (64 x 64, 20 frames, dim 64, mults (1,2,4,8)), on libdpc (surrogates_hip.py) and on the
torch modules + autograd, with the per-kernel-class split of the HIP path.
    python tools/time_design_gradient.py [batch] [--no-torch]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))       # torch_force_fn: the torch-autograd reference lives with the tests
from diffphycon_amd import _lib  # noqa: E402
from diffphycon_amd.diffusion import diffusion_2d_jellyfish as DJ  # noqa: E402
from diffphycon_amd.model import surrogates_2d as S2  # noqa: E402
from diffphycon_amd.model import surrogates_hip as SH  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
T, HW = 20, 64
dev = torch.device("cuda", 0)
torch.manual_seed(0)
fm = S2.ForceUnet(dim=64, out_dim=1, dim_mults=(1, 2, 4, 8), channels=4).to(dev).eval()
bd = S2.Unet(dim=64, out_dim=3, dim_mults=(1, 2, 4, 8), channels=3).to(dev).eval()
for q in list(fm.parameters()) + list(bd.parameters()):
    q.requires_grad_(False)
args = argparse.Namespace(only_vis_pressure=False, device=dev, reg_ratio=1000.0, p_min=-1.7, p_max=2.3, image_size=HW)
x = torch.rand(B, T, 4, HW, HW, device=dev) * 2 - 1
bd0e = torch.rand(B, 1, 3, HW, HW, device=dev).expand(-1, T, -1, -1, -1).contiguous()
design = SH.HipDesignGradient(fm, bd, args)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


if "--trace" in sys.argv:            # under rocprofv3: exactly two design-gradient calls (the first one calibrates), nothing else
    design(x, bd0e)
    design(x, bd0e)
    torch.cuda.synchronize()
    sys.exit(0)
ms_hip, g_hip = timed(lambda: design(x, bd0e), 5)
print(f"design gradient, batch {B} x {T} frames x {HW}^2: libdpc {ms_hip:.1f} ms")
th = torch.rand(B * T, device=dev)
ms_fwd, _ = timed(lambda: design.unet(bd0e.reshape(-1, 3, HW, HW), th), 5)
print(f"boundary updater forward alone: {ms_fwd:.1f} ms")
if "--no-torch" not in sys.argv:
    def torch_design():
        gs, gt = force_fn(x.clone(), bd0e, fm, bd, args)
        return torch.cat([gs, gt.unsqueeze(2)], dim=2)
    ms_t, g_t = timed(torch_design, 3)
    err = [((g_hip[:, :, c] - g_t[:, :, c]).abs().max() / g_t[:, :, c].abs().max()).item() for c in (2, 3)]
    print(f"torch autograd {ms_t:.1f} ms; relative difference pressure {err[0]:.2e} theta {err[1]:.2e}")
_lib.profile_begin()
design(x, bd0e)
rows = _lib.profile_end()
tot = sum(r["total_ms"] for r in rows.values())
for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"]):
    if r["launches"]:
        ms = max(r["total_ms"], 1e-9)
        print(f"  {name:14s} {r['launches']:5d} launches {r['total_ms']:8.2f} ms  {r['flops'] / ms / 1e9:8.1f} TFLOP/s {r['bytes'] / ms / 1e6:8.1f} GB/s")
print(f"  profiled classes total {tot:.1f} ms (launches not in a class: the norms' backward, layout glue)")
cal = design.last_calibration
if cal:
    ms = [m for m, _ in cal if m > 0]
    print(f"backward calibration: {len(cal)} convolutions, max|input| from {min(ms):.3e} to {max(ms):.3e}")
