"""How far do the backward tensors of the jellyfish surrogates move inside the fp16 window that the ONE-TIME synthetic calibration
fixes (diffphycon_amd/model/surrogates_hip.py: _Calibration)?  The surrogates of the J128 configuration (dim 64, mults (1,2,4,8), 128 x
128 images, 20 frames) are calibrated on the seeded synthetic trajectory, then the design gradient is evaluated on inputs of very
different character and every backward convolution's max |input| * scale is read back (tool only: host reads per call):
    window: full 22-bit operands for |x| * scale in [2^-3, 65504]; calibrated maximum at 16
Prints per input family the smallest and largest (max |input| * scale) over the backward convolutions, i.e. the head-room to the clamp
and the distance to the calibration point.    gpurun -- 'python tools/calibration_drift.py > gpurun_out/calibration_drift.log'"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd.model import surrogates_2d as S2  # noqa: E402
from diffphycon_amd.model import surrogates_hip as SH  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(5)
fm = S2.ForceUnet(dim=64, out_dim=1, dim_mults=(1, 2, 4, 8), channels=4).to(dev).eval()
bd = S2.Unet(dim=64, out_dim=3, dim_mults=(1, 2, 4, 8), channels=3).to(dev).eval()
args = argparse.Namespace(only_vis_pressure=False, device=dev, reg_ratio=0.3, p_min=-1.7, p_max=2.3, image_size=128)
design = SH.HipDesignGradient(fm, bd, args)
design.check_every = 1
B, T, HW = 2, 20, 128
g = torch.Generator().manual_seed(1)
yy, xx = torch.meshgrid(torch.linspace(0, 1, HW), torch.linspace(0, 1, HW), indexing="ij")


def smooth(amp):
    f = sum(torch.sin(2 * 3.14159 * (k * xx + (k + 1) * yy) + k) / (k + 1) for k in range(4))
    return (amp * f / f.abs().max())[None, None, None].expand(B, T, 3, HW, HW)


families = {
    "uniform noise [-1, 1] (the calibration's own character)": lambda: torch.rand(B, T, 3, HW, HW, generator=g) * 2 - 1,
    "clipped gaussian, sigma 1 (x0 estimate at early steps)": lambda: torch.randn(B, T, 3, HW, HW, generator=g).clamp(-1, 1),
    "gaussian, sigma 0.05 (nearly constant fields)": lambda: torch.randn(B, T, 3, HW, HW, generator=g) * 0.05,
    "smooth fields, amplitude 1": lambda: smooth(1.0),
    "smooth fields, amplitude 0.1": lambda: smooth(0.1),
    "all zeros": lambda: torch.zeros(B, T, 3, HW, HW),
    "all ones (saturated pressure)": lambda: torch.ones(B, T, 3, HW, HW),
}
masks = {"random mask 30 %": lambda: (torch.rand(B, 1, 3, HW, HW, generator=g) > 0.7).float(),
         "empty boundary": lambda: torch.zeros(B, 1, 3, HW, HW),
         "disc": lambda: (((xx - 0.5) ** 2 + (yy - 0.5) ** 2) < 0.04).float()[None, None, None].expand(B, 1, 3, HW, HW)}
print(f"{'state input':58s} {'boundary':18s} {'min(max|x| scale)':>18s} {'max(max|x| scale)':>18s} {'head-room to clamp':>20s}")
for fname, fx in families.items():
    for mname, fmask in masks.items():
        for theta in (0.0, 0.8):
            x = torch.cat([fx(), torch.full((B, T, 1, HW, HW), theta)], dim=2).to(dev).contiguous()
            b0 = fmask().expand(-1, T, -1, -1, -1).contiguous().to(dev)
            design(x, b0)
            convs = [c for c in SH._Calibration.convs if getattr(c, "peak", None) is not None and c.act_scale > 0]
            vals = [float(c.peak.item()) * c.act_scale for c in convs]
            for c in convs:
                c.peak = None
            pos = [v for v in vals if v > 0]
            print(f"{fname:58s} {mname + f', theta {theta}':18s} {min(pos):18.3e} {max(pos):18.3e} {65504 / max(pos):20.1f}", flush=True)
print("calibration (synthetic input): max |input| per backward convolution from %.2e to %.2e over %d convolutions" % (
    min(m for m, _ in design.last_calibration if m > 0), max(m for m, _ in design.last_calibration), len(design.last_calibration)))
