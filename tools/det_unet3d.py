#!/usr/bin/env python
"""Run-to-run determinism of the 3-D denoiser forward (the Winograd convolution, fused attention, implicit GEMMs) at the S64 micro-batch
extent, optionally (--load) while a second process keeps the GPU busy with bench.py: the same forward is repeated and compared bit
for bit with its first result.    python tools/det_unet3d.py [reps] [--load]"""
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6, micro_batch=4).to(dev)
x = torch.randn(4, 32, 6, 64, 64, device=dev)      # [B, F, C, H, W]
t = torch.tensor([999, 500, 10, 0], device=dev)
load = None
if "--load" in sys.argv:
    load = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "1", "--no-cpu-baseline",
                             "--no-extras"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(25)
with torch.no_grad():
    ref = m(x, t).clone()
    bad = 0
    for i in range(reps):
        y = m(x, t)
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            print(f"rep {i}: {int((y != ref).sum())} elements differ, max |diff| {d.max().item():.3e}", flush=True)
torch.cuda.synchronize()
print(f"{reps} repetitions, {bad} differed from the first; load process {'running' if load and load.poll() is None else 'none / finished'}")
if load:
    load.kill()
sys.exit(1 if bad else 0)
