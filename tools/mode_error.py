#!/usr/bin/env python
"""Deviation of the HIP U-Net forward from the reference fixtures (tests/golden/unet3d_*.npz) under the current
DPC_CONV_MODE / DPC_IGEMM_MODE / DPC_ATTN_MODE: prints max |y - y_ref| / max |y_ref| per fixture.
    for m in f32 x6 f16x3; do DPC_CONV_MODE=$m python tools/mode_error.py; done
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D  # noqa: E402

dev = torch.device("cuda:0")
out = []
for tag in ("joint", "w", "wide"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"unet3d_{tag}.npz"))
    m = Unet3D_with_Conv3D(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=int(g["channels"]))
    m.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")})
    m = m.to(dev)
    y = m(torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["t"]).to(dev)).cpu()
    ref = torch.from_numpy(g["y"])
    out.append(f"{tag} {((y - ref).abs().max() / ref.abs().max()).item():.2e}")
print("conv=%s igemm=%s attn=%s :" % tuple(os.environ.get(k, "default") for k in ("DPC_CONV_MODE", "DPC_IGEMM_MODE", "DPC_ATTN_MODE")),
      "  ".join(out))
