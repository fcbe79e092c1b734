#!/usr/bin/env python
"""Per-operator run-to-run determinism, optionally (--load) while another process keeps the GPU busy with bench.py: each op is
repeated on fixed inputs and compared bit for bit with its first result.  (r02: this is how the ln_apply low-lane fault was
isolated -- see norm.hip.)    python tools/det_ops.py [reps] [--load]"""
import os, sys, subprocess, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib
from diffphycon_amd.model import surrogates_hip as SH
dev = torch.device("cuda", 0)
torch.manual_seed(0)
ctx = SH._Ctx(dev, 8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rows, Cc = 49152, 64
x = torch.randn(rows, Cc, device=dev); r = torch.randn(rows, Cc, device=dev); g = torch.randn(Cc, device=dev); be = torch.randn(Cc, device=dev)
dy = torch.randn(rows, Cc, device=dev)
st = ctx.ln_stats(x)
B, R = 12, 4096
gst = ctx.gn_stats(x, B, R, Cc)
ss = torch.randn(B, 2 * Cc, device=dev)
w1 = torch.randn(64, 64, 1, 1, device=dev) * 0.1
w3 = torch.randn(64, 64, 3, 3, device=dev) * 0.05
c1, c3 = SH._Conv(w1), SH._Conv(w3)
c1x = SH._Conv(w1, mode="x6")
wq = torch.randn(384, 64, 1, 1, device=dev) * 0.1
cq3, cq6, cq0 = SH._Conv(wq, mode="f16x3"), SH._Conv(wq, mode="x6"), SH._Conv(wq, mode="f32")
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
u3 = Unet3D_with_Conv3D(dim=64, out_dim=6, dim_mults=(1, 2, 4), channels=6).to(dev)
xs = torch.randn(2, 32, 6, 64, 64, device=dev)
ts = torch.tensor([5, 700], device=dev)
ops = {
    "ln_apply": lambda: ctx.ln_apply(x, st, g, r),
    "ln_stats": lambda: ctx.ln_stats(x),
    "ln_bwd": lambda: ctx.ln_bwd(x, st, g, dy),
    "gn_stats": lambda: ctx.gn_stats(x, B, R, Cc),
    "gn_apply": lambda: ctx.gn_apply(x, gst, g, be, ss, B, R, Cc, resid=r),
    "gn_bwd": lambda: ctx.gn_bwd(x, dy, gst, g, be, ss, B, R, Cc, True)[0],
    "add_": lambda: ctx.add_(x.clone(), r),
    "conv1x1_f16x3": lambda: c1(x, 12, 64, 64, resid=r),
    "conv1x1_x6": lambda: c1x(x, 12, 64, 64, resid=r),
    "conv3x3": lambda: c3(x, 12, 64, 64),
    "conv_ln_f16x3": lambda: cq3(x, 12, 64, 64, ln=(st, g)),
    "conv_ln_x6": lambda: cq6(x, 12, 64, 64, ln=(st, g)),
    "conv_ln_f32": lambda: cq0(x, 12, 64, 64, ln=(st, g)),
    "unet3d_fwd": lambda: u3(xs, ts),
    "torch_mul_add": lambda: x * 1.5 + r,
}
load = None
if "--load" in sys.argv:
    load = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--steps", "600"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(30)
for name, fn in ops.items():
    ref = fn().clone()
    bad, nel = 0, 0
    for _ in range(reps):
        o = fn()
        d = (o != ref)
        n = int(d.sum().item())
        if n:
            bad += 1
            nel += n
            if bad <= 2:
                idx = d.reshape(-1).nonzero().reshape(-1)
                print(f"   {name}: {n} elements, idx {idx[:8].tolist()} ref {ref.reshape(-1)[idx[:3]].tolist()} got {o.reshape(-1)[idx[:3]].tolist()}", flush=True)
    print(f"{name:16s} {bad:4d} of {reps} repetitions differ ({nel} elements)", flush=True)
if load:
    load.kill()
