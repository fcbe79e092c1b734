#!/usr/bin/env python
"""Per-operator determinism while a SECOND STREAM of the same process keeps the GPU busy (the in-process form of tools/det_ops.py --load;
r04: tests/test_gpu_unet3d.py::test_two_denoisers_on_two_streams_equal_the_serial_forwards failed intermittently and
tools/two_stream_bisect.py localised the first differing tap to the temporal-attention block of a dim-32 net).  Each op of that block
(and the others of a forward) is repeated on the main stream while a U-Net forward loop runs on a side stream; outputs are compared bit
for bit with the op's solo result.    gpurun -- 'python tools/det_ops2.py [reps] [C]'"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib  # noqa: E402
from diffphycon_amd.model import surrogates_hip as SH  # noqa: E402
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D, _rotary_tables  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
torch.manual_seed(0)
L = _lib.lib()
ctx = SH._Ctx(dev, 8)
B, Fr, HW = 4, 8, 32 * 32
rows = B * Fr * HW
x = torch.randn(rows, Cc, device=dev)
r = torch.randn(rows, Cc, device=dev)
g = torch.randn(Cc, device=dev)
be = torch.randn(Cc, device=dev)
st = ctx.ln_stats(x)
gst = ctx.gn_stats(x, B, Fr * HW, Cc)
ss = torch.randn(B, 2 * Cc, device=dev)
wq = torch.randn(384, Cc, 1, 1, device=dev) * 0.1
wo = torch.randn(Cc, 128, 1, 1, device=dev) * 0.1
w1 = torch.randn(Cc, Cc, 1, 1, device=dev) * 0.1
cq, co, c1 = SH._Conv(wq), SH._Conv(wo), SH._Conv(w1)
qkv = torch.randn(rows, 384, device=dev)
att = torch.randn(rows, 128, device=dev)
cos, sin = (t.to(dev).contiguous() for t in _rotary_tables(Fr, 32))
bias = torch.randn(4, Fr, Fr, device=dev)


def attn_temporal():
    out = torch.empty(rows, 128, device=dev)
    _lib.check(L.dpc_attention_core(_lib.ptr(qkv), _lib.ptr(out), 4, Fr, B * HW, HW, Fr * HW, 1, HW, _lib.ptr(cos), _lib.ptr(sin),
                                    _lib.ptr(bias), _lib.stream()))
    return out


def attn_spatial():                       # whole-image sequences (the mid block's form): L = 256 tokens
    out = torch.empty(rows, 128, device=dev)
    _lib.check(L.dpc_attention_core(_lib.ptr(qkv), _lib.ptr(out), 4, 256, rows // 256, 1, 256, 0, 1, None, None, None, _lib.stream()))
    return out


def conv3d(ci, co):
    xx = torch.randn(B, Fr, 32, 32, ci, device=dev)
    w = torch.randn(co, ci, 3, 3, 3, device=dev) / (ci * 27) ** 0.5
    b = torch.randn(co, device=dev)
    out = torch.empty(B, Fr, 32, 32, co, device=dev)
    ws = _lib.workspace(L.dpc_conv_workspace_bytes(ci, co, 27) * 4, dev)

    def run():
        _lib.check(L.dpc_conv3d_cl(_lib.ptr(xx), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, Fr, 32, 32, ci, co, 3, 3, 3, 1, 1, 1, 1, 1, 1,
                                   C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream()))
        return out
    return run


ops = {
    "ln_stats": lambda: ctx.ln_stats(x),
    "ln_apply": lambda: ctx.ln_apply(x, st, g, r),
    "qkv 1x1 behind LN": lambda: cq(x, B * Fr, 32, 32, ln=(st, g)),
    "attention core, temporal (rotary + bias)": attn_temporal,
    "attention core, spatial L = 256": attn_spatial,
    "to_out 1x1 128 -> C + residual": lambda: co(att, B * Fr, 32, 32, resid=r),
    "1x1 C -> C + residual": lambda: c1(x, B * Fr, 32, 32, resid=r),
    "gn_stats": lambda: ctx.gn_stats(x, B, Fr * HW, Cc),
    "gn_apply": lambda: ctx.gn_apply(x, gst, g, be, ss, B, Fr * HW, Cc, resid=r),
    "conv3x3x3 C -> C": conv3d(Cc, Cc),
    "torch mul-add": lambda: x * 1.5 + r,
}
# ---- who is the aggressor?  DET_VICTIM=<substring of an op name>: that op alone is repeated on the main stream while ONE candidate at a
# time loops on the side stream (every op above, a ConvTranspose, the linear-attention core, the 7x7x7 stem, and whole U-Net forwards)
def convT():
    xx = torch.randn(B * Fr * 32 * 32, 64, device=dev)
    w = torch.randn(64, 64, 1, 4, 4, device=dev) * 0.05
    b = torch.randn(64, device=dev)
    out = torch.empty(B * Fr * 64 * 64, 64, device=dev)
    ws = _lib.workspace(L.dpc_conv_workspace_bytes(64, 64, 16) * 4, dev)

    def run():
        _lib.check(L.dpc_convtranspose3d_144_cl(_lib.ptr(xx), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, Fr, 32, 32, 64, 64,
                                                C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream()))
        return out
    return run


def stem(ci):
    xx = torch.randn(B, Fr, 32, 32, ci, device=dev)
    w = torch.randn(32, ci, 7, 7, 7, device=dev) * 0.02
    b = torch.randn(32, device=dev)
    out = torch.empty(B, Fr, 32, 32, 32, device=dev)
    ws = _lib.workspace(L.dpc_conv_workspace_bytes(8, 64, 343) * 4, dev)

    def run():
        _lib.check(L.dpc_conv3d_cl(_lib.ptr(xx), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, Fr, 32, 32, ci, 32, 7, 7, 7, 1, 1, 3, 3, 3, 3,
                                   C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream()))
        return out
    return run


def linattn():
    out = torch.empty(rows, 128, device=dev)
    ws = _lib.workspace(L.dpc_linear_attention_workspace_bytes(B * Fr, 4), dev)

    def run():
        _lib.check(L.dpc_linear_attention_core(_lib.ptr(qkv), _lib.ptr(out), 4, B * Fr, 32 * 32, C.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream()))
        return out
    return run


def unet(dim, mults):
    net = Unet3D_with_Conv3D(dim=dim, dim_mults=mults, channels=6).to(dev)
    xs_ = torch.randn(4, 8, 6, 32, 32, device=dev)
    ts_ = torch.tensor([5, 700, 3, 900], device=dev)
    return lambda: net(xs_, ts_)


victim_key = os.environ.get("DET_VICTIM")
side = torch.cuda.Stream()
if victim_key:
    vname = [k for k in ops if victim_key in k][0]
    victim = ops[vname]
    aggressors = dict(ops)
    aggressors.update({"conv3x3x3 64 -> 64 (Winograd)": conv3d(64, 64), "ConvTranspose (1,4,4) 64 -> 64": convT(), "linear attention core": linattn(), "U-Net dim 32 (1,2)": unet(32, (1, 2)),
                       "U-Net dim 64 (1,2)": unet(64, (1, 2))})
    ref = victim().clone()
    torch.cuda.synchronize()
    print(f"victim: {vname} (C = {Cc})", flush=True)
    for aname, afn in aggressors.items():
        afn()
        torch.cuda.synchronize()
        bad = 0
        for i in range(reps):
            if side.query():
                with torch.cuda.stream(side):
                    for _ in range(30):
                        afn()
            bad += int(not torch.equal(victim(), ref))
        side.synchronize()
        print(f"  beside {aname:44s} {bad:4d} of {reps} repetitions of the victim differ", flush=True)
    sys.exit(0)

noise_net = Unet3D_with_Conv3D(dim=32, dim_mults=(1, 2), channels=6).to(dev)
xs = torch.randn(4, 8, 6, 32, 32, device=dev)
ts = torch.tensor([5, 700, 3, 900], device=dev)
noise_net(xs, ts)
torch.cuda.synchronize()


def feed_noise(n=40):
    with torch.cuda.stream(side):
        for _ in range(n):
            noise_net(xs, ts)


for name, fn in ops.items():
    ref = fn().clone()
    torch.cuda.synchronize()
    solo = sum(int(not torch.equal(fn(), ref)) for _ in range(20))
    bad, nel = 0, 0
    for i in range(reps):
        if side.query():
            feed_noise()
        o = fn()
        d = (o != ref)
        n = int(d.sum().item())
        if n:
            bad += 1
            nel += n
            if bad <= 2:
                idx = d.reshape(-1).nonzero().reshape(-1)
                print(f"   {name}: {n} elements differ, flat idx {idx[:6].tolist()} ... {idx[-2:].tolist()}; ref {ref.reshape(-1)[idx[:2]].tolist()} "
                      f"got {o.reshape(-1)[idx[:2]].tolist()}", flush=True)
    side.synchronize()
    print(f"{name:44s} solo {solo:2d} of 20 | under load {bad:4d} of {reps} repetitions differ ({nel} elements)", flush=True)
