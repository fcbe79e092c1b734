"""Does an operator write outside the buffers it was given?  Output / workspace tensors are allocated with guard zones (a bit pattern in
front of and behind the part handed to the library) and the zones are compared after the call.    gpurun -- 'python tools/guard_check.py'"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd import _lib  # noqa: E402
from diffphycon_amd.model import surrogates_hip as SH  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
G = 1 << 20            # guard floats on either side
PAT = 0x7fc0dead


def guarded(n_floats):
    t = torch.empty(n_floats + 2 * G, device=dev, dtype=torch.float32)
    t.view(torch.int32).fill_(PAT)
    return t, t[G:G + n_floats]


def check(name, t):
    i = t.view(torch.int32)
    lo, hi = int((i[:G] != PAT).sum()), int((i[-G:] != PAT).sum())
    if hi:
        idx = (i[-G:] != PAT).nonzero().reshape(-1)
        print(f"{name}: {hi} floats BEHIND the buffer were overwritten (first at +{int(idx[0])}, last at +{int(idx[-1])})")
    if lo:
        print(f"{name}: {lo} floats IN FRONT of the buffer were overwritten")
    if not lo and not hi:
        print(f"{name}: guards intact")


B, Fr, H, W = 4, 8, 32, 32
for ci, co in ((64, 64), (128, 128)):
    xx = torch.randn(B * Fr * H * W, ci, device=dev)
    w = torch.randn(ci, co, 1, 4, 4, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    gout, out = guarded(B * Fr * 4 * H * W * co)
    nws = 4 * L.dpc_conv_workspace_bytes(ci, co, 4) + 256
    gws, ws = guarded(nws // 4 + 64)
    _lib.check(L.dpc_convtranspose3d_144_cl(_lib.ptr(xx), _lib.ptr(w), _lib.ptr(b), _lib.ptr(out), B, Fr, H, W, ci, co,
                                            C.c_void_p(ws.data_ptr()), ws.numel() * 4, _lib.stream()))
    torch.cuda.synchronize()
    check(f"ConvTranspose {ci}->{co} output", gout)
    check(f"ConvTranspose {ci}->{co} workspace ({nws} bytes = 4 x dpc_conv_workspace_bytes + 256)", gws)
    print("   output finite:", bool(torch.isfinite(out).all()))
# the same through the operator handle the surrogates / tests use (pre-packed weights, parity scatter)
wt = torch.randn(64, 64, 2, 2, device=dev) * 0.1
conv = SH._Conv(wt, ph=1, pw=1)
x0 = torch.randn(B * Fr * H * W, 64, device=dev)
gout, out = guarded(B * Fr * 4 * H * W * 64)
conv(x0, B * Fr, H, W, out=out.view(-1, 64), Ho=H, Wo=W, out_mode=2, par=(1, 1))
torch.cuda.synchronize()
check("2x2-tap parity class (1,1) through dpc_conv_run, out_mode 2", gout)
# and the victim of tools/det_ops2.py: the K = 32 qkv projection behind a LayerNorm
ctx = SH._Ctx(dev, 8)
x = torch.randn(B * Fr * H * W, 32, device=dev)
st = ctx.ln_stats(x)
g = torch.randn(32, device=dev)
cq = SH._Conv(torch.randn(384, 32, 1, 1, device=dev) * 0.1)
gout, out = guarded(B * Fr * H * W * 384)
cq(x, B * Fr, H, W, ln=(st, g), out=out.view(-1, 384))
torch.cuda.synchronize()
check("qkv 1x1 behind LN (K = 32)", gout)
