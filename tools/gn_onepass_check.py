"""A/B of the one-launch GroupNorm (csrc/norm.hip: gn_onepass_kernel) against the three-launch path it replaces, through the C ABI
(dpc_groupnorm_silu_cl), on the Burgers U-Net's deep-level shapes: the outputs must be BIT-identical (same thread mapping, fp32 row
sums, fp64 folds, butterfly order).  The kernel choice is a debug switch read once per process, so each arm runs in its own process.
  gpurun -- 'python tools/gn_onepass_check.py'
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(256, 128, 256, 8, True), (256, 32, 512, 8, True), (256, 8, 1024, 8, False), (37, 100, 256, 8, True), (5, 8, 64, 8, False)]


def arm(path):
    import torch
    from diffphycon_amd import _lib as L
    dev = torch.device("cuda:0")
    outs, times = [], []
    for B, R, Cc, G, ss in SHAPES:
        g = torch.Generator().manual_seed(B * 1000 + R)
        x = (torch.randn(B, R, Cc, generator=g) * 2 + 0.3).to(dev)
        gamma, beta = torch.randn(Cc, generator=g).to(dev), torch.randn(Cc, generator=g).to(dev)
        sshift = torch.randn(B, 2 * Cc, generator=g).to(dev) if ss else None
        ws = L.workspace(L.lib().dpc_groupnorm_workspace_bytes(B, Cc), dev)
        y = x.clone()
        args = lambda t: (L.ptr(t), L.ptr(gamma), L.ptr(beta), L.ptr(sshift), B, R, Cc, G, C.c_void_p(ws.data_ptr()), ws.numel(), L.stream())
        L.check(L.lib().dpc_groupnorm_silu_cl(*args(y)))
        torch.cuda.synchronize()
        outs.append(y.cpu())
        z = x.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.check(L.lib().dpc_groupnorm_silu_cl(*args(z)))
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / 20)
    torch.save({"outs": outs, "us": times}, path)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        arm(sys.argv[1])
        sys.exit(0)
    import torch
    res = {}
    for name, val in (("onepass", "1"), ("three_launches", "0")):
        env = dict(os.environ, DPC_DEBUG="1", DPC_GN_ONEPASS=val)
        subprocess.run([sys.executable, os.path.abspath(__file__), f"/tmp/gn_{name}.pt"], check=True, env=env)
        res[name] = torch.load(f"/tmp/gn_{name}.pt")
    for i, (B, R, Cc, G, ss) in enumerate(SHAPES):
        a, b = res["onepass"]["outs"][i], res["three_launches"]["outs"][i]
        print(f"B={B:4d} R={R:4d} C={Cc:5d} scale_shift={ss}: bit-identical {bool(torch.equal(a, b))}   "
              f"one launch {res['onepass']['us'][i]:6.1f} us   three launches {res['three_launches']['us'][i]:6.1f} us")
