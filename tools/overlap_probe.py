"""Does a batch of smoke PDE rollouts (64 persistent workgroups, ~2.1 s) run CONCURRENTLY with guided sampling steps on another stream?
Times (a) 64 rollouts alone, (b) N sampling steps alone, (c) both enqueued back to back on two streams, with and without the CU budget
of the persistent kernels (include/dpc.h: dpc_set_cu_budget), sampling on the default stream and on a second pool stream.
    gpurun -- 'python tools/overlap_probe.py > gpurun_out/overlap_probe.log'"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench as Bn  # noqa: E402
from diffphycon_amd import _lib  # noqa: E402
from diffphycon_amd.dataset.apps import evaluate_solver as E  # noqa: E402
from diffphycon_amd.diffusion.diffusion_2d_smoke import SmokeGuidance  # noqa: E402

dev = torch.device("cuda:0")
B, STEPS = 64, 8
gd, _ = Bn.build_models(dev, 32)
guide = SmokeGuidance((2.0, 18.0, 20.0, 16.0, 20.0, 1.0), 0.0)
gd.noise_seed, gd.traj_offset = 0, 0
init = Bn.synthetic_init(B, 0).to(dev)
x = gd.sample_noise([B, 32, 6, 64, 64], dev)
x[:, 0, 0] = init
rng = np.random.default_rng(0)
c1 = torch.from_numpy((rng.standard_normal((B, 32, 64, 64)) * 0.3).astype(np.float32)).to(dev)
c2 = torch.from_numpy((rng.standard_normal((B, 32, 64, 64)) * 0.3).astype(np.float32)).to(dev)
c1[:, :, 8:56, 8:56] = 0
c2[:, :, 8:56, 8:56] = 0
d0 = (Bn.synthetic_init(B, 0) * 2).to(dev)
sim = E.init_sim_128()
v0 = E.init_velocity_()
kw = dict(frame_stride=8, space_stride=2, density_dtype=torch.float32)
L = _lib.lib()


def rollouts():
    return E.solver_batch(sim, v0, d0, c1, c2, 256, **kw)


def sampling(n=STEPS):
    t = 999
    for _ in range(n):
        gd.p_sample(None, x, t, design_fn=guide, design_guidance="standard", init=init)
        t -= 1


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


rollouts()
sampling(2)
print(f"rollouts alone: {wall(rollouts):.3f} s    {STEPS} sampling steps alone: {wall(sampling):.3f} s")
side, main2 = torch.cuda.Stream(), torch.cuda.Stream()
for budget in (0, 192):
    for on_pool in (False, True):
        def both():
            with torch.cuda.stream(side):
                keep = rollouts()
            L.dpc_set_cu_budget(budget)
            if on_pool:
                main2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(main2):
                    sampling()
            else:
                sampling()
            return keep
        t = wall(both)
        L.dpc_set_cu_budget(0)
        print(f"both, CU budget {budget:3d}, sampling on {'a pool stream' if on_pool else 'the default stream'}: {t:.3f} s")
print(f"again: rollouts alone {wall(rollouts):.3f} s, sampling alone {wall(sampling):.3f} s")
