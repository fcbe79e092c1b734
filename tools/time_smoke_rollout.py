"""Times the smoke PDE evaluator (row D): B trajectories x 256 frames on one MI355X, and the CPU oracle on a bounded
sample (a few frames) for the baseline figure.   python tools/time_smoke_rollout.py [B] [T]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffphycon_amd.dataset.apps import evaluate_solver as E  # noqa: E402
from diffphycon_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
c1 = (rng.standard_normal((B, 32, 64, 64)) * 0.3).astype(np.float32)
c2 = (rng.standard_normal((B, 32, 64, 64)) * 0.3).astype(np.float32)
c1[:, :, 8:56, 8:56] = 0
c2[:, :, 8:56, 8:56] = 0
d0 = np.zeros((B, 64, 64), np.float32)
for b in range(B):
    r, c = rng.integers(10, 26), rng.integers(12, 53)
    d0[b, r:r + 5, c:c + 5] = 1
sim = E.init_sim_128()
c1d, c2d, d0d = (torch.from_numpy(a).to(dev) for a in (c1, c2, d0))
res = {}
for name, kw in (("multi_evaluate outputs (::8 frames, ::2 space, f32 density)",
                  dict(frame_stride=8, space_stride=2, density_dtype=torch.float32)),
                 ("reference outputs (all frames, 128^2, f64)", dict())):
    E.solver_batch(sim, E.init_velocity_(), d0d[:2], c1d[:2, :2], c2d[:2, :2], 2, **kw)          # warm-up / module load
    torch.cuda.synchronize()
    _lib.profile_begin()
    t0 = time.perf_counter()
    out = E.solver_batch(sim, E.init_velocity_(), d0d, c1d, c2d, T, return_cg_iterations=True, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = _lib.profile_end()
    its = out[4].double()
    res[name] = {"seconds": dt, "rollouts_per_s": B / dt, "kernel_ms": prof.get("smoke_eval", {}).get("total_ms"),
                 "mean_cg_iterations": its.mean().item(), "us_per_cg_iteration": dt * 1e6 / its.sum().item() * B,
                 "smoke_out_last_mean": out[3][:, -1].mean().item()}
# CPU oracle on 3 frames of one trajectory
from oracle import smoke_solver as O  # noqa: E402
dom = O.init_sim_128()
t0 = time.perf_counter()
its = []
O.solver(dom, O.init_velocity_(), d0[0], c1[0, :1], c2[0, :1], per_timelength=4, info=its)
cpu = time.perf_counter() - t0
res["cpu_oracle"] = {"frames": 3, "seconds": cpu, "s_per_frame": cpu / 3, "extrapolated_s_per_256_frame_rollout": cpu / 3 * 255,
                     "cg_iterations": its}
print(json.dumps({"B": B, "T": T, **res}, indent=1))
