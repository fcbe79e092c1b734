"""ORACLE (test infrastructure, not product): CPU restatement of the Burgers finite-difference evaluator.

Follows /root/reference/dataset/apps/generate_burgers.py:207-299 (`burgers_numeric_solve_free`) with the
difference stencils of `Diff_mat_1D` :95-110, in NumPy fp32 with one rounding per arithmetic op (NumPy never
contracts to FMA), vectorised over trajectories and cells.
Pinned against the reference on tests/golden/burgers_fd.npz (tests/test_oracle_burgers.py).
"""
import math

import numpy as np


def burgers_numeric_solve_free(u0, f, visc, T, dt=1e-4, num_t=10):
    """u0 [N,s] fp32, f [N,num_t,s] fp32 -> trajectory [N,num_t+1,s] fp32 (u0 prepended)."""
    u0 = np.ascontiguousarray(u0, dtype=np.float32)
    f = np.ascontiguousarray(f, dtype=np.float32)
    n, s = u0.shape
    assert f.shape == (n, num_t, s)
    dx = 1.0 / (s + 1)                                   # :240
    steps = math.ceil(T / dt)                            # :243
    record = math.floor(steps / num_t)                   # :251
    t_l, t_r = np.float32(-1.0 / (2 * dx)), np.float32(1.0 / (2 * dx))                    # :265
    d_l = np.float32(visc * 1.0 / dx ** 2)               # :267
    d_c = np.float32(visc * -2.0 / dx ** 2)
    d_r = d_l
    dtf = np.float32(dt)
    mhalf = np.float32(-0.5)
    u = np.zeros((n, s + 2), dtype=np.float32)
    u[:, 1:-1] = u0
    sol = np.zeros((n, num_t, s), dtype=np.float32)
    f_idx, c = -1, 0
    for j in range(steps):
        u[:, 0] = 0                                      # :278-279 (re-pad with zero Dirichlet cells)
        u[:, -1] = 0
        us = u * u
        transport = us[:, :-2] * t_l + us[:, 2:] * t_r
        diffusion = (u[:, :-2] * d_l + u[:, 1:-1] * d_c) + u[:, 2:] * d_r
        if j % record == 0:                              # :284-285
            f_idx += 1
        rhs = (mhalf * transport + diffusion) + f[:, f_idx, :]
        u[:, 1:-1] = u[:, 1:-1] + dtf * rhs              # :286
        if (j + 1) % record == 0 and c < num_t:          # :291-295
            sol[:, c, :] = u[:, 1:-1]
            c += 1
    return np.concatenate((u0[:, None, :], sol), axis=1)


def synthetic_inputs(n, s=128, num_t=10, seed=0):
    """Two-Gaussian initial state and forcing in the style of make_data_varying_f (generate_burgers.py:361-372)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = (np.arange(s) + 1.0) / (s + 1)

    def bumps(shape_prefix, amp_scale):
        loc1 = rng.uniform(0.2, 0.4, shape_prefix + (1,))
        loc2 = rng.uniform(0.6, 0.8, shape_prefix + (1,))
        a1 = rng.uniform(0, 2, shape_prefix + (1,)) * amp_scale
        a2 = rng.uniform(-2, 0, shape_prefix + (1,)) * amp_scale
        s1 = rng.uniform(0.05, 0.15, shape_prefix + (1,))
        s2 = rng.uniform(0.05, 0.15, shape_prefix + (1,))
        return a1 * np.exp(-0.5 * ((x - loc1) / s1) ** 2) + a2 * np.exp(-0.5 * ((x - loc2) / s2) ** 2)

    u0 = bumps((n,), 1.0).astype(np.float32)
    f = bumps((n, num_t), 0.75).astype(np.float32)
    return u0, f
