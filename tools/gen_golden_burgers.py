"""Burgers fixtures (imported by tools/gen_golden.py; build container only)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gen_burgers_fd():
    from gen_golden import save
    from dataset.apps.generate_burgers import burgers_numeric_solve_free
    from oracle.burgers import synthetic_inputs

    u0, f = synthetic_inputs(4, 128, 10, seed=7)
    with torch.no_grad():
        traj = burgers_numeric_solve_free(torch.from_numpy(u0), torch.from_numpy(f), visc=0.01, T=1.0, dt=1e-4, num_t=10)
    # a second, shorter configuration: 32 cells, 4 intervals, coarser dt
    u0b, fb = synthetic_inputs(3, 32, 4, seed=8)
    with torch.no_grad():
        trajb = burgers_numeric_solve_free(torch.from_numpy(u0b), torch.from_numpy(fb), visc=0.02, T=0.5, dt=5e-4, num_t=4)
    save("burgers_fd", u0=u0, f=f, traj=traj, u0b=u0b, fb=fb, trajb=trajb)


SECTIONS = {"burgers_fd": gen_burgers_fd}
