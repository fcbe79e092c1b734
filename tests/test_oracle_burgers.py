"""Pins oracle/burgers.py against the reference's burgers_numeric_solve_free (fixture burgers_fd.npz)."""
import numpy as np

from oracle import burgers as B
from conftest import load_golden


def test_burgers_fd_matches_reference():
    g = load_golden("burgers_fd")
    traj = B.burgers_numeric_solve_free(g["u0"], g["f"], visc=0.01, T=1.0, dt=1e-4, num_t=10)
    assert traj.shape == (4, 11, 128) and traj.dtype == np.float32
    # abs 1e-4 after 10 000 steps (einsum vs explicit stencil rounding differs in the last bits)
    d = np.abs(traj - g["traj"]).max()
    assert d < 1e-4, d
    trajb = B.burgers_numeric_solve_free(g["u0b"], g["fb"], visc=0.02, T=0.5, dt=5e-4, num_t=4)
    assert trajb.shape == (3, 5, 32)
    assert np.abs(trajb - g["trajb"]).max() < 1e-4


def test_u0_is_prepended_and_forcing_index_schedule():
    u0, f = B.synthetic_inputs(2, 16, 3, seed=1)
    traj = B.burgers_numeric_solve_free(u0, f, visc=0.01, T=0.03, dt=1e-3, num_t=3)
    assert np.array_equal(traj[:, 0], u0)
    # zero forcing + zero state stays zero (Dirichlet ghost cells)
    z = B.burgers_numeric_solve_free(np.zeros_like(u0), np.zeros_like(f), 0.01, 0.03, 1e-3, 3)
    assert not z.any()
