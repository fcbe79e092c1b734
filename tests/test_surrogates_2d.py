"""The jellyfish 2-D surrogates (diffphycon_amd/model/surrogates_2d.py, stock torch ops with autograd) against the
reference modules: identical state_dict keys, bit-identical CPU forward (fixture jelly_surrogates.npz)."""
import numpy as np
import torch

from conftest import load_golden
from diffphycon_amd.model.surrogates_2d import Unet, ForceUnet


def test_boundary_updater_matches_reference():
    g = load_golden("jelly_surrogates")
    bd = Unet(dim=8, out_dim=3, dim_mults=(1, 2), channels=3).eval()
    sd = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wbd:")}
    assert sorted(sd) == sorted(bd.state_dict())
    bd.load_state_dict(sd)
    with torch.no_grad():
        y = bd(torch.from_numpy(g["x"]), torch.from_numpy(g["dtheta"]))
    assert (y - torch.from_numpy(g["y"])).abs().max() < 1e-6


def test_force_surrogate_matches_reference_and_has_a_gradient():
    g = load_golden("jelly_surrogates")
    torch.manual_seed(int(g["fm_seed"]))
    fm = ForceUnet(dim=64, out_dim=1, dim_mults=(1, 8), channels=4).eval()
    assert np.array_equal(fm.state_dict()["init_conv.weight"][0, 0].numpy(), g["fm_first_weight"])     # same seeded weights
    xf = torch.from_numpy(g["xf"]).requires_grad_()
    yf = fm(xf)
    assert (yf.detach() - torch.from_numpy(g["yf"])).abs().max() < 1e-5
    (grad,) = torch.autograd.grad(yf.sum(), xf)
    assert grad.shape == xf.shape and torch.isfinite(grad).all() and grad.abs().max() > 0
