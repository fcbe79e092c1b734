"""Pins oracle/sampler_jelly.py (and the torch surrogates it differentiates through) against the reference's jellyfish
GaussianDiffusion (fixture jelly_sampler.npz): design gradient, teacher-forced steps, 20-step chains for the
'standard-alpha' (CLI default) and 'standard' guidance modes."""
import numpy as np
import pytest
import torch

from oracle import sampler_jelly as S
from oracle import unet3d as U3
from conftest import load_golden
from diffphycon_amd.model.surrogates_2d import Unet, ForceUnet

T, FR = 20, 4
KW = {"alpha": dict(design_guidance="standard-alpha", coeff_ratio_J=0.3, coeff_ratio_w=0.3),
      "std": dict(design_guidance="standard", standard_fixed_ratio=0.003)}


@pytest.fixture(scope="module")
def env():
    g = load_golden("jelly_sampler")
    bd = Unet(dim=8, out_dim=3, dim_mults=(1, 2), channels=3).eval()
    bd.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wbd:")})
    torch.manual_seed(int(g["fm_seed"]))
    fm = ForceUnet(dim=64, out_dim=1, dim_mults=(1, 8), channels=4).eval()
    p_min, p_max, reg = float(g["p_min"]), float(g["p_max"]), float(g["reg_ratio"])
    design_fn = lambda x, bd0e: S.force_fn(x, bd0e, fm, bd, p_min, p_max, reg)      # noqa: E731
    return g, bd, design_fn


def test_design_gradient(env):
    g, bd, design_fn = env
    bd0e = torch.from_numpy(g["bd_0"]).unsqueeze(1).expand(-1, FR, -1, -1, -1)
    got = design_fn(torch.from_numpy(g["grad:x"]).clone(), bd0e)
    ref = torch.from_numpy(g["grad:g"])
    assert (got - ref).abs().max() <= 1e-5 * ref.abs().max()


@pytest.mark.parametrize("tag", ["alpha", "std"])
def test_teacher_forced_steps(env, tag):
    g, bd, design_fn = env
    sched = S.make_schedule(T, "sigmoid")
    bd0e = torch.from_numpy(g["bd_0"]).unsqueeze(1).expand(-1, FR, -1, -1, -1)
    steps = torch.from_numpy(g[f"{tag}:noise_steps"])
    for t in (19, 7, 0):
        z = steps[T - 1 - t] if t > 0 else None
        pred, x0 = S.p_sample_step(sched, torch.from_numpy(g[f"{tag}:t{t}:x_in"]), t, torch.from_numpy(g[f"{tag}:t{t}:eps_j"]),
                                   torch.from_numpy(g[f"{tag}:t{t}:eps_w"]), z, design_fn, bd0e, **KW[tag])
        assert (x0 - torch.from_numpy(g[f"{tag}:t{t}:x0"])).abs().max() < 1e-5
        ref = torch.from_numpy(g[f"{tag}:t{t}:pred"])
        assert (pred - ref).abs().max() <= 1e-5 * max(1.0, ref.abs().max().item()), t


@pytest.mark.parametrize("tag", ["alpha", "std"])
def test_free_running_chain(env, tag):
    g, bd, design_fn = env
    sched = S.make_schedule(T, "sigmoid")
    cj = U3.Unet3DConfig(dim=8, dim_mults=(1, 2), channels=7, out_dim=4)
    cw = U3.Unet3DConfig(dim=8, dim_mults=(1, 2), channels=7, out_dim=1)
    sj = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wj:")}
    sw = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ww:")}
    B = 2
    den = lambda sd, c: (lambda x, t: U3.unet3d_forward(sd, c, x, torch.full((B,), t, dtype=torch.long)))   # noqa: E731
    noise = {"init": [torch.from_numpy(g[f"{tag}:noise_init_{k}"]) for k in ("state", "bd", "theta")],
             "steps": torch.from_numpy(g[f"{tag}:noise_steps"])}
    with torch.no_grad():
        states, theta = S.sample_chain(sched, T, FR, den(sj, cj), den(sw, cw), noise, torch.from_numpy(g["state_0"]),
                                       torch.from_numpy(g["bd_0"]), torch.from_numpy(g["thetas_0"]), bd, design_fn, **KW[tag])
    assert (states - torch.from_numpy(g[f"{tag}:states"])).abs().max() < 5e-3
    assert (theta - torch.from_numpy(g[f"{tag}:theta"])).abs().max() < 5e-3
