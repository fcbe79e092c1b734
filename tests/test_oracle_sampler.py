"""Pins oracle/sampler_smoke.py against the reference (fixtures: schedules, smoke_guidance, smoke_sampler)."""
import numpy as np
import pytest
import torch

from oracle import sampler_smoke as S
from oracle import unet3d as O
from conftest import load_golden

NAMES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
         "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
         "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]


@pytest.mark.parametrize("kind,T", [("sigmoid", 1000), ("cosine", 1000), ("linear", 1000), ("sigmoid", 20), ("cosine", 200)])
def test_schedule_buffers_bit_exact(kind, T):
    g = load_golden("schedules")
    s = S.make_schedule(T, kind)
    for n in NAMES:
        assert np.array_equal(s[n].numpy(), g[f"{kind}{T}:{n}"]), n


@pytest.mark.parametrize("T,Sn", [(1000, 100), (1000, 50), (1000, 7), (20, 5)])
def test_ddim_pairs_bit_exact(T, Sn):
    g = load_golden("schedules")
    assert np.array_equal(np.array(S.ddim_time_pairs(T, Sn), dtype=np.int64), g[f"ddim_pairs:{T}:{Sn}"])


def test_guidance_closed_form():
    g = load_golden("smoke_guidance")
    x0 = torch.from_numpy(g["x0"])
    R = S.rescaler_tensor()
    assert torch.equal(S.guidance_grad(x0, R, 0.0), torch.from_numpy(g["g_w0"]))
    assert torch.allclose(S.guidance_grad(x0, R, 0.25), torch.from_numpy(g["g_w025"]), rtol=1e-6, atol=1e-9)
    assert torch.allclose(S.guidance_grad_autograd(x0, R, 0.25), torch.from_numpy(g["g_w025"]), rtol=1e-6, atol=1e-9)


CASES = {"std": dict(standard_fixed_ratio=1e5, w_prob_exp=0.97, w_energy=0.0, design_guidance="standard"),
         "alpha": dict(standard_fixed_ratio=0.01, w_prob_exp=0.9, w_energy=0.5, design_guidance="standard-alpha", coeff_ratio=0.3)}


@pytest.mark.parametrize("tag", ["std", "alpha"])
def test_teacher_forced_ddpm_steps(tag):
    g = load_golden("smoke_sampler")
    sched = S.make_schedule(20, "sigmoid")
    R = S.rescaler_tensor()
    init = torch.from_numpy(g["init"])
    noise = torch.from_numpy(g[f"ddpm_{tag}:noise"])
    for t in (19, 18, 10, 1, 0):
        x = torch.from_numpy(g[f"ddpm_{tag}:t{t}:x_in"])
        z = noise[20 - t] if t > 0 else None      # draw 0 = initial, draw k = k-th step
        xn, x0 = S.p_sample_step(sched, x, t, torch.from_numpy(g[f"ddpm_{tag}:t{t}:eps_j"]),
                                 torch.from_numpy(g[f"ddpm_{tag}:t{t}:eps_w"]), z, init, R, **CASES[tag])
        ref = torch.from_numpy(g[f"ddpm_{tag}:t{t}:x_out_pre_inpaint"]).clone()
        ref[:, 0, 0] = init
        # abs 1e-4 on [-1,1]-scale tensors per SURVEY 8(d); identical op order -> expect ~0
        assert torch.allclose(x0, torch.from_numpy(g[f"ddpm_{tag}:t{t}:x0"]), rtol=0, atol=1e-6)
        assert torch.allclose(xn, ref, rtol=0, atol=1e-6), (t, (xn - ref).abs().max())


def _models(g):
    cj = O.Unet3DConfig(dim=8, dim_mults=(1, 2), channels=6)
    cw = O.Unet3DConfig(dim=8, dim_mults=(1, 2), channels=2)
    sj = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wj:")}
    sw = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ww:")}
    return (lambda x, t: O.unet3d_forward(sj, cj, x, t)), (lambda x, t: O.unet3d_forward(sw, cw, x, t))


@pytest.mark.parametrize("tag", ["std", "alpha"])
def test_free_running_ddpm_chain(tag):
    g = load_golden("smoke_sampler")
    mj, mw = _models(g)
    sched = S.make_schedule(20, "sigmoid")
    init = torch.from_numpy(g["init"])
    noise = torch.from_numpy(g[f"ddpm_{tag}:noise"])
    with torch.no_grad():
        x = S.p_sample_loop(sched, mj, mw, tuple(noise[0].shape), init, S.rescaler_tensor(), list(noise), **CASES[tag])
    ref = torch.from_numpy(g[f"ddpm_{tag}:final"])
    # free-running chain tolerance (abs 5e-3); oracle and reference share op order
    assert torch.allclose(x, ref, rtol=0, atol=1e-4), (x - ref).abs().max()


def test_ddim_chain_and_postprocess():
    g = load_golden("smoke_sampler")
    mj, mw = _models(g)
    sched = S.make_schedule(20, "sigmoid")
    init = torch.from_numpy(g["init"])
    noise = torch.from_numpy(g["ddim:noise"])
    with torch.no_grad():
        x = S.ddim_sample(sched, mj, mw, tuple(noise[0].shape), init, S.rescaler_tensor(), list(noise), 5, eta=1.0,
                          **CASES["std"])
    ref = torch.from_numpy(g["ddim:final"])
    assert torch.allclose(x, ref, rtol=0, atol=1e-4), (x - ref).abs().max()
    out = S.postprocess(ref.clone(), S.rescaler_tensor())
    assert torch.allclose(out, torch.from_numpy(g["post:out"]), rtol=1e-6, atol=1e-6)
