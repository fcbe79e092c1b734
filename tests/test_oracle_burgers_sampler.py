"""Pins oracle/sampler_burgers.py against the reference's Burgers GaussianDiffusion (fixture burgers_sampler.npz from
tools/gen_golden.py burgers_sampler): step-size schedules, the closed-form guidance gradient, teacher-forced single
steps and 20-step free-running chains for the two-model (DiffPhyCon), normalised-beta and single-model recipes."""
import numpy as np
import pytest
import torch

from oracle import sampler_burgers as S
from oracle import unet2d as U
from conftest import load_golden

CASES = {
    "popc": dict(two=True, prior_beta=0.9, normalize_beta=False, w_sched="sigmoid_flip", J_sched="cosine",
                 set_zero=True, cond=True, w=(1.5, 0.02, 0.3, "front_rear_quarter")),
    "norm": dict(two=True, prior_beta=0.7, normalize_beta=True, w_sched=None, J_sched=None, set_zero=False, cond=True,
                 w=(0.5, 0.01, 0.0, None)),
    "lite": dict(two=False, prior_beta=1.0, normalize_beta=False, w_sched=None, J_sched="cosine", set_zero=False,
                 cond=False, w=(0.0, 0.0, 0.0, None)),
}
T = 20


@pytest.fixture(scope="module")
def g():
    return load_golden("burgers_sampler")


def test_scheduler_tables_bit_exact(g):
    # bit-identical on the generating host; 1e-10 relative across hosts (libm cos/exp last-bit differences)
    np.testing.assert_allclose(S.scheduler_table("cosine").numpy(), g["sched:J_cosine"], rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(S.scheduler_table("sigmoid").numpy(), g["sched:sigmoid"], rtol=1e-10, atol=1e-15)
    flip = S.scheduler_table("sigmoid_flip")
    np.testing.assert_allclose(flip[[0, 1, 500, 998, 999]].numpy(), g["sched:sigmoid_flip"], rtol=1e-10, atol=1e-15)
    assert S.scheduler_table(None).eq(1).all()
    with pytest.raises(ValueError):
        S.scheduler_table("plain_cosine")


def test_guidance_gradient_closed_form(g):
    x = torch.from_numpy(g["grad:x"])
    ut = torch.from_numpy(g["u_target"]) / 10
    for tag, (wu, wf, wreg, po) in {"full": (1.5, 0.02, 0.3, None), "po": (2.0, 0.0, 0.1, "front_rear_quarter")}.items():
        ref = torch.from_numpy(g["grad:" + tag])
        got = S.guidance_grad(x, ut, wu, wf, wreg, po)
        assert (got - ref).abs().max() <= 1e-6 * ref.abs().max() + 1e-9, tag
        assert not got[:, :, 11:].any() and not got[:, 1, 10].any()          # padding rows get no gradient


def _setup(g, c):
    ut = torch.from_numpy(g["u_target"])
    sched = S.make_schedule(T, "cosine")
    wu, wf, wreg, po = c["w"]
    grad_fn = lambda x0: S.guidance_grad(x0, ut / 10, wu, wf, wreg, po)       # noqa: E731
    return ut, sched, grad_fn, S.scheduler_table(c["w_sched"]), S.scheduler_table(c["J_sched"])


@pytest.mark.parametrize("tag", list(CASES))
def test_teacher_forced_steps(g, tag):
    c = CASES[tag]
    ut, sched, grad_fn, wtab, jtab = _setup(g, c)
    noise = torch.from_numpy(g[f"{tag}:noise"])
    for t in (19, 10, 1, 0):
        x = torch.from_numpy(g[f"{tag}:t{t}:x_in"])
        e_uw = torch.from_numpy(g[f"{tag}:t{t}:eps_uw"])
        e_w = torch.from_numpy(g[f"{tag}:t{t}:eps_w"]) if c["two"] else None
        z = noise[T - t] if t > 0 else None            # draw k belongs to step t = T - k
        out, x0, eps = S.p_sample_step(sched, x, t, e_uw, e_w, z, prior_beta=c["prior_beta"],
                                       normalize_beta=c["normalize_beta"], eta_w=wtab[t], eta_J=jtab[t], grad_fn=grad_fn,
                                       two_models=c["two"])
        assert (eps - torch.from_numpy(g[f"{tag}:t{t}:pred_noise"])).abs().max() < 1e-5
        assert (x0 - torch.from_numpy(g[f"{tag}:t{t}:x0"])).abs().max() < 1e-5
        assert (out - torch.from_numpy(g[f"{tag}:t{t}:x_out"])).abs().max() < 1e-5


@pytest.mark.parametrize("tag", list(CASES))
def test_free_running_chain(g, tag):
    c = CASES[tag]
    ut, sched, grad_fn, wtab, jtab = _setup(g, c)
    cfg = U.Unet2DConfig(dim=8, dim_mults=(1, 2), resnet_block_groups=1)
    sd_uw = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wuw:")}
    sd_w = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ww:")}
    B = ut.shape[0]

    def den(sd):
        return lambda x, t: U.unet2d_forward(sd, cfg, x, torch.full((B,), t, dtype=torch.long))

    with torch.no_grad():
        out = S.sample_chain(sched, T, den(sd_uw), den(sd_w) if c["two"] else None, torch.from_numpy(g[f"{tag}:noise"]),
                             u0=ut[:, 0] / 10 if c["cond"] else None, uT=ut[:, 10] / 10 if c["cond"] else None,
                             set_unobserved_to_zero=c["set_zero"], prior_beta=c["prior_beta"],
                             normalize_beta=c["normalize_beta"], w_table=wtab if c["w_sched"] else None,
                             J_table=jtab if c["J_sched"] else None, grad_fn=grad_fn)
    ref = torch.from_numpy(g[f"{tag}:final"])
    assert (out - ref).abs().max() < 5e-3, (out - ref).abs().max()           # free-running chain abs 5e-3
