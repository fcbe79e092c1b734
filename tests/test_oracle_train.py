"""The training-step oracle (oracle/train_smoke.py) against the reference's own records (tests/golden/train_*.npz, made by
tools/gen_golden_train.py from diffusion_2d_smoke.py p_losses :809-831 + Trainer.train :998-1054): loss, every parameter
gradient, clip norm and the post-Adam weights of two consecutive optimizer steps."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import train_smoke as T
from oracle import unet3d as U


def _case(tag):
    g = load_golden(f"train_{tag}")
    cfg = U.Unet3DConfig(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=int(g["channels"]))
    sd = {k[3:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("w0:")}
    return g, cfg, sd


def _batch(g, cfg, step):
    state = torch.from_numpy(g[f"s{step}:state"])
    if cfg.channels == 2:
        state = state[:, :, 3:5]
    return state, torch.from_numpy(g[f"s{step}:t"]), torch.from_numpy(g[f"s{step}:noise"])


@pytest.mark.parametrize("tag", ["joint", "w", "wide"])
def test_training_oracle_matches_reference_records(tag):
    g, cfg, sd = _case(tag)
    sched = T.schedule(1000)
    opt = {"step": 0, "m": {}, "v": {}}
    steps = 1 if tag == "wide" else 2
    for step in range(steps):
        state, t, noise = _batch(g, cfg, step)
        loss, grads, total = T.train_step(sd, cfg, sched, opt, state, t, noise, float(g["max_grad_norm"]))
        assert abs(loss.item() - float(g[f"s{step}:loss"])) < 1e-6 * max(1, abs(loss.item()))
        names = [k[len(f"s{step}:g:"):] for k in g.files if k.startswith(f"s{step}:g:")]
        assert sorted(names) == sorted(grads), set(names) ^ set(grads)        # the same set of trainable tensors
        G = max(float(np.abs(g[f"s{step}:g:{k}"]).max()) for k in names)
        for k in names:
            ref = torch.from_numpy(g[f"s{step}:g:{k}"])
            # a conv bias in front of a one-channel-per-group GroupNorm (dim 8, 8 groups) has an exactly zero gradient: what
            # either side holds there is rounding noise (~1e-8), hence the floor relative to the largest gradient of the net
            assert (grads[k] - ref).abs().max().item() < 2e-4 * ref.abs().max().item() + 1e-6 * G, (step, k)

        assert abs(total.item() - float(g[f"s{step}:grad_norm"])) < 1e-5 * total.item()
        lr = float(g["lr"])
        for k in sd:
            ref = torch.from_numpy(g[f"s{step}:w:{k}"])
            # Adam's first steps move every weight by ~lr * sign(g) regardless of the gradient's size: compare in units of lr,
            # and where the recorded gradient element is itself rounding noise only bound the step (its sign is arbitrary)
            d = (sd[k] - ref).abs()
            live = torch.from_numpy(np.abs(g[f"s0:g:{k}"]) > 1e-4 * G) if f"s0:g:{k}" in g.files else torch.zeros_like(d, dtype=torch.bool)
            assert d[live].numel() == 0 or d[live].max().item() < 2e-2 * lr, (step, k, d[live].max().item())
            assert d.max().item() < 2.1 * lr * (step + 1), (step, k)


def test_q_sample_and_conditioning():
    """p_losses :813-816: q_sample then state[:, 0, 0] <- clean initial density, its noise target zeroed."""
    sched = T.schedule(1000)
    gen = torch.Generator().manual_seed(0)
    x0 = torch.randn(2, 3, 6, 4, 4, generator=gen)
    n = torch.randn(2, 3, 6, 4, 4, generator=gen)
    t = torch.tensor([0, 999])
    q = T.q_sample(sched, x0, t, n)
    a, b = sched["sqrt_alphas_cumprod"], sched["sqrt_one_minus_alphas_cumprod"]
    assert torch.equal(q[1], a[999] * x0[1] + b[999] * n[1])
    assert torch.allclose(q[0], x0[0], atol=0.1)             # t = 0: almost clean


def test_lr_schedule_and_ema_rule():
    assert T.lr_at(0) == 1e-3 and T.lr_at(49999) == 1e-3
    assert abs(T.lr_at(50000) - 1e-4) < 1e-18 and abs(T.lr_at(300000) - 1e-6) < 1e-18
    # ema-pytorch 0.7.3 (restated, unpinned): copies until step 100, then decay 1 - (1 + e) ** (-2/3) capped at beta
    st = T.EmaState(beta=0.995, update_every=10)
    acts = [st.action() for _ in range(131)]
    assert acts[0] == ("copy", None) and acts[1] is None and acts[100] == ("copy", None)
    kind, w = acts[110]
    assert kind == "copy+lerp" and abs(w - (1 + 10) ** (-2 / 3)) < 1e-12
    assert acts[120][0] == "lerp" and abs(acts[120][1] - (1 + 20) ** (-2 / 3)) < 1e-12
    assert T.ema_decay(10 ** 7) == 0.995
