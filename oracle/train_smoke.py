"""CPU restatement (torch fp32, autograd) of the smoke denoiser's TRAINING step -- TEST INFRASTRUCTURE ONLY (imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path).

Follows, in /root/reference:
    q_sample                  diffusion/diffusion_2d_smoke.py:791-797
    p_losses                  diffusion/diffusion_2d_smoke.py:809-831   (conditioning :815-816, mse :827)
    Trainer.train             diffusion/diffusion_2d_smoke.py:998-1054  (backward :1025, clip_grad_norm_(1.0) :1027,
                              opt.step :1035, scheduler.step :1037, ema.update :1043)
    Adam                      torch.optim.Adam (torch 2.4.1, environment.yaml:94), lr / betas from Trainer :912,
                              train/train_2d_smoke.py:69 (lr 1e-3), eps 1e-8, no weight decay, no amsgrad
    MultiStepLR               Trainer :914  milestones [50000, 150000, 300000], gamma 0.1
    EMA                       ema-pytorch 0.7.3 (environment.yaml:41; Trainer :920 EMA(model, beta = 0.995, update_every = 10))

Pinned by tests/golden/train_{joint,w,wide}.npz (tools/gen_golden_train.py: loss, every parameter gradient, gradient norm
and post-Adam weights recorded from the reference itself).  PARITY UNPINNED for `ema_update` alone: ema-pytorch is a
third-party wheel that is neither vendored in the reference nor installable offline; its published 0.7.3 update rule
(update_after_step 100, inv_gamma 1, power 2/3, min_value 0, lerp with 1 - decay, copy while step <= update_after_step)
is restated here and cannot be checked against the wheel.
"""
import math

import torch

from . import unet3d as U
from .sampler_smoke import make_schedule


def q_sample(sched, x_start, t, noise):
    """diffusion_2d_smoke.py:791-797 (sched: oracle.sampler_smoke.make_schedule; fp32 buffers)."""
    a = sched["sqrt_alphas_cumprod"][t].reshape(-1, 1, 1, 1, 1)
    b = sched["sqrt_one_minus_alphas_cumprod"][t].reshape(-1, 1, 1, 1, 1)
    return a * x_start + b * noise


def p_losses(sd, cfg, sched, state_start, t, noise):
    """diffusion_2d_smoke.py:809-831 with loss_type 'l2', objective 'pred_noise'.  Returns the scalar loss (graph attached
    when tensors of `sd` require grad)."""
    noise = noise.clone()
    state = q_sample(sched, state_start, t, noise)
    state[:, 0, 0] = state_start[:, 0, 0]                    # :815 condition on the initial state
    noise[:, 0, 0] = 0                                       # :816
    out = U.unet3d_forward(sd, cfg, state, t)
    return torch.nn.functional.mse_loss(out, noise, reduction="mean").mean()


def loss_and_grads(sd, cfg, sched, state_start, t, noise):
    """loss + d loss / d parameter for every trainable tensor of the state dict (autograd through the oracle forward)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = p_losses(leaves, cfg, sched, state_start, t, noise)
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return loss.detach(), {k: g for k, g in zip(names, grads) if g is not None}


def grad_norm(grads):
    """torch.nn.utils.clip_grad_norm_'s total norm (norm of the per-tensor 2-norms)."""
    return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))


def clip_coef(total_norm, max_norm=1.0):
    """clip_grad_norm_: grads *= clamp(max_norm / (total + 1e-6), max = 1)."""
    return torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)


def adam_step(w, g, m, v, step, lr, beta1=0.9, beta2=0.99, eps=1e-8):
    """torch.optim.Adam single-tensor update (in place on w, m, v); `step` counts from 1."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    w.addcdiv_(m, denom, value=-(lr / bc1))


def lr_at(step, base_lr=1e-3, milestones=(50000, 150000, 300000), gamma=0.1):
    """MultiStepLR: learning rate used by the optimizer step number `step` (0-based count of completed scheduler steps)."""
    return base_lr * gamma ** sum(1 for m in milestones if step >= m)


def ema_decay(step, beta=0.995, update_after_step=100, inv_gamma=1.0, power=2.0 / 3.0, min_value=0.0):
    """ema-pytorch 0.7.3 get_current_decay (`step` = the EMA's own counter AFTER the increment of the current update)."""
    epoch = max(step - update_after_step - 1, 0)
    if epoch <= 0:
        return 0.0
    value = 1 - (1 + epoch / inv_gamma) ** -power
    return min(max(value, min_value), beta)


class EmaState:
    """ema-pytorch 0.7.3 EMA.update(): counter, `initted`, copy / lerp schedule (update_every 10 as Trainer :920 passes)."""

    def __init__(self, beta=0.995, update_every=10, update_after_step=100):
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.step, self.initted = 0, False

    def action(self):
        """Returns None (skip), ('copy', None) or ('lerp', weight) for this call, advancing the counter."""
        step = self.step
        self.step += 1
        if step % self.update_every != 0:
            return None
        if step <= self.update_after_step:
            return ("copy", None)
        if not self.initted:
            self.initted = True
            return ("copy+lerp", 1.0 - ema_decay(self.step, self.beta, self.update_after_step))
        return ("lerp", 1.0 - ema_decay(self.step, self.beta, self.update_after_step))


def ema_update(ema_sd, sd, state):
    act = state.action()
    if act is None:
        return
    kind, wgt = act
    for k in ema_sd:
        if kind.startswith("copy"):
            ema_sd[k].copy_(sd[k])
        if kind.endswith("lerp"):
            ema_sd[k].lerp_(sd[k], wgt)


def train_step(sd, cfg, sched, opt_state, state_start, t, noise, max_grad_norm=1.0):
    """One Trainer.train iteration (gradient_accumulate_every 1) on `sd` in place.  opt_state: dict with 'step', 'm', 'v'."""
    loss, grads = loss_and_grads(sd, cfg, sched, state_start, t, noise)
    total = grad_norm(grads)
    c = clip_coef(total, max_grad_norm)
    lr = lr_at(opt_state["step"])
    opt_state["step"] += 1
    for k, g in grads.items():
        if k not in opt_state["m"]:
            opt_state["m"][k], opt_state["v"][k] = torch.zeros_like(sd[k]), torch.zeros_like(sd[k])
        adam_step(sd[k], g * c, opt_state["m"][k], opt_state["v"][k], opt_state["step"], lr)
    return loss, grads, total


def schedule(timesteps=1000):
    s = make_schedule(timesteps, "sigmoid")
    if "sqrt_alphas_cumprod" not in s:
        raise KeyError("oracle.sampler_smoke.make_schedule must provide sqrt_alphas_cumprod / sqrt_one_minus_alphas_cumprod")
    return s
