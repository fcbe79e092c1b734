"""ORACLE (test infrastructure, not product): CPU restatement of the jellyfish guided sampler.

Follows /root/reference/diffusion/diffusion_2d_jellyfish.py (`GaussianDiffusion` :529-1006: model_predictions :703-757,
p_mean_variance :759-771, p_sample :777-806, update_bd :809-817, p_sample_loop :820-881, ddim_sample :884-966) and the
design objective of inference/inference_2d_jellyfish.py:49-114 (`reg_theta`, `force_fn`; autograd through the two 2-D
surrogates, which the oracle runs with the same torch ops on CPU).  fp32 torch-CPU; every random draw is injected.
Pinned against the reference on tests/golden/jelly_sampler.npz (tests/test_oracle_jelly_sampler.py).
"""
import torch

from .sampler_smoke import make_schedule          # sigmoid schedule by default (:541), same fp64 derivation


def reg_theta(theta):
    """inference_2d_jellyfish.py:49-61."""
    d = theta[:, 1:] - theta[:, :-1]
    return torch.sum(d * d, dim=1)


def force_fn(x, bd_0, force_model, bd_updater, p_min, p_max, reg_ratio, only_vis_pressure=False):
    """inference_2d_jellyfish.py:85-114: dJ/d(state), dJ/d(theta map) of
    J = -mean_t(force_t * (T - t)) + reg_ratio * sum_t (theta_{t+1} - theta_t)^2."""
    if only_vis_pressure:
        state, theta_expand = x[:, :, :1], x[:, :, -1]
    else:
        state, theta_expand = x[:, :, :3], x[:, :, 3]
    state.requires_grad_()
    theta_expand.requires_grad_()
    theta = torch.mean(torch.mean(theta_expand, dim=3), dim=2)
    pressure = state[:, :, 0] if only_vis_pressure else state[:, :, 2]
    pressure = (0.5 * pressure + 0.5) * (p_max - p_min) + p_min
    pred_bd = bd_updater(bd_0.reshape(-1, *bd_0.shape[2:]), theta.reshape(-1)).reshape(bd_0.shape)
    inp = torch.cat((pressure.unsqueeze(2), pred_bd), dim=2)
    force = force_model(inp.reshape(-1, *inp.shape[2:])).reshape(state.shape[0], state.shape[1])
    weight = torch.arange(force.shape[1], 0, -1, dtype=torch.float32).expand(force.shape[0], force.shape[1])
    guidance = -torch.mean(force * weight, dim=1) + reg_ratio * reg_theta(theta)
    gs, gt = torch.autograd.grad(guidance, [state, theta_expand], grad_outputs=torch.ones_like(guidance))
    return torch.cat([gs, gt.unsqueeze(2)], dim=2)


def diffused_view(x, only_vis_pressure=False):
    """x [B,F,Cx,H,W] -> the diffused channels (:713-716)."""
    if only_vis_pressure:
        return torch.cat([x[:, :, :1], x[:, :, -1:]], dim=2)
    return torch.cat([x[:, :, :3], x[:, :, 6:]], dim=2)


def p_sample_step(sched, x, t, eps_joint, eps_w, z, design_fn, bd_0_expand, *, design_guidance="standard-alpha",
                  coeff_ratio_J=0.3, coeff_ratio_w=0.3, standard_fixed_ratio=0.01, only_vis_pressure=False,
                  use_guidance_in_model_predictions=False):
    """p_sample :777-806 after the denoiser calls.  Returns (pred [B,F,Cd,H,W], x_start)."""
    xd = diffused_view(x, only_vis_pressure)
    x0 = sched["sqrt_recip_alphas_cumprod"][t] * xd - sched["sqrt_recipm1_alphas_cumprod"][t] * eps_joint
    x0 = x0.clamp(-1.0, 1.0)
    mean = sched["posterior_mean_coef1"][t] * x0 + sched["posterior_mean_coef2"][t] * xd
    pred = mean + (0.5 * sched["posterior_log_variance_clipped"][t]).exp() * z if t > 0 else mean
    if not use_guidance_in_model_predictions and design_fn is not None:
        with torch.enable_grad():
            g = design_fn(x0.clone().detach().requires_grad_(), bd_0_expand)
        if design_guidance == "standard":
            grad_final = standard_fixed_ratio * g - standard_fixed_ratio * eps_w
        elif design_guidance == "standard-alpha":
            eta_J = coeff_ratio_J * sched["betas"].flip(0)[t]
            eta_w = coeff_ratio_w * sched["betas"].flip(0)[t]
            grad_final = eta_J * g - eta_w * eps_w          # eps_w [B,F,1,H,W] broadcasts over all diffused channels
        else:
            raise ValueError(design_guidance)
        pred = pred - grad_final
    return pred, x0


def update_bd(bd_updater, theta_expand, bd_0_expand, thetas_0_frame_expand):
    """:809-817."""
    theta = torch.mean(torch.mean(theta_expand, dim=4), dim=3).squeeze(2)
    bd = bd_updater(bd_0_expand.reshape(-1, *bd_0_expand.shape[2:]), (theta - thetas_0_frame_expand).reshape(-1))
    return bd.reshape(bd_0_expand.shape)


def sample_chain(sched, T, frames, denoise_joint, denoise_w, noise, state_0, bd_0, thetas_0, bd_updater, design_fn, *,
                 cond_steps=1, **kw):
    """p_sample_loop :820-881 for cond_steps > 0 with injected noise: noise["init"] = (state, bd, theta) draws, then
    noise["steps"][k] for the k-th step with t > 0."""
    b, h, w = state_0.shape[0], state_0.shape[-2], state_0.shape[-1]
    n_state, n_bd, n_th = [n.clone() for n in noise["init"]]
    th0_map = thetas_0.reshape(b, 1, 1, 1, 1).expand(-1, 1, 1, h, w)
    th0_frames = thetas_0.unsqueeze(1).expand(-1, frames)
    bd_0_expand = bd_0.unsqueeze(1).expand(-1, frames, -1, -1, -1)
    assert cond_steps > 0
    n_state[:, :cond_steps] = state_0.unsqueeze(1)
    n_bd[:, :cond_steps] = bd_0.unsqueeze(1)
    n_th[:, :cond_steps] = th0_map
    n_th[:, -cond_steps:] = th0_map
    state_cond = state_0.unsqueeze(1).expand(-1, frames, -1, -1, -1)
    x = torch.cat([n_state, n_bd, n_th], dim=2)
    k = 0
    ns = n_state.shape[2]
    for t in reversed(range(T)):
        e_j = denoise_joint(x, t)
        e_w = denoise_w(torch.cat([state_cond, x[:, :, -4:]], dim=2), t)
        z = None
        if t > 0:
            z = noise["steps"][k]
            k += 1
        pred, _ = p_sample_step(sched, x, t, e_j, e_w, z, design_fn, bd_0_expand, **kw)
        pred_states, pred_theta = pred[:, :, :ns].clone(), pred[:, :, ns:].clone()
        pred_bd = update_bd(bd_updater, pred_theta, bd_0_expand, th0_frames)
        pred_states[:, :cond_steps] = state_0.unsqueeze(1)
        pred_bd[:, :cond_steps] = bd_0.unsqueeze(1)
        pred_bd[:, -cond_steps:] = bd_0.unsqueeze(1)
        pred_theta[:, :cond_steps] = th0_map
        pred_theta[:, -cond_steps:] = th0_map
        x = torch.cat([pred_states, pred_bd, pred_theta], dim=2)
    theta = torch.mean(torch.mean(pred_theta, dim=4), dim=3).squeeze(2)
    return pred_states, theta
