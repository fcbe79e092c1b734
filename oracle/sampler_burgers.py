"""ORACLE (test infrastructure, not product): CPU restatement of the Burgers guided sampler.

Follows /root/reference/diffusion/diffusion_1d_burgers.py (`GaussianDiffusion` :193-690: model_predictions :396-450,
p_mean_variance :452-461, p_sample :464-470, set_condition :500-522, p_sample_loop :525-584; step-size schedules
:71-111) and the guidance objective of inference/inference_1d_burgers.py:129-165 + utils.py:1286-1328, whose autograd
gradient is restated in closed form.  fp32 torch-CPU; every random draw is injected by the caller.
Pinned against the reference on tests/golden/burgers_sampler.npz (tests/test_oracle_burgers_sampler.py).
"""
import math

import torch

from .sampler_smoke import make_schedule  # same fp64 derivation; Burgers uses kind="cosine" (diffusion_1d_burgers.py:201)

RESCALER = 10.0          # inference_1d_burgers.py:15
CONDITION_IDX = 10       # diffusion_1d_burgers.py:224


# ----------------------------------------------------------------------------- step-size schedules (B5), fp64
def _cosine_J_table(s=0.008):
    timesteps = 1000
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def _sigmoid_table(start=-3, end=3, tau=1):
    timesteps = 1000
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((x * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def scheduler_table(name):
    """1000-entry fp64 table eta[t] of `get_scheduler(name)` (inference_1d_burgers.py:306-323); None -> all ones."""
    if name is None:
        return torch.ones(1000, dtype=torch.float64)
    if name == "cosine":
        return _cosine_J_table()                       # cosine_beta_J_schedule :71-82
    if name == "sigmoid":
        return _sigmoid_table()                        # sigmoid_schedule :94-108
    if name == "sigmoid_flip":
        return _sigmoid_table().flip(0)                # sigmoid_schedule(999 - t) :110-111
    raise ValueError(f"Unknown scheduler: {name}")     # 'plain_cosine' raises in the reference too (eta.flip() w/o dims)


# ----------------------------------------------------------------------------- guidance (B4)
def guidance_grad(x0, u_target_scaled, wu=0.0, wf=0.0, wreg=0.0, partially_observed=None):
    """d/dx of ddpm_guidance_loss(u_target/RESCALER, x[:,0,:11], x[:,1,:10], wu, wf, wreg, dist_reg=mse_dist_reg)
    (utils.py:1289-1328, :1286) summed over the batch as get_nablaJ does (diffusion_1d_burgers.py:34-49).
    x0 [B,2,16,Nx]; u_target_scaled [B,11(+),Nx] (only rows 0 and -1... i.e. rows 0 and 10 of the 11 used)."""
    B, _, _, nx = x0.shape
    g = torch.zeros_like(x0)
    u = x0[:, 0, :11, :]
    f = x0[:, 1, :10, :]
    m = torch.ones(nx)
    if partially_observed == "front_rear_quarter":
        m[nx // 4: (nx * 3) // 4] = 0
    elif partially_observed is not None:
        raise ValueError("Unknown partially observed mode")
    c = 2.0 * wu / (B * nx)
    g[:, 0, 0, :] += c * (u[:, 0] - u_target_scaled[:, 0]) * m
    g[:, 0, 10, :] += c * (u[:, 10] - u_target_scaled[:, -1]) * m
    g[:, 1, :10, :] += (2.0 * wf / B) * f
    d = u[:, 1:] - u[:, :-1]
    g[:, 0, 1:11, :] += 2.0 * wreg * d
    g[:, 0, 0:10, :] -= 2.0 * wreg * d
    return g


# ----------------------------------------------------------------------------- sampler (B2, B3)
def set_conditions(img, u0=None, uT=None, set_unobserved_to_zero=False):
    """In place, before every step (p_sample_loop :539-553)."""
    if u0 is not None:
        img[:, 0, 0, :] = u0
    if uT is not None:
        img[:, 0, CONDITION_IDX, :] = uT
    if set_unobserved_to_zero:
        nx = img.size(-1)
        img[:, 0, :, nx // 4: (nx * 3) // 4] = 0
    return img


def w_model_input(x):
    """x_w of model_predictions :399-400."""
    xw = x.clone()
    xw[..., 0, 1:CONDITION_IDX, :] = 0
    return xw


def p_sample_step(sched, x, t, eps_uw, eps_w, z, *, prior_beta=1.0, normalize_beta=False, eta_w=1.0, eta_J=1.0,
                  grad_fn=None, clip_denoised=True, two_models=True):
    """One guided DDPM step after the denoiser calls: model_predictions :396-450 (guidance_u0=True) +
    p_mean_variance :452-461 + p_sample :464-470.  eta_w / eta_J are the fp64 scheduler values of this step.
    Returns (x_next, x_start, pred_noise)."""
    if two_models:
        eps_w = eps_w.clone()
        eps_w[..., 0, :, :] = 0
        if normalize_beta:
            eps = (eps_uw - (1 - prior_beta) * eps_w) / prior_beta
        else:
            coef = torch.tensor((1 - prior_beta) * float(eta_w), dtype=torch.float64)
            eps = eps_uw - coef * eps_w                       # 0-dim fp64 tensor x fp32 tensor -> fp32
    else:
        eps = eps_uw
    a, b = sched["sqrt_recip_alphas_cumprod"][t], sched["sqrt_recipm1_alphas_cumprod"][t]
    x0 = a * x - b * eps
    if grad_fn is not None:
        eps = eps + grad_fn(x0) * torch.tensor(float(eta_J), dtype=torch.float64)
        x0 = a * x - b * eps
    if clip_denoised:
        x0 = x0.clamp(-1.0, 1.0)
    mean = sched["posterior_mean_coef1"][t] * x0 + sched["posterior_mean_coef2"][t] * x
    if t > 0:
        out = mean + (0.5 * sched["posterior_log_variance_clipped"][t]).exp() * z
    else:
        out = mean
    return out, x0, eps


def sample_chain(sched, T, denoise_uw, denoise_w, noise, *, u0=None, uT=None, set_unobserved_to_zero=False,
                 prior_beta=1.0, normalize_beta=False, w_table=None, J_table=None, grad_fn=None, clip_denoised=True):
    """p_sample_loop :525-584 with injected noise [T, ...] (noise[0] = initial image, then one draw per t > 0 in
    loop order).  denoise_* map (x, t:int) -> eps; denoise_w None = single-model sampling."""
    img = noise[0].clone()
    k = 1
    for t in reversed(range(T)):
        set_conditions(img, u0, uT, set_unobserved_to_zero)
        e_uw = denoise_uw(img, t)
        e_w = denoise_w(w_model_input(img), t) if denoise_w is not None else None
        z = None
        if t > 0:
            z = noise[k]
            k += 1
        img, _, _ = p_sample_step(sched, img, t, e_uw, e_w, z, prior_beta=prior_beta, normalize_beta=normalize_beta,
                                  eta_w=1.0 if w_table is None else w_table[t], eta_J=1.0 if J_table is None else J_table[t],
                                  grad_fn=grad_fn, clip_denoised=clip_denoised, two_models=denoise_w is not None)
    return img
