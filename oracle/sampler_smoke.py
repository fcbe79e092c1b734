"""ORACLE (test infrastructure, not product): CPU restatement of the smoke guided sampler.

Follows /root/reference/diffusion/diffusion_2d_smoke.py (`GaussianDiffusion`, :451-789) and
the guidance objective of /root/reference/inference/inference_2d_smoke.py:30-44,179-197.
fp32 torch-CPU; every random draw is injected by the caller so chains are reproducible.
Pinned against the reference on tests/golden/smoke_sampler_*.npz (tests/test_oracle_sampler.py).
"""
import math

import torch
import torch.nn.functional as F

RESCALER = (2.0, 18.0, 20.0, 16.0, 20.0, 1.0)      # dataset/data_2d.py:167


# ----------------------------------------------------------------------------- schedules (A1)

def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1):
    """diffusion_2d_smoke.py:435-448 (fp64)."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def cosine_beta_schedule(timesteps, s=0.008):
    """diffusion_2d_smoke.py:423-433 / diffusion_1d_burgers.py cosine (fp64)."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def linear_beta_schedule(timesteps):
    """diffusion_2d_smoke.py:414-421."""
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def make_schedule(timesteps=1000, kind="sigmoid"):
    """fp64 derivation then cast to fp32 buffers (diffusion_2d_smoke.py:507-552)."""
    fn = {"sigmoid": sigmoid_beta_schedule, "cosine": cosine_beta_schedule, "linear": linear_beta_schedule}[kind]
    betas = fn(timesteps)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    d = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": torch.log(pv.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
    }
    return {k: v.to(torch.float32) for k, v in d.items()}


# ----------------------------------------------------------------------------- guidance (A8)

def guidance_grad(x0, rescaler, w_energy=0.0):
    """Closed form of `guidance_fn` (inference_2d_smoke.py:30-44).

    The reference rebinds x <- x*R before `grad`, so the result is dJ/d(x*R):
      -1/(H*W) on [b, F-1, C-1, :, :];  w_energy * 2*(x*R) / (F*2*H*W) on channels 3:5.
    """
    b, f, c, h, w = x0.shape
    xr = x0 * rescaler
    g = torch.zeros_like(x0)
    g[:, -1, -1] += -1.0 / (h * w)
    if w_energy != 0:
        g[:, :, 3:5] += w_energy * 2.0 * xr[:, :, 3:5] / (f * 2 * h * w)
    return g


def guidance_grad_autograd(x0, rescaler, w_energy=0.0):
    """Literal autograd form, used to cross-check the closed form."""
    x = (x0 * rescaler).detach().requires_grad_()
    succ = x[:, -1, -1].mean((-1, -2)).sum()
    energy = x[:, :, 3:5].square().mean((1, 2, 3, 4)).sum()
    j = -succ + w_energy * energy
    return torch.autograd.grad(j, x)[0]


# ----------------------------------------------------------------------------- one step (A3-A5)

def extract(a, t):
    return a[t].reshape(-1, 1, 1, 1, 1)


def model_predictions(sched, x, t, eps_joint, eps_w, rescaler, *, standard_fixed_ratio=1e5,
                      w_prob_exp=0.97, w_energy=0.0, design_guidance="standard", coeff_ratio=0.0,
                      clip_x_start=False, rederive_pred_noise=False):
    """diffusion_2d_smoke.py:610-656 given the two denoiser outputs."""
    c1 = extract(sched["sqrt_recip_alphas_cumprod"], t)
    c2 = extract(sched["sqrt_recipm1_alphas_cumprod"], t)
    pred_w = torch.zeros_like(eps_joint)
    pred_w[:, :, 3:5] = eps_w
    clip = (lambda v: v.clamp(-1.0, 1.0)) if clip_x_start else (lambda v: v)
    x0 = clip(c1 * x - c2 * eps_joint)
    g = guidance_grad(x0, rescaler, w_energy)
    if design_guidance == "standard":
        grad_final = standard_fixed_ratio * g + (w_prob_exp - 1) * pred_w
    elif design_guidance == "standard-alpha":
        eta = extract(coeff_ratio * sched["betas"].flip(0), t)
        grad_final = eta * g + (w_prob_exp - 1) * pred_w
    else:
        raise ValueError(design_guidance)
    eps = eps_joint + grad_final
    x0 = clip(c1 * x - c2 * eps)
    if clip_x_start and rederive_pred_noise:
        eps = (c1 * x - x0) / c2
    return eps, x0


def p_sample_step(sched, x, t_int, eps_joint, eps_w, z, init, rescaler, **kw):
    """One DDPM step incl. the in-paint of p_sample_loop (diffusion_2d_smoke.py:659-699, 720)."""
    b = x.shape[0]
    t = torch.full((b,), t_int, dtype=torch.long)
    _, x0 = model_predictions(sched, x, t, eps_joint, eps_w, rescaler, **kw)
    x0 = x0.clamp(-1.0, 1.0)
    mean = extract(sched["posterior_mean_coef1"], t) * x0 + extract(sched["posterior_mean_coef2"], t) * x
    logvar = extract(sched["posterior_log_variance_clipped"], t)
    noise = z if t_int > 0 else 0
    x_next = mean + (0.5 * logvar).exp() * noise
    x_next[:, 0, 0] = init
    return x_next, x0


def p_sample_loop(sched, model_joint, model_w, shape, init, rescaler, noises, steps=None, **kw):
    """diffusion_2d_smoke.py:703-723.  `noises[0]` is the initial draw, `noises[k]` the k-th in-loop draw.

    `steps`: iterable of t values (default reversed(range(T))).  model_*: callables (x, t[B]) -> eps.
    """
    x = noises[0].clone()
    x[:, 0, 0] = init
    T = sched["betas"].shape[0]
    steps = list(reversed(range(T))) if steps is None else list(steps)
    k = 1
    for t_int in steps:
        t = torch.full((shape[0],), t_int, dtype=torch.long)
        eps_j = model_joint(x, t)
        eps_w = model_w(x[:, :, 3:5], t)
        z = None
        if t_int > 0:
            z = noises[k]
            k += 1
        x, _ = p_sample_step(sched, x, t_int, eps_j, eps_w, z, init, rescaler, **kw)
    return x


# ----------------------------------------------------------------------------- DDIM (A6)

def ddim_time_pairs(total_timesteps, sampling_timesteps):
    """diffusion_2d_smoke.py:729-731 (int, bit-exact)."""
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_step(sched, x, time, time_next, eps_joint, eps_w, z, init, rescaler, eta=1.0, **kw):
    """Body of the ddim_sample loop (diffusion_2d_smoke.py:739-775)."""
    b = x.shape[0]
    t = torch.full((b,), time, dtype=torch.long)
    eps, x0 = model_predictions(sched, x, t, eps_joint, eps_w, rescaler, clip_x_start=True,
                                rederive_pred_noise=True, **kw)
    if time_next < 0:
        return x0
    alpha = sched["alphas_cumprod"][time]
    alpha_next = sched["alphas_cumprod"][time_next]
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    img = x0 * alpha_next.sqrt() + c * eps + sigma * z
    img[:, 0, 0] = init
    return img


def ddim_sample(sched, model_joint, model_w, shape, init, rescaler, noises, sampling_timesteps, eta=1.0, **kw):
    x = noises[0].clone()
    x[:, 0, 0] = init
    T = sched["betas"].shape[0]
    k = 1
    for time, time_next in ddim_time_pairs(T, sampling_timesteps):
        t = torch.full((shape[0],), time, dtype=torch.long)
        eps_j = model_joint(x, t)
        eps_w = model_w(x[:, :, 3:5], t)
        z = None
        if time_next >= 0:
            z = noises[k]
            k += 1
        x = ddim_step(sched, x, time, time_next, eps_j, eps_w, z, init, rescaler, eta=eta, **kw)
    return x


# ----------------------------------------------------------------------------- run_model tail (A9)

def postprocess(output, rescaler):
    """inference_2d_smoke.py:195-196: rescale, smoke-fraction channel <- its spatial mean."""
    out = output * rescaler
    out[:, :, -1] = out[:, :, -1].mean((-2, -1), keepdim=True).expand_as(out[:, :, -1])
    return out


def rescaler_tensor():
    return torch.tensor(RESCALER, dtype=torch.float32).reshape(1, 1, 6, 1, 1)
