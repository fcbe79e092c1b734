"""Drop-in `GaussianDiffusion` for the jellyfish task: the reference's constructor and `.sample(...)` contract
(/root/reference/diffusion/diffusion_2d_jellyfish.py:529-1006) driving libdpc.

Per step: joint Unet3D forward (7 -> 4 channels) + theta Unet3D forward (7 -> 1) on libdpc, ONE posterior kernel
(dpc_ddpm_update_jelly), the design gradient through the two learned 2-D surrogates (model/surrogates_hip.py: forward and
backward on libdpc -- the reference's `force_fn`, inference_2d_jellyfish.py:85-114, as an explicit reverse pass; the autograd
restatement of it lives in tests/torch_force_fn.py as the test reference), ONE guidance kernel (dpc_jelly_apply_guidance), the
boundary updater forward and the conditioning writes.
`Unet` / `ForceUnet` (the surrogates' checkpoint containers) and `reg_theta` keep the reference's names."""
import ctypes as C
import math

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib
from ..model.surrogates_2d import ForceUnet, Unet  # noqa: F401  (re-exported under the reference's module path)
from .diffusion_2d_smoke import (_begin_noise_epoch, cosine_beta_schedule, default, linear_beta_schedule,
                                 sigmoid_beta_schedule)


def reg_theta(theta):
    """inference_2d_jellyfish.py:49-61: sum_t (theta_{t+1} - theta_t)^2."""
    d = theta[:, 1:] - theta[:, :-1]
    return torch.sum(d * d, dim=1)


class GaussianDiffusion(nn.Module):
    def __init__(self, model, *, image_size, frames=20, cond_steps=0, timesteps=1000, sampling_timesteps=None,
                 loss_type="l1", objective="pred_noise", beta_schedule="sigmoid", schedule_fn_kwargs=dict(),
                 ddim_sampling_eta=0., auto_normalize=True, min_snr_loss_weight=False, min_snr_gamma=5, backward_steps=5,
                 backward_lr=0.01, standard_fixed_ratio=0.01, forward_fixed_ratio=0.01, coeff_ratio_J=0.3,
                 coeff_ratio_w=0.3, only_vis_pressure=False, eval_2ddpm=False, w_prob_exp=1.0,
                 use_guidance_in_model_predictions=False, return_all_timesteps=True, device=None):
        super().__init__()
        if eval_2ddpm:
            self.model_states, self.model_thetas = model
            self.channels = self.model_states.channels
            self.self_condition = self.model_states.self_condition
        else:
            self.model = model
            self.channels = self.model.channels
            self.self_condition = self.model.self_condition
        assert objective == "pred_noise", "the jellyfish sampler implements pred_noise (:719)"
        self.frames, self.cond_steps, self.image_size, self.objective = frames, cond_steps, image_size, objective
        self.standard_fixed_ratio, self.coeff_ratio_J, self.coeff_ratio_w = standard_fixed_ratio, coeff_ratio_J, coeff_ratio_w
        self.only_vis_pressure, self.eval_2ddpm, self.w_prob_exp = only_vis_pressure, eval_2ddpm, w_prob_exp
        self.use_guidance_in_model_predictions = use_guidance_in_model_predictions
        fn = {"linear": linear_beta_schedule, "cosine": cosine_beta_schedule, "sigmoid": sigmoid_beta_schedule}
        if beta_schedule not in fn:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        betas = fn[beta_schedule](timesteps, **schedule_fn_kwargs)
        alphas = 1. - betas
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.)
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        self.sampling_timesteps = default(sampling_timesteps, timesteps)
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        host = {}

        def register_buffer(name, val):
            v = val.to(torch.float32)
            host[name] = v.clone()
            self.register_buffer(name, v)

        register_buffer("betas", betas)
        register_buffer("alphas_cumprod", alphas_cumprod)
        register_buffer("alphas_cumprod_prev", alphas_cumprod_prev)
        register_buffer("sqrt_alphas_cumprod", torch.sqrt(alphas_cumprod))
        register_buffer("sqrt_one_minus_alphas_cumprod", torch.sqrt(1. - alphas_cumprod))
        register_buffer("log_one_minus_alphas_cumprod", torch.log(1. - alphas_cumprod))
        register_buffer("sqrt_recip_alphas_cumprod", torch.sqrt(1. / alphas_cumprod))
        register_buffer("sqrt_recipm1_alphas_cumprod", torch.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        register_buffer("posterior_variance", posterior_variance)
        register_buffer("posterior_log_variance_clipped", torch.log(posterior_variance.clamp(min=1e-20)))
        register_buffer("posterior_mean_coef1", betas * torch.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        register_buffer("posterior_mean_coef2", (1. - alphas_cumprod_prev) * torch.sqrt(alphas) / (1. - alphas_cumprod))
        snr = alphas_cumprod / (1 - alphas_cumprod)
        maybe_clipped_snr = snr.clone()
        if min_snr_loss_weight:
            maybe_clipped_snr.clamp_(max=min_snr_gamma)
        register_buffer("loss_weight", maybe_clipped_snr / snr)
        self._host = host
        host["sigma"] = (0.5 * host["posterior_log_variance_clipped"]).exp()
        host["eta_J"] = self.coeff_ratio_J * host["betas"].clone().flip(0)          # (:783-784)
        host["eta_w"] = self.coeff_ratio_w * host["betas"].clone().flip(0)
        if device is not None:
            self.to(device)
        self.noise_seed, self.traj_offset, self._draw = None, 0, 0
        self.noise_epoch, self._calls = None, 0      # see diffusion_2d_smoke._begin_noise_epoch

    # ------------------------------------------------------------------ noise
    def sample_noise(self, shape, device):
        """Same injection point as the reference (:773); default: Philox stream per global trajectory."""
        out = torch.empty(list(shape), device=device, dtype=torch.float32)
        b = shape[0]
        seed = torch.initial_seed() if self.noise_seed is None else self.noise_seed
        _lib.check(_lib.lib().dpc_philox_normal(_lib.ptr(out), b, out.numel() // max(b, 1), seed & (2 ** 64 - 1),
                                                self.traj_offset, self._draw, _lib.stream()))
        self._draw += 1
        return out

    # ------------------------------------------------------------------ kernels
    def _n_state(self):
        return 1 if self.only_vis_pressure else 3

    def _coef(self, t, mode=0, clip=True):
        h = self._host
        c = _lib.JellyCoef()
        c.sqrt_recip_ac = h["sqrt_recip_alphas_cumprod"][t].item()
        c.sqrt_recipm1_ac = h["sqrt_recipm1_alphas_cumprod"][t].item()
        c.mean_coef1 = h["posterior_mean_coef1"][t].item()
        c.mean_coef2 = h["posterior_mean_coef2"][t].item()
        c.sigma = h["sigma"][t].item() if t > 0 else 0.0
        c.clip_denoised, c.mode = int(clip), mode
        return c

    def _posterior(self, x, eps, z, coef, eps_guided=None):
        B, Fr, Cx, H, W = x.shape
        ns = self._n_state()
        pred = torch.empty(B, Fr, ns + 1, H, W, device=x.device, dtype=torch.float32)
        x0 = torch.empty_like(pred)
        _lib.check(_lib.lib().dpc_ddpm_update_jelly(
            _lib.ptr(x), _lib.ptr(eps), _lib.ptr(eps_guided) if eps_guided is not None else None,
            _lib.ptr(z) if z is not None else None, _lib.ptr(pred), _lib.ptr(x0), C.byref(coef), B, Fr, Cx, ns, H, W,
            _lib.stream()))
        return pred, x0

    def _guide(self, io, g, eps_w, eta_J, eta_w, pad_w, sign):
        B, Fr, Cd, H, W = io.shape
        _lib.check(_lib.lib().dpc_jelly_apply_guidance(
            _lib.ptr(io), _lib.ptr(g.contiguous()) if g is not None else None, _lib.ptr(eps_w), float(eta_J), float(eta_w),
            int(pad_w), float(sign), B, Fr, Cd, H, W, _lib.stream()))

    def _denoise(self, x, state_cond, t_b):
        eps_j = self.model_states(x, t_b)
        x_w = torch.cat([state_cond, x[:, :, -4:]], dim=2)                # (:705)
        return eps_j, self.model_thetas(x_w, t_b)

    def _design(self, design_fn, x0, bd_0_expand):
        if not getattr(design_fn, "analytic", False):
            raise TypeError("design_fn must be a diffphycon_amd HipDesignGradient (model/surrogates_hip.py: the design gradient of "
                            "inference_2d_jellyfish.py:85-114 as an explicit reverse pass on libdpc); autograd closures are not on the HIP path")
        return design_fn(x0, bd_0_expand)

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def p_sample(self, x, t: int, bd_0_expand, state_cond, x_self_cond=None, clip_denoised=True, design_fn=None,
                 design_guidance="standard"):
        """One guided DDPM step (:777-806) -> (pred [B,F,Cd,H,W], x_start)."""
        if "recurrence" in design_guidance:
            raise NotImplementedError("recurrence guidance returns None in the reference as well (:787)")
        dev = x.device
        t_b = torch.full((x.shape[0],), t, device=dev, dtype=torch.long)
        eps_j, eps_w = self._denoise(x, state_cond, t_b)
        ns = self._n_state()
        z = self.sample_noise([x.shape[0], x.shape[1], ns + 1, x.shape[3], x.shape[4]], dev) if t > 0 else None
        pred, x0 = self._posterior(x, eps_j, z, self._coef(t, 0, clip_denoised))
        if not self.use_guidance_in_model_predictions and design_fn is not None:
            if not design_guidance.startswith("standard"):
                raise ValueError(design_guidance)
            g = self._design(design_fn, x0, bd_0_expand)
            if design_guidance == "standard":
                eta_J = eta_w = self.standard_fixed_ratio                                  # (:798)
            elif design_guidance == "standard-alpha":
                eta_J, eta_w = self._host["eta_J"][t].item(), self._host["eta_w"][t].item()  # (:800)
            else:
                raise ValueError(design_guidance)
            self._guide(pred, g, eps_w, eta_J, eta_w, pad_w=0, sign=-1.0)
        return pred, x0

    def update_bd(self, bd_updater, theta_expand_start, bd_0_expand, thetas_0_frame_expand):
        theta_start = torch.mean(torch.mean(theta_expand_start, dim=4), dim=3).squeeze(2)
        pred_bd = bd_updater(bd_0_expand.reshape(-1, *bd_0_expand.shape[2:]),
                             (theta_start - thetas_0_frame_expand).reshape(-1))
        return pred_bd.reshape(bd_0_expand.shape)

    def q_sample(self, x_start, t, noise=None):
        noise = default(noise, lambda: self.sample_noise(list(x_start.shape), x_start.device))
        a = self.sqrt_alphas_cumprod[t].reshape(-1, *((1,) * (x_start.dim() - 1)))
        b = self.sqrt_one_minus_alphas_cumprod[t].reshape(-1, *((1,) * (x_start.dim() - 1)))
        return a * x_start + b * noise

    def _init_state(self, shape, cond, thetas_0, bd_updater):
        b, f, c, h, w = shape
        device = self.betas.device
        state_0, bd_0 = cond[0].to(device).float(), cond[1].to(device).float()
        ns = self._n_state()
        noise_state = self.sample_noise([b, f, ns, h, w], device)
        noise_bd = self.sample_noise([b, f, 3, h, w], device)
        thetas_0 = thetas_0.to(device).float()
        noisy_thetas = self.sample_noise([b, f, 1, h, w], device)
        th0_map = thetas_0.reshape(b, 1, 1, 1, 1).expand(-1, 1, 1, h, w)
        th0_frames = thetas_0.unsqueeze(1).expand(-1, self.frames)
        bd_0_expand = bd_0.unsqueeze(1).expand(-1, self.frames, -1, -1, -1)
        bd_updater.to(device)
        bd_updater.eval()
        if self.cond_steps > 0:
            noise_state[:, :self.cond_steps] = state_0.unsqueeze(1)
            noise_bd[:, :self.cond_steps] = bd_0.unsqueeze(1)
            noisy_thetas[:, :self.cond_steps] = th0_map
            noisy_thetas[:, -self.cond_steps:] = th0_map
        state_cond = state_0.unsqueeze(1).expand(-1, f, -1, -1, -1)
        x = torch.cat([noise_state, noise_bd, noisy_thetas], dim=2).contiguous()
        return x, state_0, bd_0, th0_map, th0_frames, bd_0_expand, state_cond

    def _assemble(self, pred, t, state_0, bd_0, th0_map, th0_frames, bd_0_expand, bd_updater, repaint):
        ns = self._n_state()
        pred_states, pred_theta = pred[:, :, :ns], pred[:, :, ns:]
        pred_bd = self.update_bd(bd_updater, pred_theta, bd_0_expand, th0_frames)
        cs = self.cond_steps
        if cs > 0:
            pred_states[:, :cs] = state_0.unsqueeze(1)
            pred_bd[:, :cs] = bd_0.unsqueeze(1)
            pred_bd[:, -cs:] = bd_0.unsqueeze(1)
            pred_theta[:, :cs] = th0_map
            pred_theta[:, -cs:] = th0_map
        elif repaint:                                     # unconditional model: noisy condition (:867-875)
            tt = torch.full((pred.shape[0],), t, device=pred.device, dtype=torch.long)
            pred_states[:, :1] = self.q_sample(state_0, tt).unsqueeze(1)
            pred_bd[:, :1] = self.q_sample(bd_0, tt).unsqueeze(1)
            th_t = self.q_sample(th0_map, tt)
            pred_theta[:, :1] = th_t
            pred_theta[:, -1:] = th_t
        theta = torch.mean(torch.mean(pred_theta, dim=4), dim=3).squeeze(2)
        return torch.cat([pred_states, pred_bd, pred_theta], dim=2).contiguous(), [pred_states, theta]

    @torch.no_grad()
    def p_sample_loop(self, shape, design_fn=None, design_guidance="standard", return_all_timesteps=None, cond=None,
                      thetas_0=None, bd_updater=None, device=None):
        assert cond is not None
        x, state_0, bd_0, th0_map, th0_frames, bd_0_expand, state_cond = self._init_state(shape, cond, thetas_0, bd_updater)
        final = None
        for t in reversed(range(0, self.num_timesteps)):
            pred, _ = self.p_sample(x, t, bd_0_expand, state_cond, None, design_fn=design_fn, design_guidance=design_guidance)
            x, final = self._assemble(pred, t, state_0, bd_0, th0_map, th0_frames, bd_0_expand, bd_updater, repaint=True)
        return final

    @torch.no_grad()
    def ddim_sample(self, shape, design_fn=None, design_guidance="standard", return_all_timesteps=None, cond=None,
                    thetas_0=None, bd_updater=None, device=None):
        """(:884-966): guidance enters the predicted noise (use_guidance_in_model_predictions=True, w padded onto the
        theta channel); x_start is the unguided, unclipped estimate."""
        if return_all_timesteps:
            raise NotImplementedError("return_all_timesteps stacks a list of lists in the reference and fails there too")
        eta = self.ddim_sampling_eta
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        x, state_0, bd_0, th0_map, th0_frames, bd_0_expand, state_cond = self._init_state(shape, cond, thetas_0, bd_updater)
        ac = self._host["alphas_cumprod"]
        ns = self._n_state()
        final = None
        for time, time_next in zip(times[:-1], times[1:]):
            t_b = torch.full((x.shape[0],), time, device=x.device, dtype=torch.long)
            eps_j, eps_w = self._denoise(x, state_cond, t_b)
            c = self._coef(time, 2, clip=False)
            _, x0 = self._posterior(x, eps_j, None, c)
            if time_next < 0:
                continue                                  # the reference drops this estimate as well (:922-925)
            eps_g = eps_j.clone()
            g = self._design(design_fn, x0, bd_0_expand) if design_fn is not None else None
            if design_guidance == "standard":
                self._guide(eps_g, g, eps_w, self.standard_fixed_ratio, -(self.w_prob_exp - 1), pad_w=1, sign=1.0)
            elif design_guidance == "standard-alpha":
                self._guide(eps_g, g, eps_w, self._host["eta_J"][time].item(), self._host["eta_w"][time].item(), pad_w=1, sign=1.0)
            else:
                raise ValueError(design_guidance)
            alpha, alpha_next = ac[time], ac[time_next]
            sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            cc = (1 - alpha_next - sigma ** 2).sqrt()
            c.mode, c.mean_coef1, c.mean_coef2, c.sigma = 1, alpha_next.sqrt().item(), cc.item(), float(sigma)
            z = self.sample_noise([x.shape[0], x.shape[1], ns + 1, x.shape[3], x.shape[4]], x.device)
            pred, _ = self._posterior(x, eps_j, z, c, eps_guided=eps_g)
            x, final = self._assemble(pred, time, state_0, bd_0, th0_map, th0_frames, bd_0_expand, bd_updater, repaint=False)
        return final

    @torch.no_grad()
    def sample(self, batch_size=16, design_fn=None, design_guidance="standard", return_all_timesteps=False, cond=None,
               thetas_0=None, bd_updater=None, device=None):
        assert self.eval_2ddpm, "sampling uses the dual-model instance (inference_2d_jellyfish.py:213-236)"
        image_size, channels, frames = self.image_size, self.channels // 2, self.frames
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        batch_size = cond[0].shape[0]
        self._draw = _begin_noise_epoch(self)
        out = sample_fn((batch_size, frames, channels, image_size, image_size), design_fn, design_guidance,
                        return_all_timesteps=return_all_timesteps, cond=cond, thetas_0=thetas_0, bd_updater=bd_updater,
                        device=device)
        # ONE host read per sample(): the f16x3 range sentinels of the two denoisers and the watched window of the design gradient's
        # backward convolutions (model/surrogates_hip.py: _Calibration) -- results that left the window fail here, loudly
        for m in (self.model_states, self.model_thetas):
            if hasattr(m, "check_range"):
                m.check_range()
        if hasattr(design_fn, "check_range"):
            design_fn.check_range()
        return out


class Trainer(object):
    """Checkpoint READER only: `Trainer(diffusion, ..., results_path=...).load(milestone)` (inference_2d_jellyfish.py:
    163-180), file `model-{milestone}.pt` with a 'model' state-dict (diffusion_2d_jellyfish.py Trainer.save)."""

    def __init__(self, diffusion_model, *a, results_path="./results", **unused):
        from pathlib import Path
        self.model = diffusion_model
        self.results_path = Path(results_path)
        self.step = 0

    def load(self, milestone):
        data = torch.load(str(self.results_path / f"model-{milestone}.pt"), map_location="cpu")
        sd = {k: v for k, v in data["model"].items() if not k.endswith("rotary_emb.freqs")}
        self.model.load_state_dict(sd)
        self.step = data.get("step", 0)
