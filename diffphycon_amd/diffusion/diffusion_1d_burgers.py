"""Drop-in `GaussianDiffusion` for the 1-D Burgers task: the reference's constructor and `.sample(...)` contract
(/root/reference/diffusion/diffusion_1d_burgers.py:193-690) driving libdpc.

Per step the hot path is: dpc_burgers_prepare (conditioning rows, zero-fill, prior-model input) -> joint Unet2D forward
(+ prior Unet2D forward) -> ONE fused guidance + posterior update kernel (dpc_ddpm_update_burgers).  No autograd graph
and no per-step device sync: the reference's `t[0].item()` scheduler look-ups (:405, :432) become host scalars.
"""
import ctypes as C
import math
import os
from collections import namedtuple

import torch
import torch.nn.functional as F
from torch import nn

from .. import _lib
from .diffusion_2d_smoke import _begin_noise_epoch

ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


# ----------------------------------------------------------------------------- step-size schedules (:71-127), fp64
def cosine_beta_J_schedule(t, s=0.008):
    timesteps = 1000
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    alphas_cumprod = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.999)[t]


def plain_cosine_schedule(t, s=0.0):
    timesteps = 1000
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    eta = torch.cos((x + s) / (timesteps + s))
    return eta.flip()[t]          # raises exactly like the reference (:92: flip() without dims)


def sigmoid_schedule(t, start=-3, end=3, tau=1, clamp_min=1e-5):
    timesteps = 1000
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    alphas_cumprod = (-((x * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.999)[t]


def sigmoid_schedule_flip(t):
    return sigmoid_schedule(999 - t)


def linear_beta_schedule(timesteps):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    alphas_cumprod = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return torch.clip(betas, 0, 0.999)


# ----------------------------------------------------------------------------- guidance
class BurgersGuidance:
    """`get_nablaJ(get_loss_fn_2dconv(...))` (inference_1d_burgers.py:129-168, utils.py:1289-1328) in closed form.

    J = wu*mean_{b,x}((u0-u0*)^2 + (uT-uT*)^2 [centre half zeroed if partially observed]) + wf*mean_b sum f^2
        + wreg*sum (u[t+1]-u[t])^2,   u = x[:,0,:11], f = x[:,1,:10], targets = u_target / RESCALER.
    Calling it returns dJ/dx like the reference's nablaJ; the sampler reads the weights and evaluates the same
    expression inside the fused update kernel."""

    def __init__(self, u_target_scaled, wu=0.0, wf=0.0, wreg=0.0, partially_observed=None):
        if partially_observed not in (None, "front_rear_quarter"):
            raise ValueError("Unknown partially observed mode")
        self.u_target = u_target_scaled            # [B, Nt(>=11), Nx], already divided by RESCALER
        self.wu, self.wf, self.wreg = float(wu), float(wf), float(wreg)
        self.partially_observed = partially_observed

    def target_rows(self, device):
        """[B, 2, Nx]: rows t=0 and t=T of the rescaled target (utils.py:1303-1304)."""
        ut = torch.as_tensor(self.u_target, dtype=torch.float32)
        return torch.stack((ut[:, 0, :], ut[:, -1, :]), dim=1).to(device).contiguous()

    def __call__(self, x):
        B, _, _, nx = x.shape
        g = torch.zeros_like(x)
        u, f = x[:, 0, :11, :], x[:, 1, :10, :]
        tr = self.target_rows(x.device)
        m = torch.ones(nx, device=x.device)
        if self.partially_observed == "front_rear_quarter":
            m[nx // 4: (nx * 3) // 4] = 0
        c = 2.0 * self.wu / (B * nx)
        g[:, 0, 0, :] += c * (u[:, 0] - tr[:, 0]) * m
        g[:, 0, 10, :] += c * (u[:, 10] - tr[:, 1]) * m
        g[:, 1, :10, :] += (2.0 * self.wf / B) * f
        d = u[:, 1:] - u[:, :-1]
        g[:, 0, 1:11, :] += 2.0 * self.wreg * d
        g[:, 0, 0:10, :] -= 2.0 * self.wreg * d
        return g


def get_nablaJ(loss_fn):
    """Reference signature (:34-49).  On the HIP path the loss must be a `BurgersGuidance` (closed form); arbitrary
    autograd closures have no kernel."""
    if isinstance(loss_fn, BurgersGuidance):
        return loss_fn
    raise TypeError("get_nablaJ needs a diffphycon_amd BurgersGuidance (closed-form ddpm_guidance_loss); arbitrary "
                    "autograd closures are not on the HIP path")


class GaussianDiffusion(nn.Module):
    def __init__(self, model, *, seq_length, timesteps=1000, sampling_timesteps=None, objective="pred_noise",
                 beta_schedule="cosine", ddim_sampling_eta=0., auto_normalize=True, guidance_u0=True, temporal=False,
                 use_conv2d=False, is_condition_u0=False, is_condition_uT=False, is_condition_u0_zero_pred_noise=True,
                 is_condition_uT_zero_pred_noise=True, train_on_partially_observed=None,
                 set_unobserved_to_zero_during_sampling=False, conditioned_on_residual=None, residual_on_u0=False,
                 recurrence=False, recurrence_k=1, is_model_w=False, eval_two_models=False, expand_condition=False,
                 prior_beta=1, normalize_beta=False, train_on_padded_locations=True, condition_idx=10):
        super().__init__()
        if not eval_two_models:
            self.model = model
            self.channels = self.model.channels
            self.self_condition = self.model.self_condition
        else:
            self.model_uw, self.model_w = model[0], model[1]
            self.channels = self.model_uw.channels
            self.self_condition = self.model_uw.self_condition
        if not (temporal and use_conv2d):
            raise NotImplementedError("the HIP path implements the 2-D conv (temporal=True, use_conv2d=True) sampler "
                                      "that get_2d_ddpm builds (train_1d_burgers.py:145-168)")
        assert type(seq_length) is tuple and len(seq_length) == 2, "should be a tuple of (Nt, Nx)"
        if auto_normalize or conditioned_on_residual is not None or expand_condition or is_model_w:
            raise NotImplementedError("auto_normalize / residual conditioning / expand_condition / is_model_w "
                                      "are not used by the DiffPhyCon inference scripts")
        self.recurrence, self.recurrence_k = bool(recurrence), int(recurrence_k)                  # (:353-354)
        assert objective == "pred_noise", "the Burgers sampler implements pred_noise"
        self.temporal, self.conv2d, self.traj_size = True, True, seq_length
        self.objective = objective
        if beta_schedule == "linear":
            betas = linear_beta_schedule(timesteps)
        elif beta_schedule == "cosine":
            betas = cosine_beta_schedule(timesteps)
        else:
            raise ValueError(f"unknown beta schedule {beta_schedule}")
        alphas = 1. - betas
        alphas_prev = F.pad(alphas[:-1], (1, 0), value=1.)
        alphas_cumprod = torch.cumprod(alphas, dim=0)
        alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.)
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
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
        self.alphas = alphas.to(torch.float32).clone()
        self.alphas_prev = alphas_prev.to(torch.float32).clone()
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
        register_buffer("loss_weight", torch.ones_like(snr))
        self._host = host
        self._host["sigma"] = (0.5 * host["posterior_log_variance_clipped"]).exp()           # (:469)
        self.guidance_u0 = guidance_u0
        self.is_condition_u0, self.is_condition_uT = is_condition_u0, is_condition_uT
        self.set_unobserved_to_zero_during_sampling = set_unobserved_to_zero_during_sampling
        self.eval_two_models = eval_two_models
        self.prior_beta, self.normalize_beta = prior_beta, normalize_beta
        self.condition_idx = condition_idx
        self.noise_seed = None          # None -> torch.initial_seed() at sample() time
        self.traj_offset = 0            # global index of this rank's first trajectory (batch sharding)
        self.guidance_batch = None      # batch size the guidance loss averages over (None -> local batch)
        self.noise_epoch, self._calls = None, 0      # see diffusion_2d_smoke._begin_noise_epoch
        self._draw = 0
        self.use_graph = os.environ.get("DPC_BURGERS_GRAPH", "1") != "0"     # HIP-graph replay of the denoiser forwards
        # r06: the prior net on a forked side stream (two parallel branches of the captured graph): 16.65 -> 15.02 ms per POPC step at
        # B = 256, bit-identical (tests/test_gpu_burgers_sampler.py); DPC_BURGERS_TWO_STREAMS=0 restores the single stream
        self.two_streams = os.environ.get("DPC_BURGERS_TWO_STREAMS", "1") != "0"
        self._graphs, self._t_static = {}, None

    # ------------------------------------------------------------------ noise
    def sample_noise(self, shape, device):
        """Counter-based N(0,1) keyed (seed, global trajectory index, draw): sharding-invariant."""
        out = torch.empty(shape, device=device, dtype=torch.float32)
        b = shape[0]
        per = out.numel() // max(b, 1)
        seed = torch.initial_seed() if self.noise_seed is None else self.noise_seed
        _lib.check(_lib.lib().dpc_philox_normal(_lib.ptr(out), b, per, seed & (2 ** 64 - 1), self.traj_offset, self._draw,
                                                _lib.stream()))
        self._draw += 1
        return out

    # ------------------------------------------------------------------ per-step pieces
    def _coef(self, t, guide, J_scheduler, w_scheduler, clip_denoised, batch):
        h = self._host
        c = _lib.BurgersCoef()
        c.sqrt_recip_ac = h["sqrt_recip_alphas_cumprod"][t].item()
        c.sqrt_recipm1_ac = h["sqrt_recipm1_alphas_cumprod"][t].item()
        c.mean_coef1 = h["posterior_mean_coef1"][t].item()
        c.mean_coef2 = h["posterior_mean_coef2"][t].item()
        c.sigma = h["sigma"][t].item() if t > 0 else 0.0
        c.two_models = int(self.eval_two_models)
        c.normalize_beta = int(bool(self.normalize_beta))
        c.prior_beta = float(self.prior_beta)
        if self.eval_two_models:
            eta = float(w_scheduler(t)) if w_scheduler is not None else 1.0                  # (:405)
            c.w_coef = (1 - self.prior_beta) if self.normalize_beta else (1 - self.prior_beta) * eta
        c.eta_J = float(J_scheduler(t)) if J_scheduler is not None else 1.0                  # (:432, :495)
        if guide is not None:
            c.wu, c.wf, c.wreg = guide.wu, guide.wf, guide.wreg
            c.partially_observed = int(guide.partially_observed == "front_rear_quarter")
        c.guidance_batch = int(self.guidance_batch or batch)
        c.clip_denoised = int(bool(clip_denoised))
        c.cond_idx = self.condition_idx
        return c

    def _prepare(self, img, x_w, u0, uT):
        B, _, nt, nx = img.shape
        _lib.check(_lib.lib().dpc_burgers_prepare(
            _lib.ptr(img), _lib.ptr(x_w) if x_w is not None else None, _lib.ptr(u0) if u0 is not None else None,
            _lib.ptr(uT) if uT is not None else None, B, nt, nx, self.condition_idx,
            int(self.set_unobserved_to_zero_during_sampling), _lib.stream()))

    def _update(self, x, e_uw, e_w, z, tgt, out, coef, x0_out=None, eps_out=None):
        B, _, nt, nx = x.shape
        _lib.check(_lib.lib().dpc_ddpm_update_burgers(
            _lib.ptr(x), _lib.ptr(e_uw), _lib.ptr(e_w) if e_w is not None else None, _lib.ptr(z) if z is not None else None,
            _lib.ptr(tgt) if tgt is not None else None, _lib.ptr(out), _lib.ptr(x0_out) if x0_out is not None else None,
            _lib.ptr(eps_out) if eps_out is not None else None, C.byref(coef), B, nt, nx, _lib.stream()))

    def _denoise(self, img, x_w, t_b):
        if self.eval_two_models:
            if getattr(self, "two_streams", False):
                # the two denoisers are independent (own weights, own workspaces): the prior net runs on a side stream, forked from and
                # joined to the current one -- inside a HIP-graph capture this becomes two parallel branches of the graph.  At B = 256 most
                # launches of these nets do not fill the device (deep levels: 64 workgroups), so the branches overlap.
                cur = torch.cuda.current_stream()
                side = self._side_stream()
                side.wait_stream(cur)
                e_uw = self.model_uw(img, t_b)
                with torch.cuda.stream(side):
                    e_w = self.model_w(x_w, t_b)
                cur.wait_stream(side)
                return e_uw, e_w
            return self.model_uw(img, t_b), self.model_w(x_w, t_b)
        return self.model(img, t_b), None

    def _side_stream(self):
        s = getattr(self, "_side", None)
        if s is None:
            s = self._side = torch.cuda.Stream()
        return s

    def _denoise_step(self, img, x_w, t: int):
        """The two denoiser forwards of step t.  With `use_graph` the several hundred launches of the two Unet2D forwards are
        captured ONCE into a HIP graph (keyed by the state buffers' addresses and shape) and replayed per step; the time step
        is a device tensor refreshed before each replay, the state is updated in place by the sampler, so nothing else changes
        between replays.  At B = 256 the forwards are launch-latency bound (16 x 128 images down to 1 x 8): replaying removes
        the per-launch host cost and the gaps between launches.  Not used while per-kernel event timing is on (bench.py's
        roofline leg): events inside a captured stream cannot be read back."""
        B = img.shape[0]
        if not self.use_graph:
            return self._denoise(img, x_w, torch.full((B,), t, device=img.device, dtype=torch.long))
        models = (self.model_uw, self.model_w) if self.eval_two_models else (self.model,)
        if any(m._dirty or m._device != img.device for m in models):
            self._graphs.clear()                    # new weights / device: a replay would run the old ones (it skips _sync)

        def stamp():
            # everything a captured launch sequence bakes in besides the state buffers: the weights' upload generation, the
            # activation workspace address, the arithmetic mode and the debug-tap switch of each denoiser
            return tuple((m._version, 0 if m._ws is None else m._ws.data_ptr(), str(m.arithmetic), bool(getattr(m, "_taps", False)))
                         for m in models)
        key = (img.data_ptr(), 0 if x_w is None else x_w.data_ptr(), tuple(img.shape))
        if key in self._graphs and self._graphs[key][2] != stamp():
            del self._graphs[key]
        if key not in self._graphs:
            if len(self._graphs) >= 2:              # (two state buffers when the sampler ping-pongs; anything else: start over)
                self._graphs.clear()
            if self._t_static is None or self._t_static.shape[0] != B or self._t_static.device != img.device:
                self._t_static = torch.full((B,), t, device=img.device, dtype=torch.long)
                self._graphs.clear()
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):            # eager warm-up: weight upload, workspace, one-time attribute / scratch set-up
                for _ in range(2):
                    self._denoise(img, x_w, self._t_static)
            cur.wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._denoise(img, x_w, self._t_static)
            self._graphs[key] = (g, out, stamp())
        self._t_static.fill_(t)
        g, out, _ = self._graphs[key]
        g.replay()
        return out

    @staticmethod
    def _guide(kwargs):
        nablaJ = kwargs.get("nablaJ")
        if nablaJ is not None and not isinstance(nablaJ, BurgersGuidance):
            raise TypeError("nablaJ must be a diffphycon_amd BurgersGuidance (see get_nablaJ)")
        if kwargs.get("proj_guidance") is not None:
            raise NotImplementedError("proj_guidance is commented out in the reference's inference script "
                                      "(inference_1d_burgers.py:379) and has no kernel")
        return nablaJ

    @torch.no_grad()
    def p_sample(self, x, t: int, x_self_cond=None, residual=None, **kwargs):
        """One guided step (:464-470) on an already conditioned x: returns (pred_img, x_start, pred_noise)."""
        guide = self._guide(kwargs)
        dev = x.device
        B = x.shape[0]
        t_b = torch.full((B,), t, device=dev, dtype=torch.long)
        x_w = None
        if self.eval_two_models:                     # x is taken as already conditioned: only build the prior input
            x_w = torch.empty_like(x)
            _lib.check(_lib.lib().dpc_burgers_prepare(_lib.ptr(x), _lib.ptr(x_w), None, None, B, x.shape[2], x.shape[3],
                                                      self.condition_idx, 0, _lib.stream()))
        e_uw, e_w = self._denoise(x, x_w, t_b)
        z = self.sample_noise(list(x.shape), dev) if t > 0 else None
        out, x0, eps = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        tgt = guide.target_rows(dev) if guide is not None and guide.wu != 0 else None
        coef = self._coef(t, guide, kwargs.get("J_scheduler"), kwargs.get("w_scheduler"), kwargs.get("clip_denoised", True), B)
        self._update(x, e_uw, e_w, z, tgt, out, coef, x0, eps)
        return out, x0, eps

    @torch.no_grad()
    def p_sample_loop(self, shape, **kwargs):
        assert not self.is_ddim_sampling, "wrong branch!"
        if not self.guidance_u0:
            raise NotImplementedError("guidance_u0=False (guidance on x_t with a second p_sample) is not used by the scripts")
        guide = self._guide(kwargs)
        device = self.betas.device
        B = shape[0]
        u0 = kwargs["u_init"].to(device=device, dtype=torch.float32).contiguous() if self.is_condition_u0 else None
        uT = kwargs["u_final"].to(device=device, dtype=torch.float32).contiguous() if self.is_condition_uT else None
        tgt = guide.target_rows(device) if guide is not None and guide.wu != 0 else None
        J_sched, w_sched = kwargs.get("J_scheduler"), kwargs.get("w_scheduler")
        clip = kwargs.get("clip_denoised", True)
        img = self.sample_noise(list(shape), device)
        x_w = torch.empty_like(img) if self.eval_two_models else None
        pingpong = torch.empty_like(img) if (guide is not None and guide.wreg != 0) else None
        for t in reversed(range(0, self.num_timesteps)):
            for _k in range(self.recurrence_k):                  # (:535) one pass unless --recurrence
                self._prepare(img, x_w, u0, uT)
                e_uw, e_w = self._denoise_step(img, x_w, t)
                z = self.sample_noise(list(shape), device) if t > 0 else None
                coef = self._coef(t, guide, J_sched, w_sched, clip, B)
                if pingpong is None:
                    self._update(img, e_uw, e_w, z, tgt, img, coef)
                else:
                    self._update(img, e_uw, e_w, z, tgt, pingpong, coef)
                    img, pingpong = pingpong, img
                if not self.recurrence:
                    break
                img.copy_(self.recurrent_sample(img, t))         # (:578-582) self recurrence: add the noise of level t back
        return img                                  # unnormalize = identity (auto_normalize=False, train_1d_burgers.py:148)

    @torch.no_grad()
    def recurrent_sample(self, x_tm1, t: int):
        """(:472-482) x_t = sqrt(a_t / a_{t-1}) x_{t-1} + sqrt(1 - a_t / a_{t-1}) z with the per-step alphas (a_{-1} = 1);
        no noise at t == 0.  Elementwise on the device; the noise is the sampler's counter-based stream."""
        ratio = self.alphas[t] / self.alphas_prev[t]                                              # fp32 host scalars, as extract() reads
        xtm1_coef, noise_coef = torch.sqrt(ratio), torch.sqrt(1 - ratio)
        if t > 0:
            return xtm1_coef.item() * x_tm1 + noise_coef.item() * self.sample_noise(list(x_tm1.shape), x_tm1.device)
        return xtm1_coef.item() * x_tm1

    @torch.no_grad()
    def ddim_sample(self, shape, return_all_timesteps=False, **kwargs):
        """(:587-644) the reference's Burgers DDIM: single model only (it asserts eval_two_models == False), NO guidance,
        clip_x_start + rederived noise; conditioning / zero-fill per step as in the DDPM loop."""
        assert not self.eval_two_models, "ddim_sample: the reference asserts eval_two_models == False (:618)"
        if return_all_timesteps:
            raise NotImplementedError("return_all_timesteps is not used by the scripts")
        device, total, S, eta = self.betas.device, self.num_timesteps, self.sampling_timesteps, self.ddim_sampling_eta
        times = list(reversed(torch.linspace(-1, total - 1, steps=S + 1).int().tolist()))
        u0 = kwargs["u_init"].to(device=device, dtype=torch.float32).contiguous() if self.is_condition_u0 else None
        uT = kwargs["u_final"].to(device=device, dtype=torch.float32).contiguous() if self.is_condition_uT else None
        ac = self._host["alphas_cumprod"]
        img = self.sample_noise(list(shape), device)
        B = shape[0]
        h = self._host
        x0, scratch = torch.empty_like(img), torch.empty_like(img)
        for time, time_next in zip(times[:-1], times[1:]):
            self._prepare(img, None, u0, uT)
            e_uw, _ = self._denoise_step(img, None, time)
            # x_start = clip(c1 x - c2 eps) from the fused kernel (no guidance: model_predictions is called without nablaJ, :620);
            # the DDIM combination itself is three elementwise passes over a [B, 2, 16, 128] tensor, done with device tensor ops
            # (the reference's Burgers DDIM is unguided and rejected by the two-model sampler: not a hot path)
            self._update(img, e_uw, None, None, None, scratch, self._coef(time, None, None, None, True, B), x0)
            if time_next < 0:
                img.copy_(x0)
                continue
            c1, c2 = h["sqrt_recip_alphas_cumprod"][time].item(), h["sqrt_recipm1_alphas_cumprod"][time].item()
            pred_noise = (c1 * img - x0) / c2                                                     # rederive_pred_noise (:438-439)
            alpha, alpha_next = ac[time], ac[time_next]
            sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            cc = (1 - alpha_next - sigma ** 2).sqrt()
            z = self.sample_noise(list(shape), device)
            img.copy_(x0 * alpha_next.sqrt().item() + cc.item() * pred_noise + float(sigma) * z)
        return img

    @torch.no_grad()
    def sample(self, batch_size=16, clip_denoised=True, **kwargs):
        if "guidance_u0" in kwargs:
            self.guidance_u0 = kwargs["guidance_u0"]
        if self.is_condition_u0:
            assert "u_init" in kwargs and kwargs["u_init"] is not None
        if self.is_condition_uT:
            assert "u_final" in kwargs and kwargs["u_final"] is not None
        sample_size = (batch_size, self.channels, *self.traj_size)
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        self._draw = _begin_noise_epoch(self)
        return sample_fn(sample_size, clip_denoised=clip_denoised, **kwargs)


class Trainer(object):
    """Checkpoint READER only (training is out of scope): `Trainer(ddpm, dataset, results_folder=...).load(m)` as used
    by inference_1d_burgers.py:182-206; file name and format of diffusion_1d_burgers.py:934-965
    (`cos10000-model-{milestone}.pt`, torch.save({'step','model','opt','ema','scaler','version'}))."""

    def __init__(self, diffusion_model, dataset=None, *, results_folder="./results", train_num_steps=100000,
                 save_and_sample_every=1000, **unused):
        from pathlib import Path
        self.model = diffusion_model
        self.results_folder = Path(results_folder)
        self.step = 0

    def load(self, milestone):
        data = torch.load(str(self.results_folder / f"cos10000-model-{milestone}.pt"), map_location="cpu")
        self.model.load_state_dict(data["model"])
        self.step = data.get("step", 0)

    def save(self, milestone):
        self.results_folder.mkdir(exist_ok=True, parents=True)
        data = {"step": self.step, "model": self.model.state_dict(), "opt": None, "ema": None, "scaler": None}
        torch.save(data, str(self.results_folder / f"cos10000-model-{milestone}.pt"))
