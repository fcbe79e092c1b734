"""Generate tests/golden/*.npz by importing the reference (build container only).

    python tools/gen_golden.py [section ...]        sections: unet3d sampler schedules burgers phi unet2d

Every fixture is data only (inputs + the reference's outputs); no reference source text is
stored.  Weights stored here are the reference modules' own default initialisation under a
stated torch seed.  Re-running regenerates byte-identical files in this image.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("wrote", path, f"{os.path.getsize(path)/1024:.0f} KiB")


def sd_arrays(module, prefix="w:"):
    return {prefix + k: v for k, v in module.state_dict().items() if not k.endswith("rotary_emb.freqs")}


# ----------------------------------------------------------------------------- unet3d

def gen_unet3d():
    from model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import (
        Unet3D_with_Conv3D, RelativePositionBias)

    tabs = {}
    for n in (4, 20, 32, 64):
        q = torch.arange(n)
        rel = q[None, :] - q[:, None]
        tabs[f"n{n}"] = RelativePositionBias._relative_position_bucket(rel, num_buckets=32, max_distance=32)
    save("relpos_bucket", **tabs)

    for tag, channels, dim, mults, seed in (("joint", 6, 8, (1, 2), 0), ("w", 2, 8, (1, 2), 1),
                                            ("wide", 6, 16, (1, 2), 2)):
        torch.manual_seed(seed)
        m = Unet3D_with_Conv3D(dim=dim, dim_mults=mults, channels=channels).eval()
        frames, hw = (4, 16) if tag != "wide" else (3, 8)
        x = torch.randn(2, frames, channels, hw, hw)
        t = torch.tensor([7, 431])
        taps = {}
        hooks = []

        def mk(name):
            def hook(_m, _i, o):
                taps["tap:" + name] = o.detach().clone()
            return hook

        named = dict(m.named_modules())
        for name in ("init_conv", "init_temporal_attn", "time_mlp", "downs.0.0", "downs.0.1", "downs.0.2",
                     "downs.0.3", "downs.0.4", "mid_block1", "mid_spatial_attn", "mid_temporal_attn",
                     "mid_block2", "ups.0.0", "ups.0.2", "ups.0.3", "ups.0.4", "final_conv.0"):
            hooks.append(named[name].register_forward_hook(mk(name)))
        with torch.no_grad():
            y = m(x, t)
        for h in hooks:
            h.remove()
        arrays = dict(x=x, t=t, y=y, dim=dim, dim_mults=np.array(mults), channels=channels)
        if tag == "wide":
            # keep the file small: weights are large at dim 16 -> store, but drop most taps
            taps = {k: v for k, v in taps.items() if k in ("tap:init_conv", "tap:mid_block2")}
        arrays.update(taps)
        arrays.update(sd_arrays(m))
        save(f"unet3d_{tag}", **arrays)


# ----------------------------------------------------------------------------- schedules

def gen_schedules():
    from diffusion.diffusion_2d_smoke import GaussianDiffusion as GD2

    class _M:
        channels = 6
        self_condition = False

    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2"]
    out = {}
    for kind, T in (("sigmoid", 1000), ("cosine", 1000), ("linear", 1000), ("sigmoid", 20), ("cosine", 200)):
        gd = GD2.__new__(GD2)
        torch.nn.Module.__init__(gd)
        GD2.__init__(gd, _M(), image_size=16, frames=4, timesteps=T, beta_schedule=kind)
        for n in names:
            out[f"{kind}{T}:{n}"] = getattr(gd, n)
    # DDIM integer time pairs (diffusion_2d_smoke.py:729-731)
    for T, S in ((1000, 100), (1000, 50), (1000, 7), (20, 5)):
        times = torch.linspace(-1, T - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        out[f"ddim_pairs:{T}:{S}"] = np.array(list(zip(times[:-1], times[1:])), dtype=np.int64)
    save("schedules", **out)


# ----------------------------------------------------------------------------- smoke sampler

def gen_sampler():
    from model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffusion.diffusion_2d_smoke import GaussianDiffusion
    import inference.inference_2d_smoke as inf

    R = torch.tensor([2, 18, 20, 16, 20, 1], dtype=torch.float32).reshape(1, 1, 6, 1, 1)

    # guidance_fn itself (A8)
    torch.manual_seed(3)
    x0 = torch.randn(2, 4, 6, 16, 16)
    g0 = inf.guidance_fn(x0.clone().requires_grad_(), None, R, w_energy=0)
    g1 = inf.guidance_fn(x0.clone().requires_grad_(), None, R, w_energy=0.25)
    save("smoke_guidance", x0=x0, g_w0=g0, g_w025=g1)

    torch.manual_seed(0)
    mj = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2), channels=6).eval()
    mw = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2), channels=2).eval()
    B, Fr, HW = 2, 4, 16
    init = torch.zeros(B, HW, HW)
    init[0, 3:6, 4:7] = 0.5
    init[1, 8:11, 2:5] = 0.5

    def design_fn(x, low=None, init=None, init_u=None):
        return inf.guidance_fn(x, None, R, w_energy=0.0)

    def design_fn_e(x, low=None, init=None, init_u=None):
        return inf.guidance_fn(x, None, R, w_energy=0.5)

    common = dict(image_size=HW, frames=Fr, loss_type="l2", objective="pred_noise")
    arrays = dict(init=init)
    arrays.update(sd_arrays(mj, "wj:"))
    arrays.update(sd_arrays(mw, "ww:"))

    # ---- DDPM: T=20 schedule, full 20-step free-running chain + teacher-forced records of 3 steps
    for tag, dfn, guid, ratio, coeff, wexp in (("std", design_fn, "standard", 1e5, 0.0, 0.97),
                                               ("alpha", design_fn_e, "standard-alpha", 0.01, 0.3, 0.9)):
        gd = GaussianDiffusion([mj, mw], timesteps=20, sampling_timesteps=20, standard_fixed_ratio=ratio,
                               coeff_ratio=coeff, eval_2ddpm=True, w_prob_exp=wexp, **common)
        draws = []
        gen = torch.Generator().manual_seed(11)

        def sample_noise(shape, device, _d=draws, _g=gen):
            z = torch.randn(shape, generator=_g)
            _d.append(z)
            return z

        gd.sample_noise = sample_noise
        rec = []
        orig = gd.p_sample

        def p_sample(shape, x, t, *a, _orig=orig, _rec=rec, **k):
            xin = x.clone()
            out, x0_ = _orig(shape, x, t, *a, **k)
            _rec.append((t, xin, out.clone(), x0_.clone()))
            return out, x0_

        gd.p_sample = p_sample
        with torch.no_grad():
            res = gd.sample(batch_size=B, design_fn=dfn, design_guidance=guid, init=init)
        arrays[f"ddpm_{tag}:final"] = res
        arrays[f"ddpm_{tag}:noise"] = torch.stack(draws)          # [T, B,F,C,H,W]: initial + one per t>0
        for (t, xin, xout, x0_) in rec:
            if t in (19, 18, 10, 1, 0):
                arrays[f"ddpm_{tag}:t{t}:x_in"] = xin
                arrays[f"ddpm_{tag}:t{t}:x_out_pre_inpaint"] = xout
                arrays[f"ddpm_{tag}:t{t}:x0"] = x0_
                with torch.no_grad():
                    tt = torch.full((B,), t, dtype=torch.long)
                    arrays[f"ddpm_{tag}:t{t}:eps_j"] = mj(xin, tt)
                    arrays[f"ddpm_{tag}:t{t}:eps_w"] = mw(xin[:, :, 3:5], tt)

    # ---- DDIM: T=20, S=5, eta=1 (the CLI default path shape), torch global RNG -> record draws by replay
    gd = GaussianDiffusion([mj, mw], timesteps=20, sampling_timesteps=5, ddim_sampling_eta=1.0,
                           standard_fixed_ratio=1e5, coeff_ratio=0.0, eval_2ddpm=True, w_prob_exp=0.97, **common)
    torch.manual_seed(21)
    with torch.no_grad():
        res = gd.sample(batch_size=B, design_fn=design_fn, design_guidance="standard", init=init)
    torch.manual_seed(21)
    draws = [torch.randn(B, Fr, 6, HW, HW)] + [torch.randn(B, Fr, 6, HW, HW) for _ in range(4)]
    arrays["ddim:final"] = res
    arrays["ddim:noise"] = torch.stack(draws)

    # ---- run_model tail (A9)
    out = res * R
    out[:, :, -1] = out[:, :, -1].mean((-2, -1)).unsqueeze(-1).unsqueeze(-1).expand(-1, -1, HW, HW)
    arrays["post:out"] = out
    save("smoke_sampler", **arrays)


SECTIONS = {"unet3d": gen_unet3d, "schedules": gen_schedules, "sampler": gen_sampler}

if __name__ == "__main__":
    try:
        import gen_golden_burgers  # noqa: F401  (registers more sections when present)
        SECTIONS.update(gen_golden_burgers.SECTIONS)
    except ImportError:
        pass
    # the phi / smoke-evaluator fixtures have their own driver: python tools/gen_golden_phi.py
    want = sys.argv[1:] or list(SECTIONS)
    for s in want:
        print("==", s)
        SECTIONS[s]()
