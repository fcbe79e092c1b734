"""Burgers fixtures (imported by tools/gen_golden.py; build container only)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gen_burgers_fd():
    from gen_golden import save
    from dataset.apps.generate_burgers import burgers_numeric_solve_free
    from oracle.burgers import synthetic_inputs

    u0, f = synthetic_inputs(4, 128, 10, seed=7)
    with torch.no_grad():
        traj = burgers_numeric_solve_free(torch.from_numpy(u0), torch.from_numpy(f), visc=0.01, T=1.0, dt=1e-4, num_t=10)
    # a second, shorter configuration: 32 cells, 4 intervals, coarser dt
    u0b, fb = synthetic_inputs(3, 32, 4, seed=8)
    with torch.no_grad():
        trajb = burgers_numeric_solve_free(torch.from_numpy(u0b), torch.from_numpy(fb), visc=0.02, T=0.5, dt=5e-4, num_t=4)
    save("burgers_fd", u0=u0, f=f, traj=traj, u0b=u0b, fb=fb, trajb=trajb)


SECTIONS = {"burgers_fd": gen_burgers_fd}


# ----------------------------------------------------------------------------- Unet2D (B6)
def _unet2d_fixture(tag, dim, mults, groups, shape, seed):
    from gen_golden import save, sd_arrays
    from model.burgers_1d.unet import Unet2D

    torch.manual_seed(seed)
    m = Unet2D(dim=dim, init_dim=None, out_dim=2, dim_mults=mults, channels=2, resnet_block_groups=groups).eval()
    with torch.no_grad():                      # make the norm gains/biases matter
        for n_, p_ in m.named_parameters():
            if n_.endswith(".g") or n_.endswith("norm.weight") or n_.endswith("norm.bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    x = torch.randn(*shape)
    t = torch.tensor([3, 977][: shape[0]])
    taps, hooks = {}, []
    named = dict(m.named_modules())
    names = ["init_conv", "time_mlp", "downs.0.0", "downs.0.1", "downs.0.2", "downs.0.3", "mid_block1", "mid_attn",
             "mid_block2", "ups.0.0", "ups.0.1", "ups.0.2", "ups.0.3", "final_res_block"]
    last = len(mults) - 1
    names += [f"downs.{last}.3", f"ups.{last}.3"]
    for name in names:
        hooks.append(named[name].register_forward_hook(
            lambda _m, _i, o, name=name: taps.__setitem__("tap:" + name, o.detach().clone())))
    with torch.no_grad():
        y = m(x, t)
    for h in hooks:
        h.remove()
    arrays = dict(x=x, t=t, y=y, dim=dim, dim_mults=np.array(mults), groups=groups)
    arrays.update(taps)
    arrays.update(sd_arrays(m))
    save(f"unet2d_{tag}", **arrays)


def gen_unet2d():
    _unet2d_fixture("a", 8, (1, 2), 1, (2, 2, 16, 32), 0)          # groups = 1 as the scripts launch it
    _unet2d_fixture("b", 16, (1, 2, 4), 8, (1, 2, 16, 16), 1)      # library default groups = 8, three levels


# ----------------------------------------------------------------------------- Burgers sampler (B1-B5)
def gen_burgers_sampler():
    from gen_golden import save, sd_arrays
    from model.burgers_1d.unet import Unet2D
    from diffusion.diffusion_1d_burgers import (GaussianDiffusion, get_nablaJ, cosine_beta_J_schedule,
                                                sigmoid_schedule, sigmoid_schedule_flip)
    from utils import ddpm_guidance_loss, mse_dist_reg

    arrays = {}
    # B1/B5 tables
    tt = torch.arange(1000)
    arrays["sched:J_cosine"] = cosine_beta_J_schedule(tt)
    arrays["sched:sigmoid"] = sigmoid_schedule(tt)
    arrays["sched:sigmoid_flip"] = torch.stack([sigmoid_schedule_flip(int(i)) for i in (0, 1, 500, 998, 999)])
    torch.manual_seed(0)
    kw = dict(init_dim=None, out_dim=2, channels=2, resnet_block_groups=1)
    m_uw = Unet2D(dim=8, dim_mults=(1, 2), **kw).eval()
    m_w = Unet2D(dim=8, dim_mults=(1, 2), **kw).eval()
    arrays.update(sd_arrays(m_uw, "wuw:"))
    arrays.update(sd_arrays(m_w, "ww:"))
    B, Nx, T = 3, 32, 20
    g = torch.Generator().manual_seed(5)
    u_target = torch.randn(B, 11, Nx, generator=g)              # un-scaled target trajectories
    u0 = u_target[:, 0] / 10
    uT = u_target[:, 10] / 10
    arrays.update(u_target=u_target)
    for name in ("betas", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1",
                 "posterior_mean_coef2", "posterior_log_variance_clipped"):
        pass

    def loss_fn(wu, wf, wreg, po):
        def f(x):
            return ddpm_guidance_loss(u_target / 10, x[:, 0, :11, :], x[:, 1, :10, :], wu=wu, wf=wf, wreg=wreg,
                                      dist_reg=mse_dist_reg, partially_observed=po)
        return f

    # closed-form gradient check data (B4)
    xg = torch.randn(B, 2, 16, Nx, generator=g)
    for tag, (wu, wf, wreg, po) in {"full": (1.5, 0.02, 0.3, None), "po": (2.0, 0.0, 0.1, "front_rear_quarter")}.items():
        arrays[f"grad:{tag}"] = get_nablaJ(loss_fn(wu, wf, wreg, po))(xg.clone())
    arrays["grad:x"] = xg

    cases = {
        # DiffPhyCon POPC recipe (scripts/burgers_inference_partial_obs_partial_ctr.sh) with non-zero guidance weights
        "popc": dict(two=True, prior_beta=0.9, normalize_beta=False, w_sched="sigmoid_flip", J_sched="cosine",
                     set_zero=True, cond=True, w=(1.5, 0.02, 0.3, "front_rear_quarter")),
        # normalised-beta variant, fully observed, no schedulers
        "norm": dict(two=True, prior_beta=0.7, normalize_beta=True, w_sched=None, J_sched=None, set_zero=False,
                     cond=True, w=(0.5, 0.01, 0.0, None)),
        # single model (DiffPhyCon-lite), unconditioned, zero guidance weights as the shipped scripts
        "lite": dict(two=False, prior_beta=1.0, normalize_beta=False, w_sched=None, J_sched="cosine", set_zero=False,
                     cond=False, w=(0.0, 0.0, 0.0, None)),
    }
    sched_fn = {None: None, "cosine": cosine_beta_J_schedule, "sigmoid_flip": sigmoid_schedule_flip}
    for tag, c in cases.items():
        gd = GaussianDiffusion((m_uw, m_w) if c["two"] else m_uw, seq_length=(16, Nx), timesteps=T,
                               auto_normalize=False, use_conv2d=True, temporal=True, is_condition_u0=c["cond"],
                               is_condition_uT=c["cond"], set_unobserved_to_zero_during_sampling=c["set_zero"],
                               eval_two_models=c["two"], prior_beta=c["prior_beta"], normalize_beta=c["normalize_beta"])
        rec = []
        orig = gd.p_sample

        def p_sample(x, t, *a, _orig=orig, _rec=rec, **k):
            xin = x.detach().clone()
            out = _orig(x, t, *a, **k)
            _rec.append((t, xin, out[0].detach().clone(), out[1].detach().clone(), out[2].detach().clone()))
            return out

        gd.p_sample = p_sample
        torch.manual_seed(31)
        res = gd.sample(batch_size=B, clip_denoised=True, nablaJ=get_nablaJ(loss_fn(*c["w"])),
                        J_scheduler=sched_fn[c["J_sched"]], w_scheduler=sched_fn[c["w_sched"]], guidance_u0=True,
                        u_init=u0, u_final=uT)
        torch.manual_seed(31)
        draws = [torch.randn(B, 2, 16, Nx)] + [torch.randn(B, 2, 16, Nx) for _ in range(T - 1)]
        arrays[f"{tag}:final"] = res.detach()
        arrays[f"{tag}:noise"] = torch.stack(draws)
        for (t, xin, xout, x0_, pn) in rec:
            if t in (19, 10, 1, 0):
                arrays[f"{tag}:t{t}:x_in"] = xin
                arrays[f"{tag}:t{t}:x_out"] = xout
                arrays[f"{tag}:t{t}:x0"] = x0_
                arrays[f"{tag}:t{t}:pred_noise"] = pn
                with torch.no_grad():
                    tb = torch.full((B,), t, dtype=torch.long)
                    arrays[f"{tag}:t{t}:eps_uw"] = m_uw(xin, tb)
                    if c["two"]:
                        xw = xin.clone()
                        xw[..., 0, 1:10, :] = 0
                        arrays[f"{tag}:t{t}:eps_w"] = m_w(xw, tb)
    save("burgers_sampler", **arrays)


SECTIONS.update({"unet2d": gen_unet2d, "burgers_sampler": gen_burgers_sampler})


# ----------------------------------------------------------------------------- jellyfish 2-D surrogates (C2)
def gen_jelly_surrogates():
    from gen_golden import save, sd_arrays
    from diffusion.diffusion_2d_jellyfish import Unet, ForceUnet
    from diffphycon_amd.model import surrogates_2d as mine

    torch.manual_seed(4)
    bd = Unet(dim=8, out_dim=3, dim_mults=(1, 2), channels=3).eval()
    with torch.no_grad():
        for n_, p_ in bd.named_parameters():
            if n_.endswith(".g") or n_.endswith("norm.weight") or n_.endswith("norm.bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    x = torch.randn(3, 3, 16, 16)
    dth = torch.randn(3) * 0.2
    # ForceUnet ends in Linear(512, out): its bottleneck must be 512 wide (14 M weights).  To keep the fixture small the
    # weights are NOT stored: they are this repo's module initialised under torch.manual_seed(9) (deterministic CPU
    # generator), loaded into the reference module, whose output is what gets stored.
    torch.manual_seed(9)
    fm_mine = mine.ForceUnet(dim=64, out_dim=1, dim_mults=(1, 8), channels=4)
    fm = ForceUnet(dim=64, out_dim=1, dim_mults=(1, 8), channels=4).eval()
    fm.load_state_dict(fm_mine.state_dict())
    xf = torch.randn(2, 4, 8, 8)
    with torch.no_grad():
        y = bd(x, dth)
        yf = fm(xf)
    arrays = dict(x=x, dtheta=dth, y=y, xf=xf, yf=yf, fm_seed=9, fm_first_weight=fm_mine.state_dict()["init_conv.weight"][0, 0])
    arrays.update(sd_arrays(bd, "wbd:"))
    save("jelly_surrogates", **arrays)


SECTIONS.update({"jelly_surrogates": gen_jelly_surrogates})


# ----------------------------------------------------------------------------- jellyfish sampler (C1)
def gen_jelly_sampler():
    import argparse
    from gen_golden import save, sd_arrays
    from model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    import diffusion.diffusion_2d_jellyfish as DJ
    from torch.autograd import grad

    torch.manual_seed(2)
    mj = Unet3D_with_Conv3D(dim=8, out_dim=4, dim_mults=(1, 2), channels=7).eval()
    mw = Unet3D_with_Conv3D(dim=8, out_dim=1, dim_mults=(1, 2), channels=7).eval()
    bd = DJ.Unet(dim=8, out_dim=3, dim_mults=(1, 2), channels=3).eval()
    torch.manual_seed(9)
    from diffphycon_amd.model import surrogates_2d as mine
    fm_mine = mine.ForceUnet(dim=64, out_dim=1, dim_mults=(1, 8), channels=4)
    fm = DJ.ForceUnet(dim=64, out_dim=1, dim_mults=(1, 8), channels=4).eval()
    fm.load_state_dict(fm_mine.state_dict())
    B, Fr, HW, T = 2, 4, 16, 20
    p_min, p_max, reg_ratio = -1.7, 2.3, 1000.0
    g = torch.Generator().manual_seed(8)
    state_0 = torch.rand(B, 3, HW, HW, generator=g) * 2 - 1
    bd_0 = torch.rand(B, 3, HW, HW, generator=g)
    thetas_0 = torch.rand(B, generator=g) * 0.7 + 0.2
    args = argparse.Namespace(only_vis_pressure=False, device="cpu", reg_ratio=reg_ratio)

    def reg_theta(theta):
        return torch.sum((theta[:, 1:] - theta[:, :-1]) ** 2, dim=1)

    # force_fn of inference_2d_jellyfish.py:85-114 cannot be imported (the module unpickles a dataset file at import);
    # this is its call sequence on the reference's own surrogate modules, used only to PRODUCE the reference gradient
    def design_fn(x, bd0e):
        state, theta_expand = x[:, :, :3], x[:, :, 3]
        state.requires_grad_()
        theta_expand.requires_grad_()
        theta = torch.mean(torch.mean(theta_expand, dim=3), dim=2)
        pressure = (0.5 * state[:, :, 2] + 0.5) * (p_max - p_min) + p_min
        pred_bd = bd(bd0e.reshape(-1, *bd0e.shape[2:]), theta.reshape(-1)).reshape(bd0e.shape)
        inp = torch.cat((pressure.unsqueeze(2), pred_bd), dim=2)
        force = fm(inp.reshape(-1, *inp.shape[2:])).reshape(state.shape[0], state.shape[1])
        weight = torch.FloatTensor(range(force.shape[1], 0, -1)).expand(force.shape[0], force.shape[1])
        guidance = -torch.mean(force * weight, dim=1) + reg_ratio * reg_theta(theta)
        gs, gt = grad(guidance, [state, theta_expand], grad_outputs=torch.ones_like(guidance))
        return torch.cat([gs, gt.unsqueeze(2)], dim=2)

    arrays = dict(state_0=state_0, bd_0=bd_0, thetas_0=thetas_0, p_min=p_min, p_max=p_max, reg_ratio=reg_ratio, fm_seed=9)
    arrays.update(sd_arrays(mj, "wj:"))
    arrays.update(sd_arrays(mw, "ww:"))
    arrays.update(sd_arrays(bd, "wbd:"))
    xg = torch.rand(B, Fr, 4, HW, HW, generator=g) * 2 - 1
    arrays["grad:x"] = xg
    arrays["grad:g"] = design_fn(xg.clone(), bd_0.unsqueeze(1).expand(-1, Fr, -1, -1, -1))
    for tag, guid, kw in (("alpha", "standard-alpha", dict(coeff_ratio_J=0.3, coeff_ratio_w=0.3)),
                          ("std", "standard", dict(standard_fixed_ratio=0.003))):
        gd = DJ.GaussianDiffusion([mj, mw], image_size=HW, frames=Fr, cond_steps=1, timesteps=T, sampling_timesteps=T,
                                  loss_type="l2", objective="pred_noise", eval_2ddpm=True, device="cpu", **kw)
        draws = []
        gen = torch.Generator().manual_seed(12)

        def sample_noise(shape, device, _d=draws, _g=gen):
            zz = torch.randn(shape, generator=_g)
            _d.append(zz)
            return zz

        gd.sample_noise = sample_noise
        rec = []
        orig = gd.p_sample

        def p_sample(x, t, *a, _orig=orig, _rec=rec, **k):
            xin = x.clone()
            out = _orig(x, t, *a, **k)
            _rec.append((t, xin, out[0].clone(), out[1].clone()))
            return out

        gd.p_sample = p_sample
        with torch.no_grad():
            states, theta = gd.sample(design_fn=design_fn, design_guidance=guid, cond=[state_0, bd_0], thetas_0=thetas_0,
                                      bd_updater=bd)
        arrays[f"{tag}:states"], arrays[f"{tag}:theta"] = states, theta
        arrays[f"{tag}:noise_init_state"], arrays[f"{tag}:noise_init_bd"], arrays[f"{tag}:noise_init_theta"] = draws[:3]
        arrays[f"{tag}:noise_steps"] = torch.stack(draws[3:])
        for (t, xin, pred, x0_) in rec:
            if t in (19, 7, 0):
                arrays[f"{tag}:t{t}:x_in"], arrays[f"{tag}:t{t}:pred"], arrays[f"{tag}:t{t}:x0"] = xin, pred, x0_
                with torch.no_grad():
                    tb = torch.full((B,), t, dtype=torch.long)
                    arrays[f"{tag}:t{t}:eps_j"] = mj(xin, tb)
                    sc = state_0.unsqueeze(1).expand(-1, Fr, -1, -1, -1)
                    arrays[f"{tag}:t{t}:eps_w"] = mw(torch.cat([sc, xin[:, :, -4:]], dim=2), tb)
    save("jelly_sampler", **arrays)


SECTIONS.update({"jelly_sampler": gen_jelly_sampler})
