"""GPU parity of the jellyfish guided sampler (Unet3D 7->4 / 7->1 on libdpc, dpc_ddpm_update_jelly,
dpc_jelly_apply_guidance, torch surrogates for the design gradient) against the reference's records
(tests/golden/jelly_sampler.npz).  Tolerances: teacher-forced step abs 1e-4, 20-step chain abs 5e-3."""
import argparse

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
T, FR = 20, 4
KW = {"alpha": ("standard-alpha", dict(coeff_ratio_J=0.3, coeff_ratio_w=0.3)),
      "std": ("standard", dict(standard_fixed_ratio=0.003))}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def env(dev):
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffphycon_amd.diffusion import diffusion_2d_jellyfish as DJ
    g = load_golden("jelly_sampler")
    mj = Unet3D_with_Conv3D(dim=8, out_dim=4, dim_mults=(1, 2), channels=7)
    mw = Unet3D_with_Conv3D(dim=8, out_dim=1, dim_mults=(1, 2), channels=7)
    mj.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wj:")})
    mw.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ww:")})
    bd = DJ.Unet(dim=8, out_dim=3, dim_mults=(1, 2), channels=3).eval()
    bd.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("wbd:")})
    torch.manual_seed(int(g["fm_seed"]))
    fm = DJ.ForceUnet(dim=64, out_dim=1, dim_mults=(1, 8), channels=4).eval()
    bd, fm = bd.to(dev), fm.to(dev)
    args = argparse.Namespace(only_vis_pressure=False, device=dev, reg_ratio=float(g["reg_ratio"]), p_min=float(g["p_min"]),
                              p_max=float(g["p_max"]))

    from torch_force_fn import design_fn_torch
    design_fn = design_fn_torch(fm, bd, args)        # the reference's force_fn on the torch modules + autograd (test reference)

    def make(tag):
        guid, kw = KW[tag]
        return DJ.GaussianDiffusion([mj.to(dev), mw.to(dev)], image_size=16, frames=FR, cond_steps=1, timesteps=T,
                                    sampling_timesteps=T, loss_type="l2", objective="pred_noise", eval_2ddpm=True,
                                    device=dev, **kw), guid

    from diffphycon_amd.model.surrogates_hip import HipDesignGradient
    args.image_size = 16
    hip = HipDesignGradient(fm, bd, args)
    return g, bd, design_fn, make, hip


def test_design_gradient_on_gpu(env, dev):
    g, bd, design_fn, make, hip = env
    bd0e = torch.from_numpy(g["bd_0"]).to(dev).unsqueeze(1).expand(-1, FR, -1, -1, -1)
    got = design_fn(torch.from_numpy(g["grad:x"]).to(dev).clone(), bd0e).cpu()
    ref = torch.from_numpy(g["grad:g"])
    assert (got - ref).abs().max() <= 2e-4 * ref.abs().max()        # MIOpen/rocBLAS fp32 vs oneDNN through 2 U-Nets + backward


@pytest.mark.parametrize("tag", ["alpha", "std"])
def test_teacher_forced_posterior_and_guidance_kernels(env, tag, dev):
    """The two HIP kernels on the reference's recorded inputs, with the reference-side gradient injected (isolates the
    kernels from the surrogate backward)."""
    from oracle import sampler_jelly as S
    g, bd, design_fn, make, hip = env
    gd, guid = make(tag)
    sched = S.make_schedule(T, "sigmoid")
    steps = torch.from_numpy(g[f"{tag}:noise_steps"])
    for t in (19, 7, 0):
        x = torch.from_numpy(g[f"{tag}:t{t}:x_in"]).to(dev)
        e_j = torch.from_numpy(g[f"{tag}:t{t}:eps_j"]).to(dev)
        e_w = torch.from_numpy(g[f"{tag}:t{t}:eps_w"]).to(dev)
        z = steps[T - 1 - t].to(dev) if t > 0 else None
        pred, x0 = gd._posterior(x, e_j, z, gd._coef(t, 0, True))
        assert (x0.cpu() - torch.from_numpy(g[f"{tag}:t{t}:x0"])).abs().max() < 1e-5
        gg = design_fn(x0.clone(), torch.from_numpy(g["bd_0"]).to(dev).unsqueeze(1).expand(-1, FR, -1, -1, -1))
        if guid == "standard":
            eJ = eW = gd.standard_fixed_ratio
        else:
            eJ, eW = gd._host["eta_J"][t].item(), gd._host["eta_w"][t].item()
        gd._guide(pred, gg, e_w, eJ, eW, pad_w=0, sign=-1.0)
        ref = torch.from_numpy(g[f"{tag}:t{t}:pred"])
        assert (pred.cpu() - ref).abs().max() <= 1e-4 * max(1.0, ref.abs().max().item()), t


@pytest.mark.parametrize("tag", ["alpha", "std"])
def test_free_running_chain_vs_reference(env, tag, dev):
    g, bd, design_fn, make, hip = env
    gd, guid = make(tag)
    draws = [torch.from_numpy(g[f"{tag}:noise_init_{k}"]) for k in ("state", "bd", "theta")] + \
        list(torch.from_numpy(g[f"{tag}:noise_steps"]))
    it = iter(draws)
    gd.sample_noise = lambda shape, device: next(it).to(device).clone()
    states, theta = gd.sample(design_fn=design_fn, design_guidance=guid, cond=[torch.from_numpy(g["state_0"]),
                              torch.from_numpy(g["bd_0"])], thetas_0=torch.from_numpy(g["thetas_0"]), bd_updater=bd)
    assert (states.cpu() - torch.from_numpy(g[f"{tag}:states"])).abs().max() < 5e-3
    assert (theta.cpu() - torch.from_numpy(g[f"{tag}:theta"])).abs().max() < 5e-3


@pytest.mark.parametrize("tag", ["alpha", "std"])
def test_free_running_chain_with_the_hip_surrogates(env, tag, dev):
    """Same chain with the design gradient AND the boundary updater on libdpc (no torch surrogate in the loop)."""
    g, bd, design_fn, make, hip = env
    gd, guid = make(tag)
    draws = [torch.from_numpy(g[f"{tag}:noise_init_{k}"]) for k in ("state", "bd", "theta")] + \
        list(torch.from_numpy(g[f"{tag}:noise_steps"]))
    it = iter(draws)
    gd.sample_noise = lambda shape, device: next(it).to(device).clone()
    states, theta = gd.sample(design_fn=hip, design_guidance=guid, cond=[torch.from_numpy(g["state_0"]),
                              torch.from_numpy(g["bd_0"])], thetas_0=torch.from_numpy(g["thetas_0"]), bd_updater=hip.unet)
    assert (states.cpu() - torch.from_numpy(g[f"{tag}:states"])).abs().max() < 5e-3
    assert (theta.cpu() - torch.from_numpy(g[f"{tag}:theta"])).abs().max() < 5e-3


def test_ddim_and_unconditional_paths_run(env, dev):
    from diffphycon_amd.diffusion import diffusion_2d_jellyfish as DJ
    g, bd, design_fn, make, hip = env
    gd, guid = make("alpha")
    gd2 = DJ.GaussianDiffusion([gd.model_states, gd.model_thetas], image_size=16, frames=FR, cond_steps=1, timesteps=T,
                               sampling_timesteps=5, ddim_sampling_eta=1.0, loss_type="l2", eval_2ddpm=True, device=dev)
    cond = [torch.from_numpy(g["state_0"]), torch.from_numpy(g["bd_0"])]
    s, th = gd2.sample(design_fn=design_fn, design_guidance="standard-alpha", cond=cond,
                       thetas_0=torch.from_numpy(g["thetas_0"]), bd_updater=bd)
    assert s.shape == (2, FR, 3, 16, 16) and th.shape == (2, FR) and torch.isfinite(s).all()
    gd3 = DJ.GaussianDiffusion([gd.model_states, gd.model_thetas], image_size=16, frames=FR, cond_steps=0, timesteps=4,
                               loss_type="l2", eval_2ddpm=True, device=dev)
    s, th = gd3.sample(design_fn=design_fn, design_guidance="standard-alpha", cond=cond,
                       thetas_0=torch.from_numpy(g["thetas_0"]), bd_updater=bd)
    assert torch.isfinite(s).all() and torch.isfinite(th).all()
