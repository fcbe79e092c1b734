"""Burgers helpers of the reference's utils.py (:1188-1407) and train/train_1d_burgers.py:get_2d_ddpm (:113-168),
with the same names and argument meaning, on device tensors + libdpc."""
import os

import numpy as np
import torch

from .diffusion.diffusion_1d_burgers import BurgersGuidance, GaussianDiffusion
from .evaluators import burgers_numeric_solve_free
from .model.burgers_1d.unet import Unet2D


def mse_deviation(u1, u2, partially_observed=None, report_all=False):
    """utils.py:1188-1200."""
    if partially_observed is not None:
        u1, u2 = u1.clone(), u2.clone()
        if partially_observed == "front_rear_quarter":
            Nx = u1.shape[-1]
            idx = torch.cat((torch.arange(0, Nx // 4), torch.arange((3 * Nx) // 4, Nx))).to(u1.device)
            u1, u2 = u1[..., idx], u2[..., idx]
    if report_all:
        mse = (u1 - u2).square().mean((-1, -2))
        mae = (u1 - u2).abs().mean((-1, -2))
        ep = 1e-5
        return mse, mae, mse / (u2 + ep).square().mean(), mae / (u2 + ep).abs().mean()
    return (u1 - u2).square().mean((-1, -2))


def burgers_metric(u_target, f, target="final_u", partial_control="full", report_all=False, diffused_u=None,
                   evaluate_u=False, partially_observed=None,
                   solver=lambda u_target, f: burgers_numeric_solve_free(u_target[:, 0, :], f, visc=0.01, T=1.0, dt=1e-4, num_t=10),
                   **kwargs):
    """utils.py:1203-1284: zero the uncontrolled half of f, re-simulate with the HIP finite-difference solver, score."""
    if kwargs != {}:
        print("WARNING: kwargs", [k for k in kwargs.keys()], "are not used.")
    u_target, f = u_target.clone(), f.clone()
    assert len(u_target.size()) == len(f.size()) == 3
    if partial_control is None or partial_control == "full":
        pass
    elif partial_control == "front_rear_quarter":
        Nx = f.size(2)
        f[:, :, Nx // 4: (Nx * 3) // 4] = 0
    u_controlled = diffused_u.clone() if evaluate_u else solver(u_target, f)
    if partially_observed is not None:
        Nx = u_controlled.size(-1)
        if partially_observed == "front_rear_quarter":
            idx = torch.cat((torch.arange(0, Nx // 4), torch.arange((3 * Nx) // 4, Nx))).to(u_controlled.device)
            u_controlled, u_target = u_controlled[..., idx], u_target[..., idx]
        else:
            raise NotImplementedError
    if target != "final_u":
        raise ValueError("Undefined target to evaluate")
    d = u_controlled[:, -1, :] - u_target[:, -1, :]
    mse = d.square().mean(-1)
    mse_median, _ = d.square().median(-1)
    mae = d.abs().mean(-1)
    mae_median, _ = d.abs().median(-1)
    ep = 1e-5
    nmse = d.square().mean(-1) / (u_target[:, -1, :].square().mean() + ep)
    nmae = d.abs().mean(-1) / (u_target[:, -1, :].abs().mean() + ep)
    J_actual = mse if not report_all else (mse, mse_median, mae, mae_median, nmse, nmae)
    return J_actual, f.square().sum((-1, -2))


def mse_dist_reg(u):
    """utils.py:1286 (kept for signature parity; the HIP path folds it into BurgersGuidance.wreg)."""
    return (u[:, 1:, :] - u[:, :-1, :]).square().sum()


def ddpm_guidance_loss(u_target, u=None, f=None, wu=0, wf=0, wreg=0, wpinn=0, dist_reg=None, pinn_loss_mode="mean",
                       partially_observed=None):
    """utils.py:1289-1328 as a closed-form guidance object: `u_target` is the (already rescaled) target; `u`, `f`
    are ignored (the kernel differentiates the loss analytically)."""
    if wpinn != 0:
        raise NotImplementedError("wpinn != 0 raises NotImplementedError in the reference as well (utils.py:1321-1323)")
    return BurgersGuidance(u_target, wu=wu, wf=wf, wreg=wreg if dist_reg is not None else 0.0,
                           partially_observed=partially_observed)


def synthetic_targets(idx, device, seed=0):
    """Stand-in for the `free_u_f_1e5` test split: u0 = two Gaussian bumps
    (generate_burgers.py:361-372), uT = u0 rolled by 16 cells; rows 1..9 are zero (only rows 0 and 10 are read)."""
    idx = [idx] if isinstance(idx, int) else list(idx)
    xg = torch.linspace(0, 1, 128)
    out = torch.zeros(len(idx), 11, 128)
    for k, i in enumerate(idx):
        g = torch.Generator().manual_seed(seed * 100003 + int(i))
        r = torch.rand(6, generator=g)
        u0 = (2.0 * r[1]) * torch.exp(-0.5 * ((xg - (0.2 + 0.2 * r[0])) / (0.05 + 0.1 * r[2])) ** 2) \
            + (-2.0 * r[4]) * torch.exp(-0.5 * ((xg - (0.6 + 0.2 * r[3])) / (0.05 + 0.1 * r[5])) ** 2)
        out[k, 0], out[k, 10] = u0, torch.roll(u0, 16)
    return out.to(device)


def get_target(target_i, f=False, device=0, dataset="free_u_f_1e5", synthetic=False,
               partially_observed_fill_zero_unobserved=None, **dataset_kwargs):
    """utils.py:1353-1395.  Returns the UNRESCALED target states [B, 11, 128] (or the forces [B, 10, 128] with f=True) of the
    test split through `Burgers1D` (dataset/data_1d.py).  The split is an HDF5 file (dataset/apps/burgers_h5py.py:206-255): the
    file open needs h5py, which this image does not ship -- then `synthetic=True`, or a `dataset_cache=BurgersCache(...)`."""
    if isinstance(device, int):       # the reference's default `device=0`: here the rank's current GPU (one process per GPU)
        dev = torch.device("cuda", torch.cuda.current_device() if device == 0 and torch.cuda.is_available() else device)
    else:
        dev = device
    if synthetic:
        if f:
            raise NotImplementedError("synthetic targets carry no reference forces")
        u = synthetic_targets(target_i, dev)
    else:
        from .dataset.data_1d import Burgers1D
        # utils.py:1357-1370: the test split through Burgers1D with rescaler 1 (the file open needs h5py and says so if missing)
        ds = Burgers1D(dataset="burgers", input_steps=1, output_steps=10, time_interval=1, is_y_diff=False, split="test",
                       transform=None, pre_transform=None, verbose=False, root_path=f"data/{dataset}", device="cuda", rescaler=1,
                       nt_total=11, partially_observed_fill_zero_unobserved=partially_observed_fill_zero_unobserved,
                       **dataset_kwargs)                    # (dataset_cache=BurgersCache(...) bypasses the HDF5 file open)
        idx = [target_i] if isinstance(target_i, int) else list(target_i)
        rows = torch.stack(tuple(ds.get(i) for i in idx), dim=0)
        return (rows[:, 11:, :] if f else rows[:, :11, :]).to(dev)
    if partially_observed_fill_zero_unobserved == "front_rear_quarter":
        nx = u.shape[-1]
        u[..., nx // 4: (nx * 3) // 4] = 0
    elif partially_observed_fill_zero_unobserved is not None:
        raise ValueError("Unknown partially observed mode")
    return u


def get_2d_ddpm(args):
    """train/train_1d_burgers.py:113-168."""
    sim_time_stamps, sim_space_grids = 16, 128
    if getattr(args, "condition_on_residual", None) is not None or getattr(args, "expand_condition", False):
        raise NotImplementedError("condition_on_residual / expand_condition are not used by the inference scripts")
    if not args.eval_two_models:
        u_net = Unet2D(dim=args.dim, init_dim=None, out_dim=2, dim_mults=tuple(args.dim_muls), channels=2,
                       self_condition=False, resnet_block_groups=args.resnet_block_groups, learned_variance=False,
                       learned_sinusoidal_cond=False, random_fourier_features=False, learned_sinusoidal_dim=16,
                       sinusoidal_pos_emb_theta=10000, attn_dim_head=32, attn_heads=4)
    return GaussianDiffusion(
        u_net if not args.eval_two_models else (args.unet_uw, args.unet_w), seq_length=(sim_time_stamps, sim_space_grids),
        auto_normalize=False, use_conv2d=True, temporal=True, is_condition_u0=args.is_condition_u0,
        is_condition_uT=args.is_condition_uT, is_condition_u0_zero_pred_noise=args.is_condition_u0_zero_pred_noise,
        is_condition_uT_zero_pred_noise=args.is_condition_uT_zero_pred_noise,
        train_on_partially_observed=args.train_on_partially_observed,
        set_unobserved_to_zero_during_sampling=args.set_unobserved_to_zero_during_sampling,
        conditioned_on_residual=None, residual_on_u0=args.residual_on_u0, recurrence=args.recurrence,
        recurrence_k=args.recurrence_k, is_model_w=args.is_model_w, eval_two_models=args.eval_two_models,
        expand_condition=False, prior_beta=args.prior_beta, normalize_beta=getattr(args, "normalize_beta", False),
        sampling_timesteps=args.ddim_sampling_steps if args.using_ddim else 1000, ddim_sampling_eta=args.ddim_eta)
