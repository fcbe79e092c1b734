"""1-D Burgers control inference with the reference's entry surface (inference/inference_1d_burgers.py): same flags
and the same `get_loss_fn_2dconv / get_nablaJ_2dconv / load_2dconv_model* / diffuse_2dconv / get_scheduler / evaluate`
structure, running on libdpc (HIP).  `--synthetic True` (extra flag) replaces the HDF5 test split and the checkpoints by
seeded synthetic targets and random-initialised U-Nets; everything else follows the reference line by line."""
import argparse
import copy
import os
import sys

import numpy as np
import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffphycon_amd.diffusion.diffusion_1d_burgers import (Trainer, cosine_beta_J_schedule, get_nablaJ,  # noqa: E402
                                                           plain_cosine_schedule, sigmoid_schedule, sigmoid_schedule_flip)
from diffphycon_amd.evaluators import burgers_numeric_solve_free  # noqa: E402
from diffphycon_amd.utils_burgers import (mse_deviation, mse_dist_reg, ddpm_guidance_loss, burgers_metric,  # noqa: E402
                                          get_target, get_2d_ddpm)
from diffphycon_amd import parallel  # noqa: E402
from diffphycon_amd.diffusion.diffusion_1d_burgers import BurgersGuidance  # noqa: E402

none_or_str = lambda x: None if x == "None" else x  # noqa: E731
RESCALER = 10

parser = argparse.ArgumentParser(description="Eval EBM model")
parser.add_argument("--exp_id", type=str)
parser.add_argument("--model_str_in_key", default="", type=str)
parser.add_argument("--save_file", default="burgers_results/result_zerowf.yaml", type=str)
parser.add_argument("--dataset", default="free_u_f_1e5", type=str)
parser.add_argument("--model_str", default="", type=str)
parser.add_argument("--n_test_samples", default=50, type=int)
parser.add_argument("--partial_control", default="full", type=none_or_str)
parser.add_argument("--partially_observed", default=None, type=none_or_str)
parser.add_argument("--train_on_partially_observed", default=None, type=none_or_str)
parser.add_argument("--set_unobserved_to_zero_during_sampling", default=False, type=eval)
parser.add_argument("--checkpoint", default=10, type=int)
parser.add_argument("--checkpoint_interval", default=10000, type=int)
parser.add_argument("--train_num_steps", default=100000, type=int)
parser.add_argument("--using_ddim", default=False, type=eval)
parser.add_argument("--ddim_eta", default=0., type=float)
parser.add_argument("--ddim_sampling_steps", default=1000, type=int)
parser.add_argument("--J_scheduler", default=None, type=str)
parser.add_argument("--recurrence", default=False, type=eval)
parser.add_argument("--recurrence_k", default=1, type=int)
parser.add_argument("--wfs", nargs="+", default=[0], type=float)
parser.add_argument("--wus", nargs="+", default=[0], type=float)
parser.add_argument("--wreg", default=0, type=float)
parser.add_argument("--wpinns", nargs="+", default=[0], type=float)
parser.add_argument("--pinn_loss_mode", default="mean", type=str)
parser.add_argument("--condition_on_residual", default=None, type=str)
parser.add_argument("--residual_on_u0", default=False, type=eval)
parser.add_argument("--is_condition_u0", default=False, type=eval)
parser.add_argument("--is_condition_uT", default=False, type=eval)
parser.add_argument("--is_condition_u0_zero_pred_noise", default=True, type=eval)
parser.add_argument("--is_condition_uT_zero_pred_noise", default=True, type=eval)
parser.add_argument("--dim", default=64, type=int)
parser.add_argument("--resnet_block_groups", default=1, type=int)
parser.add_argument("--dim_muls", nargs="+", default=[1, 2, 4, 8], type=int)
parser.add_argument("--is_model_w", default=False, type=eval)
parser.add_argument("--eval_two_models", default=False, type=eval)
parser.add_argument("--expand_condition", default=False, type=eval)
parser.add_argument("--prior_beta", default=1, type=float)
parser.add_argument("--normalize_beta", default=False, type=eval)
parser.add_argument("--w_scheduler", default=None, type=none_or_str)
parser.add_argument("--exp_id__model_w", type=str)
parser.add_argument("--checkpoint__model_w", default=10, type=int)
parser.add_argument("--checkpoint_interval__model_w", default=10000, type=int)
parser.add_argument("--train_num_steps__model_w", default=100000, type=int)
parser.add_argument("--dim__model_w", default=64, type=int)
parser.add_argument("--resnet_block_groups__model_w", default=1, type=int)
parser.add_argument("--dim_muls__model_w", nargs="+", default=[1, 2, 4, 8], type=int)
# extra (not in the reference)
parser.add_argument("--synthetic", default=False, type=eval, help="synthetic targets + random-init U-Nets")
parser.add_argument("--batch_size", default=50, type=int, help="trajectories per sample() call (reference: 50)")
parser.add_argument("--timesteps_override", default=None, type=int, help="debug: shorter diffusion chain")


def get_loss_fn_2dconv(wf=0, wu=0, wpinn=0, target_i=0, wu_eval=1, wf_eval=0, device=0, dataset="free_u_f_1e5",
                       dist_reg=lambda x: 0, wreg=0, partially_observed=None, pinn_loss_mode="mean", synthetic=False):
    """(:129-165) -> closed-form guidance on the rescaled target (u_target / RESCALER)."""
    u_target = get_target(target_i, device=device, dataset=dataset, synthetic=synthetic,
                          partially_observed_fill_zero_unobserved=partially_observed)
    return ddpm_guidance_loss(u_target / RESCALER, wu=wu, wf=wf, wpinn=wpinn, dist_reg=dist_reg,
                              pinn_loss_mode=pinn_loss_mode, wreg=wreg, partially_observed=partially_observed)


def get_nablaJ_2dconv(**kwargs):
    return get_nablaJ(get_loss_fn_2dconv(**kwargs))


def use_args_w(args):
    args = copy.deepcopy(args)
    key = "__model_w"
    for k in list(args.__dict__.keys()):
        if key in k:
            setattr(args, k[:-len(key)], getattr(args, k))
    return args


def _load(ddpm, folder, args, checkpoint):
    if not args.synthetic:
        Trainer(ddpm, None, results_folder=folder, train_num_steps=args.train_num_steps,
                save_and_sample_every=args.checkpoint_interval).load(checkpoint)
    return ddpm


def load_2dconv_model_two_ddpm(i, args):
    args = copy.deepcopy(args)
    args.is_ddpm_w, args.eval_two_models = False, False
    ddpm_uw = _load(get_2d_ddpm(args), f"./trained_models/burgers/{args.exp_id}/", args, args.checkpoint)
    unet_uw = ddpm_uw.model
    args.is_ddpm_w = True
    args_w = use_args_w(args)         # the reference REBINDS args here (:199), so the prior model is loaded with ITS milestone
    ddpm_w = _load(get_2d_ddpm(args_w), f"./trained_models/burgers_w/{args.exp_id__model_w}/", args_w, args_w.checkpoint)
    unet_w = ddpm_w.model
    args.eval_two_models, args.is_ddpm_w = True, False
    args.unet_uw, args.unet_w = unet_uw, unet_w
    return get_2d_ddpm(args).cuda()


def load_2dconv_model(i, args, new=True):
    if args.eval_two_models:
        assert not args.is_model_w
        return load_2dconv_model_two_ddpm(i, args)
    if args.is_model_w:
        raise NotImplementedError("sampling from the prior model alone is not used by the DiffPhyCon scripts")
    return _load(get_2d_ddpm(args), f"./trained_models/burgers/{i}/", args, args.checkpoint).cuda()


def diffuse_2dconv(args, custom_metric, model_i, seed=0, ret_ls=False, **kwargs):
    """(:248-303) sample, re-simulate the sampled forcing with the finite-difference solver, score."""
    u_from_x = lambda x: x[:, 0, :11, :]     # noqa: E731
    u0_from_x = lambda x: x[:, 0, 0, :]      # noqa: E731
    f_from_x = lambda x: x[:, 1, :10, :]     # noqa: E731
    torch.manual_seed(seed)
    ddpm = load_2dconv_model(model_i, args)
    ddpm.noise_seed = seed
    # one process per GPU: rank r samples trajectories [a, b) of this batch (counter-based noise keyed by the index inside the
    # batch, guidance normalised by the WHOLE batch), then the tiny samples [B, 2, 16, 128] are gathered once (RCCL) and every
    # rank scores the full batch exactly as a single rank would -- no exchange inside the sampling loop
    rank, world = getattr(args, "rank", 0), getattr(args, "world_size", 1)
    if world > 1:
        B = kwargs["batch_size"]
        a, b = parallel.shard_range(B, rank, world)
        kw = dict(kwargs)
        kw["batch_size"] = b - a
        kw["u_init"], kw["u_final"] = kwargs["u_init"][a:b], kwargs["u_final"][a:b]
        g = kwargs.get("nablaJ")
        if g is not None:
            kw["nablaJ"] = BurgersGuidance(g.u_target[a:b], g.wu, g.wf, g.wreg, g.partially_observed)
        ddpm.traj_offset, ddpm.guidance_batch = a, B
        x_local = ddpm.sample(**kw)
        x = parallel.gather_metric_rows(x_local.reshape(b - a, -1)).reshape(B, *x_local.shape[1:]) * RESCALER
    else:
        x = ddpm.sample(**kwargs) * RESCALER
    x_gt = burgers_numeric_solve_free(u0_from_x(x), f_from_x(x), visc=0.01, T=1.0, dt=1e-4, num_t=10)
    ddpm_mse = mse_deviation(u_from_x(x), x_gt, partially_observed=args.partially_observed).cpu()
    J_diffused, _ = custom_metric(f_from_x(x), diffused_u=u_from_x(x), evaluate_u=True)
    J_actual, energy = custom_metric(f_from_x(x))
    to_np = lambda v: v.cpu().numpy() if type(v) is not tuple else np.array([vi.cpu().numpy() for vi in v])  # noqa: E731
    return ddpm_mse, to_np(J_diffused), to_np(J_actual), energy.cpu().numpy()


def get_scheduler(scheduler):
    if scheduler is None:
        return None
    if scheduler == "linear":
        raise NotImplementedError
    if scheduler == "cosine":
        return cosine_beta_J_schedule
    if scheduler == "plain_cosine":
        return plain_cosine_schedule
    if scheduler == "sigmoid":
        return sigmoid_schedule
    if scheduler == "sigmoid_flip":
        return sigmoid_schedule_flip
    raise ValueError(f"Unknown scheduler: {scheduler}")


def evaluate(model_i, args, wu=0, wf=0, wpinn=0, wf_eval=0, wu_eval=1, conv2d=True):
    n_test_samples, batch_size = args.n_test_samples, args.batch_size
    assert n_test_samples % batch_size == 0
    l_gts, energies = [], []
    for i in range(n_test_samples // batch_size):
        target_idx = list(range(i * batch_size, (i + 1) * batch_size))
        tgt = lambda **kw: get_target(target_idx, dataset=args.dataset, synthetic=args.synthetic, **kw)   # noqa: E731
        tgt_po = tgt(partially_observed_fill_zero_unobserved=args.partially_observed)
        _, _, J_actual, energy = diffuse_2dconv(
            args,
            custom_metric=lambda f, **kw: burgers_metric(tgt(), f, target="final_u", partial_control=args.partial_control,
                                                        report_all=True, partially_observed=args.partially_observed, **kw),
            model_i=model_i, seed=i,
            nablaJ=get_nablaJ_2dconv(target_i=target_idx, wu=wu, wf=wf, wpinn=wpinn, wf_eval=wf_eval, wu_eval=wu_eval,
                                     dist_reg=mse_dist_reg, wreg=args.wreg, dataset=args.dataset,
                                     partially_observed=args.partially_observed, pinn_loss_mode=args.pinn_loss_mode,
                                     synthetic=args.synthetic),
            J_scheduler=get_scheduler(args.J_scheduler), w_scheduler=get_scheduler(args.w_scheduler),
            clip_denoised=True, guidance_u0=True, batch_size=batch_size,
            u_init=tgt_po[:, 0, :] / RESCALER, u_final=tgt_po[:, 10, :] / RESCALER)
        l_gts.append(J_actual)
        energies.append(energy)
        print("J_actual:", l_gts[0][0].mean())
        print("Energy:", energies[0].mean())
    return l_gts, energies


if __name__ == "__main__":
    args = parser.parse_args()
    assert torch.cuda.is_available(), "the HIP path needs a GPU"
    args.rank, args.world_size = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    if args.world_size > 1:                                   # torchrun, one rank per GPU; "nccl" = RCCL over xGMI
        parallel.init_process_group(args.rank, args.world_size, torch.device("cuda", torch.cuda.current_device()))
    if args.timesteps_override:
        import diffphycon_amd.utils_burgers as ub
        _orig = ub.GaussianDiffusion
        ub.GaussianDiffusion = lambda *a, **k: _orig(*a, **{**k, "timesteps": args.timesteps_override,
                                                             "sampling_timesteps": args.timesteps_override})
    results = evaluate(model_i=args.exp_id, args=args, conv2d=True)
