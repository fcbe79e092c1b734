"""2-D jellyfish control inference with the reference's entry surface (inference/inference_2d_jellyfish.py, DDPM
method): same flags for the sampler, `reg_theta / force_fn / load_model / InferencePipeline.run_model_DDPM` structure,
running the two space-time U-Nets and (forward + backward) the two 2-D surrogates on libdpc (no autograd graph, no other backend).

Not carried over: the SAC / MPC baselines and the surrogate-simulator evaluation pipeline (`sim_ppl_2d`), which are
baselines/ evaluation code outside the sampling hot path.  Without `--synthetic True` the test split is read
from `--dataset_path` in the reference's on-disk layout (dataset/data_2d.py Jellyfish) and the four checkpoints are loaded;
`--synthetic True` (extra flag) fabricates initial states / boundaries / angles and random-initialises all four networks when
nothing is mounted under JELLYFISH_DATA_PATH; the normalisation constants then default to p in [-1, 1]."""
import argparse
import os
import pickle
import sys

import numpy as np
import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffphycon_amd.diffusion.diffusion_2d_jellyfish import ForceUnet, GaussianDiffusion, Trainer, Unet, reg_theta  # noqa: E402,F401
from diffphycon_amd.model.surrogates_hip import HipDesignGradient  # noqa: E402
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D  # noqa: E402
from diffphycon_amd import parallel  # noqa: E402
from filepath import JELLYFISH_DATA_PATH  # noqa: E402


def load_normalization(args):
    path = os.path.join(args.dataset_path, "train_data/normalization_max_min.pkl")
    if os.path.exists(path):
        nd = pickle.load(open(path, "rb"))
        args.p_max, args.p_min = nd["p_max"], nd["p_min"]
    elif args.synthetic:
        args.p_max, args.p_min = 1.0, -1.0
    else:
        raise FileNotFoundError(path)


def _ddpm(model, args, **kw):
    return GaussianDiffusion(
        model, image_size=args.image_size, frames=args.frames, cond_steps=args.cond_steps, timesteps=args.timesteps,
        sampling_timesteps=min(args.sampling_timesteps, args.timesteps), loss_type="l2", objective="pred_noise",
        backward_steps=args.backward_steps, backward_lr=args.backward_lr, standard_fixed_ratio=args.standard_fixed_ratio,
        forward_fixed_ratio=args.forward_fixed_ratio, coeff_ratio_J=args.coeff_ratio_J, coeff_ratio_w=args.coeff_ratio_w,
        only_vis_pressure=args.only_vis_pressure, device=args.device, **kw)


def load_model(args):
    if args.inference_method != "DDPM":
        raise NotImplementedError("the SAC / MPC baselines are out of scope (baselines/, sim_ppl_2d)")
    inp_dim = 5 if args.only_vis_pressure else 7
    out_dim = 2 if args.only_vis_pressure else 4
    model_joint = Unet3D_with_Conv3D(dim=64, out_dim=out_dim, dim_mults=(1, 2, 4), channels=inp_dim).to(args.device)
    diffusion_joint = _ddpm(model_joint, args, eval_2ddpm=False)
    model_thetas = Unet3D_with_Conv3D(dim=64, out_dim=1, dim_mults=(1, 2, 4), channels=inp_dim).to(args.device)
    diffusion_thetas = _ddpm(model_thetas, args, eval_2ddpm=False)
    # the reference builds both surrogates with dim = args.image_size (inference_2d_jellyfish.py:257-272) but ForceUnet's head is a
    # hard-coded Linear(512, .) (diffusion_2d_jellyfish.py:454): only dim = 64 runs.  --surrogate_dim (host-side extra) keeps the
    # released 64-wide surrogates at other resolutions (BASELINE.json's J128: 128 x 128 images); default = the reference's expression
    sdim = args.surrogate_dim or args.image_size
    force_model = ForceUnet(dim=sdim, out_dim=1, dim_mults=(1, 2, 4, 8), channels=4)
    bd_updater = Unet(dim=sdim, out_dim=3, dim_mults=(1, 2, 4, 8), channels=3)
    if not args.synthetic:
        Trainer(diffusion_joint, results_path=args.diffusion_joint_model_path).load(args.diffusion_joint_checkpoint)
        Trainer(diffusion_thetas, results_path=args.diffusion_w_model_path).load(args.diffusion_w_checkpoint)
        force_model.load_state_dict(torch.load(args.force_model_checkpoint, map_location="cpu"))
        bd_updater.load_state_dict(torch.load(args.boundary_updater_model_checkpoint, map_location="cpu"))
    force_model.to(args.device).eval()
    bd_updater.to(args.device).eval()
    for q in list(force_model.parameters()) + list(bd_updater.parameters()):      # checkpoint containers: inference only
        q.requires_grad_(False)
    diffusion = _ddpm([diffusion_joint.model, diffusion_thetas.model], args, eval_2ddpm=True, w_prob_exp=args.w_prob_exp,
                      use_guidance_in_model_predictions=args.use_guidance_in_model_predictions)

    # `force_fn` (:85-114): forward + input-gradient backward of both surrogates on libdpc (csrc/surr.hip), no autograd graph
    design_fn = HipDesignGradient(force_model, bd_updater, args)
    bd_updater = design_fn.unet

    return force_model, diffusion, bd_updater, design_fn


def pad_data(state_0, bd_mask_offset_0, image_size):
    """inference_2d_jellyfish.py:328-340: fields stored at (image_size - 2)^2 are centred in a zero image_size^2 frame."""
    def pad(x):
        if x.shape[2] == image_size:
            return x
        assert x.shape[1] == 3 and x.shape[2] == image_size - 2 and x.shape[3] == image_size - 2
        out = torch.zeros(x.shape[0], 3, image_size, image_size)
        out[:, :, 1:-1, 1:-1] = x
        return out
    return pad(state_0), pad(bd_mask_offset_0)


def dataset_batches(args):
    """load_data (:841-855) + the unpacking of run (:808-811): the Jellyfish test split in the reference's on-disk layout."""
    from diffphycon_amd.dataset.data_2d import Jellyfish
    ds = Jellyfish(dataset="jellyfish", dataset_path=args.dataset_path, time_steps=40, steps=args.frames, time_interval=1,
                   is_train=False, is_testdata=args.is_testdata, only_vis_pressure=args.only_vis_pressure)
    loader = torch.utils.data.DataLoader(ds, batch_size=args.batch_size, shuffle=False, pin_memory=True, num_workers=0)
    print("number of batch in test_loader: ", len(loader))
    for state_0, thetas_0, bd_0, sim_id, _ in loader:
        state_0, bd_0 = pad_data(state_0, bd_0, args.image_size)
        yield sim_id, state_0, bd_0, thetas_0


class InferencePipeline(object):
    def __init__(self, model, args=None, results_path=None, args_general=None):
        self.model, self.args, self.results_path, self.args_general = model, args, results_path, args_general
        for sub in ("", "thetas", "states"):
            os.makedirs(os.path.join(results_path, sub), exist_ok=True)

    def save(self, sim_id, pred, results_path):
        pred_states, pred_thetas = pred
        for index in range(sim_id.shape[0]):
            i = int(sim_id[index])
            np.save(os.path.join(results_path, "thetas", f"{i}.npy"), pred_thetas[index].cpu().numpy())
            np.save(os.path.join(results_path, "states", f"{i}.npy"), pred_states[index].cpu().numpy())

    def run_model_DDPM(self, state_0, bd_0, thetas_0):
        return self.model.sample(design_fn=self.args["design_fn"], design_guidance=self.args["design_guidance"],
                                 cond=[state_0, bd_0], thetas_0=thetas_0, bd_updater=self.args["bd_updater"])

    def run(self, batches):
        objs = []
        rank, world = getattr(self.args_general, "rank", 0), getattr(self.args_general, "world_size", 1)
        for sim_id, state_0, bd_0, thetas_0 in batches:
            ids = [int(v) for v in sim_id]
            assert ids == list(range(ids[0], ids[0] + len(ids))), "batches must hold consecutive simulation ids"
            # one process per GPU: rank r samples simulations [a, b) of the batch; the noise is keyed by the global simulation
            # id (Philox), so a simulation's sample does not depend on batch size or sharding, and consecutive batches differ
            a, b = parallel.shard_range(len(ids), rank, world)
            self.model.traj_offset, self.model.noise_epoch = ids[0] + a, 0
            states, thetas = self.run_model_DDPM(state_0[a:b], bd_0[a:b], thetas_0[a:b])
            if world > 1:            # final gather only (RCCL all_gather of the sampled states / angles)
                shp = states.shape[1:]
                states = parallel.gather_metric_rows(states.reshape(b - a, -1)).reshape(len(ids), *shp)
                thetas = parallel.gather_metric_rows(thetas.reshape(b - a, -1)).reshape(len(ids), -1)
            if rank == 0:
                self.save(sim_id, (states, thetas), self.results_path)
            R = reg_theta(thetas)
            print(f"batch ids {sim_id.tolist()}: theta range [{thetas.min().item():.3f}, {thetas.max().item():.3f}], "
                  f"R(theta) mean {R.mean().item():.4e}")
            objs.append(R.mean().item())
        print("Final results! mean R(theta):", float(np.mean(objs)))
        return objs


def synthetic_batches(args):
    """SURVEY.md 8(d) J recipe: state_0 ~ U(-1,1), bd_0 = elliptic mask + offsets in [-1,1], theta_0 ~ U(0.2, 0.9)."""
    g = torch.Generator().manual_seed(args.seed)
    s = args.image_size
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, s), torch.linspace(-1, 1, s), indexing="ij")
    for b in range(args.num_batches):
        B = args.batch_size
        state_0 = torch.rand(B, 3, s, s, generator=g) * 2 - 1
        mask = (((xx - 0.3) / 0.35) ** 2 + (yy / 0.6) ** 2 < 1).float() + (((xx + 0.3) / 0.35) ** 2 + (yy / 0.6) ** 2 < 1).float()
        bd_0 = torch.stack((mask.clamp(max=1).expand(B, -1, -1), torch.rand(B, s, s, generator=g) * 2 - 1,
                            torch.rand(B, s, s, generator=g) * 2 - 1), dim=1)
        thetas_0 = torch.rand(B, generator=g) * 0.7 + 0.2
        yield torch.arange(b * B, (b + 1) * B), state_0, bd_0, thetas_0


def build_parser():
    p = argparse.ArgumentParser(description="inference 2d inverse design model")
    p.add_argument("--dataset", default="jellyfish", type=str)
    p.add_argument("--dataset_path", default=JELLYFISH_DATA_PATH, type=str)
    p.add_argument("--only_vis_pressure", action="store_true")
    p.add_argument("--use_guidance_in_model_predictions", action="store_true")
    p.add_argument("--batch_size", default=10, type=int)
    p.add_argument("--num_batches", default=20, type=int)
    p.add_argument("--frames", default=20, type=int)
    p.add_argument("--backward_steps", default=5, type=int)
    p.add_argument("--backward_lr", default=0.01, type=float)
    p.add_argument("--standard_fixed_ratio", default=0.003, type=float)
    p.add_argument("--forward_fixed_ratio", default=0.01, type=float)
    p.add_argument("--coeff_ratio", default=0.1, type=float)
    p.add_argument("--coeff_ratio_J", default=0.3, type=float)
    p.add_argument("--coeff_ratio_w", default=0.3, type=float)
    p.add_argument("--reg_ratio", default=1000, type=float)
    p.add_argument("--cond_steps", default=1, type=int)
    p.add_argument("--diffusion_joint_model_path", default=os.path.join(JELLYFISH_DATA_PATH, "checkpoints"), type=str)
    p.add_argument("--diffusion_w_model_path", default=os.path.join(JELLYFISH_DATA_PATH, "checkpoints"), type=str)
    p.add_argument("--diffusion_joint_checkpoint", default=100, type=int)
    p.add_argument("--diffusion_w_checkpoint", default=50, type=int)
    p.add_argument("--sampling_timesteps", default=1000, type=int)
    p.add_argument("--image_size", type=int, default=64)
    p.add_argument("--inference_result_path", default="./results_jellyfish/", type=str)
    # the two flags the reference's DDPM branch also takes (inference_2d_jellyfish.py:916-919): results are written under
    # --inference_result_subpath (reference :835; there it is always re-derived as <inference_result_path>/<timestamp>_coeff_ratio_w_.._J.._ ,
    # :956-959 -- here an explicit value is honoured and the default is the result path itself, so reruns overwrite instead of piling up);
    # --log_path is the Trainer's log folder (reference :167), only created here: sampling writes no training logs
    p.add_argument("--inference_result_subpath", default=None, type=str)
    p.add_argument("--log_path", default=None, type=str)
    p.add_argument("--design_guidance", default="standard-alpha", type=str)
    p.add_argument("--inference_method", default="DDPM", type=str)
    p.add_argument("--force_model_checkpoint", type=str,
                   default=os.path.join(JELLYFISH_DATA_PATH, "checkpoints/force_surrogate_model/force_model_epoch_9.pth"))
    p.add_argument("--boundary_updater_model_checkpoint", type=str,
                   default=os.path.join(JELLYFISH_DATA_PATH, "checkpoints/boundary_updater/boundary_updater_epoch_9.pth"))
    p.add_argument("--gpu", type=int, default=0)
    p.add_argument("--w_prob_exp", type=float, default=0.7)
    # extra (not in the reference)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--synthetic", default=False, type=eval)
    p.add_argument("--surrogate_dim", default=None, type=int, help="width of the two 2-D surrogates (default: image_size, as the reference)")
    p.add_argument("--timesteps", default=1000, type=int, help="debug: shorter diffusion chain")
    p.add_argument("--is_testdata", default=True, type=bool, help="50-simulation test split (reference default)")
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    assert torch.cuda.is_available(), "the HIP path needs a GPU"
    args.rank, args.world_size = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    args.device = torch.device("cuda", int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count() if args.world_size > 1 else args.gpu)
    torch.cuda.set_device(args.device)
    if args.world_size > 1:                                   # torchrun, one rank per GPU; "nccl" = RCCL over xGMI
        parallel.init_process_group(args.rank, args.world_size, args.device)
    torch.manual_seed(args.seed)
    if not args.synthetic and not os.path.isdir(os.path.join(args.dataset_path, "test_data")):
        # fail before any checkpoint is read
        raise FileNotFoundError(f"no Jellyfish test split under {args.dataset_path}/test_data; mount it or pass --synthetic True")
    load_normalization(args)
    if args.inference_result_subpath is None:
        args.inference_result_subpath = args.inference_result_path
    if args.log_path and args.rank == 0:
        os.makedirs(args.log_path, exist_ok=True)
    force_model, diffusion, bd_updater, design_fn = load_model(args)
    ppl = InferencePipeline(diffusion, {"design_fn": design_fn, "design_guidance": args.design_guidance,
                                        "bd_updater": bd_updater}, results_path=args.inference_result_subpath, args_general=args)
    ppl.run(synthetic_batches(args) if args.synthetic else dataset_batches(args))
