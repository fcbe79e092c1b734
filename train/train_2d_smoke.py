"""Training entry of the smoke denoisers with the reference's surface (/root/reference/train/train_2d_smoke.py: same flags,
same model / diffusion / Trainer construction :31-74), running on libdpc: p_losses forward + hand-written backward + fused
clip / Adam / EMA (diffphycon_amd.diffusion.diffusion_2d_smoke.Trainer).

    python train/train_2d_smoke.py [--is_w_model] [--batch_size 6] [--train_num_steps 200000]
    torchrun --nproc-per-node 8 train/train_2d_smoke.py ...        # one rank per GPU, flat-gradient all-reduce over RCCL

Extra flags (host-side only): --synthetic True fabricates the training split when no dataset is mounted (SyntheticSmoke);
--bwd_mode x6|f16x3|f32 and --loss_scale select the arithmetic of the backward-data convolutions."""
import argparse
import os
import sys

import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffphycon_amd import parallel  # noqa: E402
from diffphycon_amd.dataset.data_2d import Smoke, SyntheticSmoke  # noqa: E402
from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion, Trainer  # noqa: E402
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D  # noqa: E402
from filepath import SMOKE_DATA_PATH, SMOKE_RESULTS_PATH  # noqa: E402


def build_parser():
    parser = argparse.ArgumentParser(description="Train EBM model")
    parser.add_argument("--dataset", default="Smoke", type=str, help="dataset to evaluate")
    parser.add_argument("--dataset_path", default=SMOKE_DATA_PATH, type=str, help="path to dataset")
    parser.add_argument("--batch_size", default=6, type=int, help="size of batch of input to use")
    parser.add_argument("--train_num_steps", default=200000, type=int, help="total training steps")
    parser.add_argument("--is_w_model", action="store_true", help="whether to train w model")
    parser.add_argument("--results_path", default=os.path.join(SMOKE_RESULTS_PATH, "checkpoints"), type=str,
                        help="folder to save training checkpoints")
    # host-side extras
    parser.add_argument("--synthetic", default=False, type=lambda s: str(s).lower() in ("1", "true", "yes"))
    parser.add_argument("--bwd_mode", default="f16x3", choices=["x6", "f16x3", "f32"])
    parser.add_argument("--loss_scale", default=None, type=lambda v: v if v == "dynamic" else float(v),
                        help="power of two, or 'dynamic' (start 2^20, halve on overflow, double every 2000 clean steps); default: dynamic "
                             "for --bwd_mode f16x3, 1 for the exact modes")
    parser.add_argument("--save_and_sample_every", default=10000, type=int)
    parser.add_argument("--image_size", default=64, type=int)
    parser.add_argument("--seed", default=0, type=int)
    return parser


def synthetic_loader(batch, size, seed):
    ds = SyntheticSmoke(n_simu=64, size=size, seed=seed, is_train=True)
    g = torch.Generator().manual_seed(seed)
    while True:
        idx = torch.randint(0, len(ds), (batch,), generator=g).tolist()
        state = torch.stack([ds[i][0] for i in idx])
        # the synthetic split only carries the initial density: add structure so that the loss has something to fit
        state = state + 0.25 * torch.randn(state.shape, generator=g)
        yield state, torch.tensor(idx)


def main(argv=None):
    FLAGS = build_parser().parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local % max(torch.cuda.device_count(), 1))      # (several gloo ranks may share one GPU in the tests)
    torch.cuda.set_device(device)
    if world > 1:
        parallel.init_process_group(rank, world, device)
    torch.manual_seed(FLAGS.seed + rank)                      # per-rank t / noise streams; weights are seeded identically below
    if rank == 0:
        print(FLAGS)
    channels = 6 if not FLAGS.is_w_model else 2
    init_state = torch.random.get_rng_state()
    torch.manual_seed(FLAGS.seed)                             # identical initial replicas on every rank (DDP broadcasts rank 0's)
    model = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=channels)
    torch.random.set_rng_state(init_state)
    if rank == 0:
        print("number of parameters Unet3D_with_Conv3D: ", sum(p.numel() for p in model.parameters()))
    results_path = os.path.join(FLAGS.results_path, "w" if FLAGS.is_w_model else "joint")
    if rank == 0:
        print("Saved at: ", results_path)
    diffusion = GaussianDiffusion(model, image_size=FLAGS.image_size, frames=32, timesteps=1000, sampling_timesteps=250,
                                  loss_type="l2", objective="pred_noise", device=device)
    per_rank = max(1, FLAGS.batch_size // world)
    data = synthetic_loader(per_rank, FLAGS.image_size, FLAGS.seed + 1000 * rank) if FLAGS.synthetic else None
    if not FLAGS.synthetic:
        _, _ = Smoke(dataset_path=FLAGS.dataset_path, is_train=True)[0]       # (:36-40: fails early when the dataset is absent)
    trainer = Trainer(diffusion, FLAGS.dataset, FLAGS.dataset_path, train_batch_size=FLAGS.batch_size, train_lr=1e-3,
                      train_num_steps=FLAGS.train_num_steps, gradient_accumulate_every=1, ema_decay=0.995,
                      save_and_sample_every=FLAGS.save_and_sample_every, results_path=results_path, amp=False,
                      is_w_model=FLAGS.is_w_model, bwd_mode=FLAGS.bwd_mode, loss_scale=FLAGS.loss_scale, data=data)
    trainer.train()
    if world > 1:
        import torch.distributed as dist
        w = trainer._t.w
        probe = torch.stack((w.double().sum(), (w.double() ** 2).sum())).cpu()
        lo, hi = probe.clone(), probe.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if rank == 0:
            print("replicas identical:", bool(torch.equal(lo, hi)))
        dist.barrier()
        dist.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main()
