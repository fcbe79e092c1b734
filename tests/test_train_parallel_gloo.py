"""Data-parallel training step at world size 2 on CPU (gloo): the flat-gradient all-reduce of diffphycon_amd.parallel
(Trainer.train :1025 -- accelerate's DDP averaging in the reference) reproduces the full-batch gradient, and the replicas stay
BIT-identical after the optimizer update.  The per-rank gradients and the Adam arithmetic come from the CPU oracle here (the
product's kernels need a GPU; tests/test_gpu_train.py runs the real Trainer under the same collective)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_golden
    from diffphycon_amd import parallel
    from oracle import train_smoke as T
    from oracle import unet3d as U
    g = load_golden("train_joint")
    cfg = U.Unet3DConfig(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=6)
    sd = {k[3:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("w0:")}
    sched = T.schedule(1000)
    names = sorted(sd)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(vv) for k, vv in sd.items()}
    for step in range(2):
        lo, hi = parallel.shard_range(2, rank, world)                    # split_batches: each rank takes its slice of the batch
        state = torch.from_numpy(g[f"s{step}:state"])[lo:hi]
        t = torch.from_numpy(g[f"s{step}:t"])[lo:hi]
        noise = torch.from_numpy(g[f"s{step}:noise"])[lo:hi]
        _, grads = T.loss_and_grads(sd, cfg, sched, state, t, noise)
        flat = torch.cat([grads[k].reshape(-1) if k in grads else torch.zeros(sd[k].numel()) for k in names])
        w = parallel.allreduce_sum_(flat)
        assert w == world
        flat /= w
        total = torch.linalg.vector_norm(flat)
        flat *= T.clip_coef(total, 1.0)
        o = 0
        for k in names:
            n = sd[k].numel()
            if k in grads:
                T.adam_step(sd[k], flat[o:o + n].view_as(sd[k]), m[k], v[k], step + 1, 1e-3)
            o += n
        if step == 0:
            torch.save({"grad": flat.clone(), "norm": total}, os.path.join(out_dir, f"g{rank}.pt"))
    torch.save(sd, os.path.join(out_dir, f"w{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_keeps_replicas_bit_identical(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    assert all(torch.equal(w0[k], w1[k]) for k in w0)                    # bit-equal replicas after two optimizer steps
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    assert torch.equal(g0["grad"], g1["grad"])
    # the averaged gradient is the reference's full-batch gradient (mean loss over B = 2 == mean of the two per-sample means)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    g = load_golden("train_joint")
    names = sorted(k[3:] for k in g.files if k.startswith("w0:"))
    ref = torch.cat([torch.from_numpy(g[f"s0:g:{k}"]).reshape(-1) for k in names])
    got = g0["grad"] / float(min(1.0, 1.0 / (float(g0["norm"]) + 1e-6)))            # undo the clip
    assert (got - ref).abs().max().item() < 2e-4 * ref.abs().max().item()
    assert abs(float(g0["norm"]) - float(g["s0:grad_norm"])) < 1e-4 * float(g["s0:grad_norm"])
    # and the two-rank weights follow the reference's single-process weights
    for k in names:
        d = (w0[k] - torch.from_numpy(g[f"s1:w:{k}"])).abs()
        live = torch.from_numpy(abs(g[f"s0:g:{k}"]) > 1e-4 * float(ref.abs().max()))
        assert d[live].numel() == 0 or d[live].max().item() < 3e-5, k
