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


def _skip_worker(rank, world, port, out_dir):
    """The skip protocol of the dynamic loss scale (Trainer.optimizer_step) on the product's collective: rank 1's gradient overflowed in
    step 0 -- dpc_train_range_poison would have written +inf into its g[0] --, the SUM all-reduce carries it to rank 0, both ranks see a
    non-finite norm, both skip and halve; step 1 is clean on both and is applied."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffphycon_amd import parallel
    from diffphycon_amd.diffusion.diffusion_2d_smoke import loss_scale_update
    torch.manual_seed(7)                                  # same "weights" on both ranks
    w = torch.randn(1000)
    scale, good, log = 2.0 ** 20, 0, []
    for step in range(3):
        g = torch.full((1000,), 1e-3 * (rank + 1) * scale)
        if step == 0 and rank == 1:
            g[0] = float("inf")                           # what the poison kernel does on the rank that overflowed
        n = parallel.allreduce_sum_(g)
        norm = float(torch.linalg.vector_norm(g / (scale * n)))
        apply, scale, good = loss_scale_update(scale, good, norm, growth_interval=2)
        if apply:
            w -= 0.1 * g / (scale if step != 2 else scale / 2) / n      # (step 2 grew the scale AFTER its gradients were formed)
        log.append((apply, scale, good))
    torch.save({"w": w, "log": log}, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


def test_overflow_on_one_rank_skips_the_step_on_every_rank(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_skip_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    assert a["log"] == b["log"] == [(False, 2.0 ** 19, 0), (True, 2.0 ** 19, 1), (True, 2.0 ** 20, 2)]
    assert torch.equal(a["w"], b["w"])


def test_loss_scale_rule():
    sys.path.insert(0, ROOT)
    from diffphycon_amd.diffusion.diffusion_2d_smoke import loss_scale_update
    assert loss_scale_update(2.0 ** 20, 5, float("nan")) == (False, 2.0 ** 19, 0)
    assert loss_scale_update(2.0 ** 20, 1998, 0.5) == (True, 2.0 ** 20, 1999)
    assert loss_scale_update(2.0 ** 20, 1999, 0.5) == (True, 2.0 ** 21, 2000)
    assert loss_scale_update(2.0 ** 24, 1999, 0.5) == (True, 2.0 ** 24, 2000)          # capped
    with pytest.raises(FloatingPointError):
        loss_scale_update(1.0, 0, float("inf"))
