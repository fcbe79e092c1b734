"""Batch sharding over ranks (one process per GPU; `torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  Trajectories are independent, so the only exchanges are the final
gather of per-trajectory metric rows and the max-over-ranks of the elapsed time; nothing is exchanged inside the
sampling loop."""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """Contiguous split of trajectories [0, global_batch): rank r owns [start, stop). Remainders go to the low ranks."""
    base, rem = divmod(global_batch, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_process_group(rank, world_size, device):
    """One process per GPU.  Backend "nccl" (= RCCL over xGMI on ROCm) unless DPC_DIST_BACKEND says otherwise ("gloo" lets
    the sharded entry scripts be exercised with several ranks on ONE GPU: RCCL refuses two ranks on the same device)."""
    import os
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("DPC_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world_size)


def _no_exchange(force):
    """True when a helper may skip its collective: no process group, or a group of one (unless `force`: tests/test_gpu_rccl.py runs
    the real collectives through RCCL on the one GPU a test box has)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and not force


def gather_metric_rows(rows, force=False):
    """rows: [B_local, M] per rank (B_local may differ) -> [B_global, M] on every rank, in global trajectory order."""
    if _no_exchange(force):
        return rows
    if dist.get_backend() == "gloo" and rows.is_cuda:          # gloo gathers host tensors
        return gather_metric_rows(rows.cpu()).to(rows.device)
    world = dist.get_world_size()
    n = torch.tensor([rows.shape[0]], dtype=torch.long, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    pad = max(counts)
    buf = torch.zeros((pad,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    buf[: rows.shape[0]] = rows
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0)


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def allreduce_sum_(flat, force=False):
    """SUM of the flat gradient buffer over the ranks, in place; returns the world size (the caller folds the 1 / world of DDP's
    gradient averaging into the optimizer kernel's scale).  The training step's only collective (Trainer.train :1025: accelerate's
    DDP all-reduce in the reference): ONE call on ONE contiguous buffer (~92 MB at dim 64, (1, 2, 4)) -- a single large ring
    all-reduce is what the point-to-point xGMI links want, instead of DDP's 25 MB buckets.  Every rank receives the same bits,
    so the optimizer keeps the replicas bit-identical."""
    if _no_exchange(force):
        return 1
    if dist.get_backend() == "gloo" and flat.is_cuda:          # gloo reduces host tensors (several ranks on ONE GPU in the tests)
        h = flat.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        flat.copy_(h)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return dist.get_world_size()


def max_over_ranks(seconds, device):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()
