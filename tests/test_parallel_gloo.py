"""world_size-2 gloo test of the N>1 path (CPU): contiguous batch sharding, metric gather order, max-over-ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, global_batch, q):
    from diffphycon_amd import parallel as P
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = P.shard_range(global_batch, rank, world)
    # a per-trajectory "metric row" that depends only on the GLOBAL trajectory index
    rows = torch.stack([torch.tensor([float(i), float(i) ** 2, 1.0 / (i + 1)], dtype=torch.float64) for i in range(a, b)])
    allrows = P.gather_metric_rows(rows)
    tmax = P.max_over_ranks(1.0 + rank, torch.device("cpu"))
    q.put((rank, (a, b), allrows.tolist(), tmax))      # plain lists: no shared-memory handles across exit
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    gb = 7                                   # ragged: 4 + 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 4) and res[1][1] == (4, 7)
    expect = torch.stack([torch.tensor([float(i), float(i) ** 2, 1.0 / (i + 1)], dtype=torch.float64) for i in range(gb)])
    for _, _, allrows, tmax in res:
        assert allrows == expect.tolist()       # same as the single-rank result, in global order
        assert tmax == 2.0


def test_shard_range_partitions_exactly():
    from diffphycon_amd.parallel import shard_range
    for gb in (0, 1, 8, 50, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
