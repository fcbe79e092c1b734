"""RCCL itself, on the one GPU a test box has: world size 1 with the **nccl** backend (= RCCL on ROCm; it allows one rank per device)
through the code the N-GPU runs use -- parallel.init_process_group(device_id=...), bench.Ctx.sync's barrier, gather_metric_rows,
max_over_ranks, allreduce_sum_ -- so that a broken device_id / environment (HSA_ENABLE_IPC_MODE_LEGACY, MASTER_ADDR) shows up here
and not on the first 8-GPU run.  Multi-rank semantics are covered by the gloo tests (tests/test_parallel_gloo.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, socket, time
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from diffphycon_amd import parallel
import bench
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.pop("DPC_DIST_BACKEND", None)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
parallel.init_process_group(0, 1, dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
ctx = bench.Ctx(0, 1, dev, dist)
ctx.sync()                                                    # device sync + NCCL barrier + device sync
hi, lo = ctx.reduce(0.25)
assert hi == 0.25 and lo == 0.25
rows = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
# world 1 short-circuits in the helpers: drive the collectives themselves once
buf = [torch.empty_like(rows)]
dist.all_gather(buf, rows)
assert torch.equal(buf[0], rows)
g = torch.ones(1 << 20, device=dev)
dist.all_reduce(g, op=dist.ReduceOp.SUM)
assert float(g.sum()) == float(1 << 20)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 1.5
assert torch.equal(parallel.gather_metric_rows(rows), rows) and parallel.max_over_ranks(0.5, dev) == 0.5
assert parallel.allreduce_sum_(g) == 1
# the training step's collective at its REAL size through RCCL: the flat fp32 gradient buffer of the dim-64 (1,2,4) denoiser
# (23.0 M parameters = 92 MB), filled by a libdpc kernel on the launch stream right before the all-reduce (stream ordering between
# the library's kernels and RCCL's), then the gradient norm on the reduced buffer -- /root/reference/diffusion/diffusion_2d_smoke.py:1025-1027
from diffphycon_amd import _lib as L
n = 23_012_352
flat = torch.empty(n, device=dev)
L.check(L.lib().dpc_philox_normal(L.ptr(flat), 1, n, 1234, 0, 0, L.stream()))
before = flat.clone()
assert parallel.allreduce_sum_(flat, force=True) == 1
norm = torch.zeros(1, device=dev)
ws = L.workspace(L.lib().dpc_reduce_workspace_bytes(), dev)
import ctypes as C
L.check(L.lib().dpc_l2_norm(L.ptr(flat), n, 1.0, L.ptr(norm), C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
assert torch.equal(flat, before)                              # SUM over one rank: the same bits back
assert abs(norm.item() - before.double().norm().item()) < 1e-3 * norm.item()
t0 = time.perf_counter()
for _ in range(5):
    parallel.allreduce_sum_(flat, force=True)
torch.cuda.synchronize()
print("allreduce 92 MB x5: %%.1f ms" %% ((time.perf_counter() - t0) * 1e3))
# the inference scripts' one exchange: ragged per-rank metric rows (float64 [B, 5]) through two all_gathers
mrows = torch.arange(64 * 5, dtype=torch.float64, device=dev).reshape(64, 5)
assert torch.equal(parallel.gather_metric_rows(mrows, force=True), mrows)
assert parallel.gather_metric_rows(mrows[:0], force=True).shape == (0, 5)
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_backend_initialises_and_runs_the_collectives_at_world_one():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
