"""`python bench.py --gpus N` must launch its own ranks when no launcher did (the driver's invocation): CPU test of the
launcher / barrier / max-over-ranks plumbing with the gloo backend and a stubbed step (no GPU work, `--stub`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "3", "--warmup", "1"] + extra,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 prints exactly ONE JSON line
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    out = _run(["--gpus", "2"])
    assert out["stub"] is True and out["n_gpus"] == 2 and out["world_size_seen_by_backend"] == 2
    assert out["launcher"].startswith("self")
    assert out["config"]["global_batch"] == 128
    # rank r sleeps 2 (1 + r) ms per step: the reported step time is the MAX over ranks, the min is reported beside it
    assert out["ms_per_step"] >= 3.9 and out["ms_per_step_min_rank"] < out["ms_per_step"]
    assert abs(out["value"] - 128 / (1000 * out["ms_per_step"] * 1e-3)) < 1e-9


def test_single_rank_needs_no_launcher():
    out = _run(["--gpus", "1"])
    assert out["n_gpus"] == 1 and out["launcher"] == "single process"


def test_world_size_mismatch_is_an_error():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["WORLD_SIZE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stdout + p.stderr)


def test_self_launch_eight_ranks_as_the_driver_does():
    """The node-level configuration the driver measures (N = 8, one rank per GPU): 8 gloo ranks, global batch 8 x 64 = 512, one
    JSON line from rank 0, max / min over the ranks (rank r sleeps 2 (1 + r) ms: the slowest, rank 7, sets the step time; the
    closing barrier is inside the timed region, so every rank's elapsed time is that of the slowest)."""
    out = _run(["--gpus", "8"])
    assert out["n_gpus"] == 8 and out["world_size_seen_by_backend"] == 8
    assert out["config"]["global_batch"] == 512
    assert out["ms_per_step"] >= 15.9 and out["ms_per_step_min_rank"] <= out["ms_per_step"]
    assert abs(out["value"] - 512 / (1000 * out["ms_per_step"] * 1e-3)) < 1e-9
