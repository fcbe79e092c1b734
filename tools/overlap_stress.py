"""Is the overlapped evaluator schedule of inference/inference_2d_smoke.py bit-safe at PRODUCTION shapes (64 trajectories per batch: rollouts on
64 CUs beside full-size sampling kernels on the other 192)?  N repetitions of a 3-batch run (DDIM, 12 steps) with --overlap_evaluator True
against ONE serial run: every per-trajectory metric row (J_total, J_target, J_energy, mse, n_l2) must be bit-equal.
    gpurun -- 'python tools/overlap_stress.py [repetitions] > gpurun_out/overlap_stress.log'"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "inference"))
import inference_2d_smoke as S  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def run(overlap, tag):
    a = S.build_parser().parse_args(["--synthetic", "True", "--n_test", "192", "--batch_size", "64", "--ddim_sampling_steps", "12",
                                     "--overlap_evaluator", str(overlap), "--inference_result_path", f"/tmp/overlap_stress_{tag}"])
    a.device, a.rank, a.world_size = torch.device("cuda:0"), 0, 1
    a.inference_result_subpath = f"/tmp/overlap_stress_{tag}/r"
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        loader, rescaler = S.load_data(a)
        diffusion, design_fn = S.load_model(a, rescaler, a.w_energy, w_init=a.w_init)
        ppl = S.InferencePipeline(diffusion, {"design_fn": design_fn, "design_guidance": a.design_guidance}, rescaler,
                                  results_path=a.inference_result_subpath, args_general=a)
        ppl.run(loader)
    return torch.cat(ppl.all_rows).cpu().numpy()


ref = run(False, "serial")
assert ref.shape == (192, 5) and np.isfinite(ref).all()
bad = 0
for r in range(reps):
    rows = run(True, f"ov{r}")
    diff = int((rows != ref).any(axis=1).sum())
    bad += diff
    print(f"repetition {r}: {diff} of 192 trajectories differ from the serial schedule", flush=True)
print(f"overlapped vs serial, {reps} x 192 trajectories: {bad} differ")
