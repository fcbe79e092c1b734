"""Repeat the Burgers two-rank-vs-one-rank entry-script comparison of tests/test_gpu_inference_scripts.py N times and print every
J_actual / Energy value (flake hunting: one full-suite run of r03 saw the two-rank result differ once).
  gpurun -- 'python tools/rank_stress.py [repeats] [graph 0/1]'
  gpurun -- 'python tools/rank_stress.py smoke [repeats]'    the smoke entry script (3-D denoisers) instead of the Burgers one
  gpurun -- 'python tools/rank_stress.py jelly [repeats]'    the jellyfish entry script (Unet3D denoisers + the 2-D surrogates' forward / backward)
  gpurun -- 'python tools/rank_stress.py solo [repeats]'     one rank ALONE on the GPU at batch 1 and batch 2 (the shard sizes of the
                                                             two-rank run): separates "small-batch path is not repeatable" from
                                                             "two processes time-sharing the GPU"
r03 (profiles/r03_bh_rank_stress.log, 70 repeats): 6 "mismatches" were the launcher's shared pipe interleaving the ranks' output
mid-line (fixed in run_ranks: per-rank files), 2 were real: J_actual 0.91172576 / 0.91172546 against 0.91172606 (3e-7 relative).
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_inference_scripts as T

solo = len(sys.argv) > 1 and sys.argv[1] == "solo"
smoke = len(sys.argv) > 1 and sys.argv[1] == "smoke"
jelly = len(sys.argv) > 1 and sys.argv[1] == "jelly"
if solo or smoke or jelly:
    sys.argv.pop(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if len(sys.argv) > 2:
    os.environ["DPC_BURGERS_GRAPH"] = sys.argv[2]
args = ["inference/inference_1d_burgers.py", "--dataset", "free_u_f_1e5_front_rear_quarter", "--partial_control",
        "front_rear_quarter", "--partially_observed", "front_rear_quarter", "--train_on_partially_observed", "None",
        "--set_unobserved_to_zero_during_sampling", "True", "--is_condition_u0", "True", "--is_condition_uT", "True",
        "--J_scheduler", "cosine", "--dim", "16", "--dim_muls", "1", "2", "4", "--exp_id", "POPC",
        "--dim__model_w", "16", "--dim_muls__model_w", "1", "2", "--exp_id__model_w", "POPC_w",
        "--is_model_w", "False", "--eval_two_models", "True", "--prior_beta", "0.9", "--w_scheduler", "sigmoid_flip",
        "--wus", "0.5", "--synthetic", "True", "--n_test_samples", "3", "--batch_size", "3", "--timesteps_override", "6"]
vals = lambda out: tuple(repr(T._floats_after(out, k)) for k in ("J_actual:", "Energy:"))
if jelly:
    # tests/test_gpu_inference_scripts.py::test_jellyfish_script_two_ranks_match_single_rank, repeated: saved thetas / states bit-equal
    import tempfile
    import numpy as np
    jargs = ["inference/inference_2d_jellyfish.py", "--synthetic", "True", "--batch_size", "3", "--num_batches", "1", "--frames", "4",
             "--image_size", "64", "--timesteps", "2"]
    with tempfile.TemporaryDirectory() as td:
        T.run(jargs + ["--inference_result_path", td + "/a"], ROOT)
        bad = 0
        for i in range(n):
            T.run_ranks(2, jargs + ["--inference_result_path", td + f"/b{i}"], ROOT)
            diffs = []
            for k in range(3):
                for sub in ("thetas", "states"):
                    x, y = np.load(f"{td}/a/{sub}/{k}.npy"), np.load(f"{td}/b{i}/{sub}/{k}.npy")
                    if not np.array_equal(x, y):
                        diffs.append((sub, k, float(np.abs(x - y).max())))
            bad += bool(diffs)
            print(i, "two ranks", "same" if not diffs else diffs, flush=True)
        print("mismatches", bad)
    sys.exit(0)
if smoke:
    # the smoke entry script (Unet3D denoisers: conv3w / conv3f3c, fused attention, panel / tile implicit GEMM, stem): two ranks on one
    # GPU against one rank, the test's arguments; every rank prints the gathered metrics
    import tempfile
    sargs = ["inference/inference_2d_smoke.py", "--synthetic", "True", "--n_test", "3", "--batch_size", "1", "--ddim_sampling_steps", "2"]
    keys = ("J_total:", "J_target:", "mse:", "n_l2:")
    sv = lambda out: tuple(repr(T._floats_after(out, k)[-1:]) for k in keys)
    with tempfile.TemporaryDirectory() as td:
        ref = sv(T.run(sargs + ["--inference_result_path", td + "/a"], ROOT))
        print("one rank ", ref, flush=True)
        bad = 0
        for i in range(n):
            out = T.run_ranks(2, sargs + ["--inference_result_path", td + f"/b{i}"], ROOT)
            allv = [T._floats_after(out, k) for k in keys]
            ok = all(repr([x]) == r for vs, r in zip(allv, ref) for x in vs)
            bad += not ok
            print(i, "two ranks", "same" if ok else allv, flush=True)
        print("mismatches", bad)
    sys.exit(0)
if solo:
    for bs in (1, 2):
        a = list(args)
        a[a.index("--n_test_samples") + 1] = str(bs)
        a[a.index("--batch_size") + 1] = str(bs)
        ref = vals(T.run(a, ROOT))
        diff = 0
        for i in range(n):
            v = vals(T.run(a, ROOT))
            if v != ref:
                diff += 1
                print("batch", bs, "repeat", i, v, "!=", ref, flush=True)
        print("batch", bs, ":", n, "repeats alone,", diff, "differ from the first run", ref, flush=True)
    sys.exit(0)
ref = vals(T.run(args, ROOT))
print("one rank ", ref, flush=True)
bad = 0
skip_one = os.environ.get("RANK_STRESS_SKIP_ONE") == "1"      # (hypothesis runs: only the two-rank leg, twice as many per GPU-minute)
for i in range(n):
    one = ref if skip_one else vals(T.run(args, ROOT))
    two = vals(T.run_ranks(2, args, ROOT))
    if one != ref or two[0].count(eval(ref[0])[-1].__repr__()) == 0:
        pass
    ok1 = one == ref
    ok2 = all(x == eval(ref[0])[-1] for x in eval(two[0])) and all(x == eval(ref[1])[-1] for x in eval(two[1]))
    bad += (not ok1) + (not ok2)
    print(i, "one-rank repeat", "same" if ok1 else one, "| two ranks", "same" if ok2 else two, flush=True)
print("mismatches", bad)
