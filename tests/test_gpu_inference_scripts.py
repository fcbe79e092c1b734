"""End-to-end smoke runs of the reference's entry surface (inference/*.py) on the HIP path with --synthetic inputs and
shortened chains: the scripts parse the reference's flags, sample, evaluate with the PDE solvers and report metrics."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd):
    p = subprocess.run([sys.executable] + cmd, cwd=cwd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_burgers_inference_script_popc_flags(tmp_path):
    out = run(["inference/inference_1d_burgers.py", "--dataset", "free_u_f_1e5_front_rear_quarter", "--partial_control",
               "front_rear_quarter", "--partially_observed", "front_rear_quarter", "--train_on_partially_observed", "None",
               "--set_unobserved_to_zero_during_sampling", "True", "--is_condition_u0", "True", "--is_condition_uT", "True",
               "--J_scheduler", "cosine", "--dim", "16", "--dim_muls", "1", "2", "4", "--exp_id", "POPC",
               "--dim__model_w", "16", "--dim_muls__model_w", "1", "2", "--exp_id__model_w", "POPC_w",
               "--is_model_w", "False", "--eval_two_models", "True", "--prior_beta", "0.9", "--w_scheduler", "sigmoid_flip",
               "--synthetic", "True", "--n_test_samples", "4", "--batch_size", "4", "--timesteps_override", "8"], ROOT)
    assert "J_actual:" in out and "Energy:" in out


def test_smoke_inference_script_ddim(tmp_path):
    out = run(["inference/inference_2d_smoke.py", "--synthetic", "True", "--n_test", "2", "--batch_size", "2",
               "--ddim_sampling_steps", "2", "--inference_result_path", str(tmp_path)], ROOT)
    assert "Final results!" in out and "J_total" in out
    assert any(f == "results.txt" for _, _, fs in os.walk(tmp_path) for f in fs)


def test_jellyfish_inference_script(tmp_path):
    out = run(["inference/inference_2d_jellyfish.py", "--synthetic", "True", "--batch_size", "1", "--num_batches", "1",
               "--frames", "4", "--image_size", "64", "--timesteps", "3", "--inference_result_path", str(tmp_path)], ROOT)
    assert "Final results!" in out
    assert os.path.exists(os.path.join(str(tmp_path), "thetas", "0.npy"))
