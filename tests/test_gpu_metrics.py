"""The numbers a user actually reads -- J_total / J_target / J_energy / mse / n_l2 (smoke), J_actual / Energy / mse_deviation
(Burgers), and run_model's rescaled output -- against the REFERENCE's own functions run on the same inputs
(tools/gen_golden_r02.py: inference_2d_smoke.InferencePipeline.multi_evaluate / run_model :179-197, :317-427 with full
256-frame phi rollouts; utils.burgers_metric / mse_deviation :1188-1284).  Through the product's entry surface on the GPU."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def smoke_inf():
    sys.path.insert(0, os.path.join(ROOT, "inference"))
    import importlib
    mod = importlib.import_module("inference_2d_smoke")
    yield mod
    sys.path.remove(os.path.join(ROOT, "inference"))


def _seeded_pred(seed, B=2):                      # = tools/gen_golden_r02.py:seeded_pred (NumPy PCG64: host independent)
    rng = np.random.default_rng(seed)
    pred = rng.standard_normal((B, 32, 6, 64, 64)).astype(np.float32)
    pred[:, :, 3:5] *= 0.6
    pred[:, :, 5] = rng.uniform(0.1, 0.9, (B, 32, 1, 1)).astype(np.float32)
    return pred


def _seeded_data(seed, B=2):
    rng = np.random.default_rng(seed + 1000)
    data = np.zeros((B, 256, 6, 64, 64), np.float32)
    for b in range(B):
        r, c = rng.integers(10, 26), rng.integers(12, 53)
        data[b, 0, 0, r:r + 5, c:c + 5] = 1.0
    return data


def test_multi_evaluate_metric_rows_match_reference(smoke_inf, dev, tmp_path):
    """Rows A9 / D5: sampled controls -> 256-frame PDE rollouts -> (J_total, J_target, J_energy, mse, n_l2).  The rollout is
    bit-exact (test_gpu_smoke_solver.py), so the metrics agree to fp64 reduction-order noise."""
    g = load_golden("metrics_smoke")
    args = types.SimpleNamespace(image_size=64, device=dev, upsample=0, w_energy=float(g["w_energy"]), world_size=1)
    ppl = smoke_inf.InferencePipeline([None], {}, RESCALER=torch.ones(1, 1, 6, 1, 1, device=dev), results_path=str(tmp_path),
                                      args_general=args)
    pred = torch.from_numpy(_seeded_pred(int(g["seed"]))).to(dev)
    data = torch.from_numpy(_seeded_data(int(g["seed"])))
    out = ppl.multi_evaluate(pred, data, plot=False)
    for name, got in zip(("J_total", "J_target", "J_energy", "mse", "n_l2"), out):
        ref = g[name]
        assert got.shape == ref.shape == (1,)
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-9, err_msg=name)
    # per-trajectory rows kept for the sharded gather have the same means
    np.testing.assert_allclose(ppl.last_rows.mean(0).cpu().numpy()[0], g["J_total"][0], rtol=1e-6)


def test_run_model_tail_matches_reference(smoke_inf, dev, tmp_path):
    """Row A9: state[:, ::8] sub-sampling, init / init_u / control arguments handed to `sample`, output * RESCALER and the
    smoke-fraction channel replaced by its spatial mean."""
    g = load_golden("run_model")
    rng = np.random.default_rng(int(g["seed"]))
    state = torch.from_numpy(rng.standard_normal((2, 16, 6, 64, 64)).astype(np.float32))
    sample_out = torch.from_numpy(rng.standard_normal((2, 2, 6, 64, 64)).astype(np.float32)).to(dev)
    rec = {}

    class Stub:
        def sample(self, **kw):
            rec.update(kw)
            return sample_out.clone()
    R = torch.tensor([2, 18, 20, 16, 20, 1], dtype=torch.float32, device=dev).reshape(1, 1, 6, 1, 1)
    args = types.SimpleNamespace(image_size=64, device=dev, upsample=0, w_energy=0.0, world_size=1)
    ppl = smoke_inf.InferencePipeline([Stub()], {"design_fn": None, "design_guidance": "standard"}, RESCALER=R,
                                      results_path=str(tmp_path), args_general=args)
    out = ppl.run_model(state)
    assert rec["batch_size"] == int(g["batch_size"]) and rec["low"] is None
    for k in ("init", "init_u", "control"):
        assert torch.equal(rec[k].cpu(), torch.from_numpy(g[k])), k
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-6, atol=1e-6)


def test_burgers_metric_and_mse_deviation_match_reference(dev):
    """Row B8: burgers_metric (uncontrolled-half zeroing, re-simulation with the HIP finite-difference solver, MSE / median /
    MAE / normalised errors on the observed cells, control energy) and mse_deviation, every option the three scripts use."""
    from diffphycon_amd.utils_burgers import burgers_metric, mse_deviation
    g = load_golden("metrics_burgers")
    ut, f, ud = (torch.from_numpy(g[k]).to(dev) for k in ("u_target", "f", "u_diffused"))
    for tag, pc, po in (("full", "full", None), ("popc", "front_rear_quarter", "front_rear_quarter"),
                        ("fopc", "front_rear_quarter", None)):
        J, E = burgers_metric(ut, f, target="final_u", partial_control=pc, report_all=True, partially_observed=po)
        for name, v in zip(("mse", "mse_median", "mae", "mae_median", "nmse", "nmae"), J):
            # the re-simulated trajectory is bit-exact (B7); the metric reductions differ only by summation order
            np.testing.assert_allclose(v.cpu().numpy(), g[f"{tag}:J:{name}"], rtol=2e-5, atol=1e-9, err_msg=f"{tag} {name}")
        np.testing.assert_allclose(E.cpu().numpy(), g[f"{tag}:energy"], rtol=2e-6)
        Jd, _ = burgers_metric(ut, f, target="final_u", partial_control=pc, report_all=True, partially_observed=po,
                               diffused_u=ud, evaluate_u=True)
        np.testing.assert_allclose(Jd[0].cpu().numpy(), g[f"{tag}:Jdiff:mse"], rtol=2e-5)
        np.testing.assert_allclose(Jd[5].cpu().numpy(), g[f"{tag}:Jdiff:nmae"], rtol=2e-5)
        J1, _ = burgers_metric(ut, f, target="final_u", partial_control=pc, report_all=False, partially_observed=po)
        np.testing.assert_allclose(J1.cpu().numpy(), g[f"{tag}:J1"], rtol=2e-5, atol=1e-9)
    for tag, po in (("full", None), ("po", "front_rear_quarter")):
        np.testing.assert_allclose(mse_deviation(ud, ut, partially_observed=po).cpu().numpy(), g[f"dev:{tag}"], rtol=2e-5)
        for name, v in zip(("mse", "mae", "nmse", "nmae"), mse_deviation(ud, ut, partially_observed=po, report_all=True)):
            np.testing.assert_allclose(v.cpu().numpy(), g[f"dev:{tag}:{name}"], rtol=2e-5, err_msg=name)
