"""Round-2 fixtures made by importing the reference (build container only; the reference never travels):

    python tools/gen_golden_r02.py [metrics_smoke metrics_burgers run_model burgers_fopc datasets]

  metrics_smoke   : InferencePipeline.multi_evaluate (inference/inference_2d_smoke.py:317-427) on 2 seeded control sequences,
                    full 256-frame phi rollouts -> per-batch (J_total, J_target, J_energy, mse, n_l2)        [rows A9/D5]
  run_model       : InferencePipeline.run_model's tail (:179-197) around a recorded `sample` stub                   [row A9]
  metrics_burgers : utils.burgers_metric / mse_deviation (utils.py:1188-1284) for N = 4, every option the scripts use [row B8]
  burgers_fopc    : teacher-forced steps of the B-FOPC recipe (scripts/burgers_inference_full_obs_partial_ctr.sh: joint dim 64
                    (1,2,4) groups 1 + prior dim 32 (1,2,4,8), prior_beta 1.5, J cosine, w sigmoid_flip) at timesteps = 200
                    (BASELINE.json configs[0]) with SEEDED SYNTHETIC weights (oracle.unet2d.synthetic_state_dict; regenerated
                    from the seed by the tests, so the 9.9 M + 2.3 M parameters are not stored)
  datasets        : dataset/data_2d.py Jellyfish (test split) and Smoke (test split) readers on tiny files written from a seed

Every fixture is data only: seeds / small inputs and the reference's outputs.
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import refshim  # noqa: E402

refshim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gen_golden import save  # noqa: E402

torch.set_num_threads(8)


def seeded_pred(seed, B=2):
    """Sampler-output-like tensor [B, 32, 6, 64, 64] (already multiplied by RESCALER) from a NumPy PCG64 stream: identical on
    every host, so the fixture stores only the seed."""
    rng = np.random.default_rng(seed)
    pred = rng.standard_normal((B, 32, 6, 64, 64)).astype(np.float32)
    pred[:, :, 3:5] *= 0.6                       # controls of the magnitude the rollouts in phi_rollout*.npz use
    pred[:, :, 5] = rng.uniform(0.1, 0.9, (B, 32, 1, 1)).astype(np.float32)      # smoke fraction: spatially constant
    return pred


def seeded_data(seed, B=2):
    """Test-split sample [B, 256, 6, 64, 64]: only the initial density (frame 0, channel 0) is read by the metrics path."""
    rng = np.random.default_rng(seed + 1000)
    data = np.zeros((B, 256, 6, 64, 64), np.float32)
    for b in range(B):
        r, c = rng.integers(10, 26), rng.integers(12, 53)
        data[b, 0, 0, r:r + 5, c:c + 5] = 1.0
    return data


def gen_metrics_smoke():
    import inference.inference_2d_smoke as inf

    class A:
        image_size, device, upsample, w_energy = 64, "cpu", 0, 0.25
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)                                 # multi_evaluate drops plot_pred.npy into the cwd
    try:
        ppl = inf.InferencePipeline([None], {}, RESCALER=torch.ones(1, 1, 6, 1, 1), results_path=tmp, args_general=A)
        pred = torch.from_numpy(seeded_pred(7))
        data = torch.from_numpy(seeded_data(7))
        out = ppl.multi_evaluate(pred.clone(), data, plot=False)
    finally:
        os.chdir(cwd)
    save("metrics_smoke", seed=7, w_energy=A.w_energy, J_total=out[0], J_target=out[1], J_energy=out[2], mse=out[3],
         n_l2=out[4])


def gen_run_model():
    import inference.inference_2d_smoke as inf

    class A:
        image_size, device, upsample, w_energy = 64, "cpu", 0, 0.0
    R = torch.tensor([2, 18, 20, 16, 20, 1], dtype=torch.float32).reshape(1, 1, 6, 1, 1)      # data_2d.py:167 as load_data shapes it
    rng = np.random.default_rng(11)
    state = torch.from_numpy(rng.standard_normal((2, 16, 6, 64, 64)).astype(np.float32))
    sample_out = torch.from_numpy(rng.standard_normal((2, 2, 6, 64, 64)).astype(np.float32))
    rec = {}

    class Stub:
        def sample(self, **kw):
            rec.update({k: (v.clone() if torch.is_tensor(v) else v) for k, v in kw.items()})
            return sample_out.clone()
    tmp = tempfile.mkdtemp()
    ppl = inf.InferencePipeline([Stub()], {"design_fn": None, "design_guidance": "standard"}, RESCALER=R, results_path=tmp,
                                args_general=A)
    out = ppl.run_model(state)
    save("run_model", seed=11, out=out, batch_size=rec["batch_size"], init=rec["init"], init_u=rec["init_u"],
         control=rec["control"])


def gen_metrics_burgers():
    from utils import burgers_metric, mse_deviation
    rng = np.random.default_rng(5)
    N = 4
    xg = np.linspace(0, 1, 128)
    u_target = np.zeros((N, 11, 128), np.float32)
    for k in range(N):
        u0 = rng.uniform(0, 2) * np.exp(-0.5 * ((xg - rng.uniform(0.2, 0.4)) / rng.uniform(0.05, 0.15)) ** 2) \
            - rng.uniform(0, 2) * np.exp(-0.5 * ((xg - rng.uniform(0.6, 0.8)) / rng.uniform(0.05, 0.15)) ** 2)
        for t in range(11):
            u_target[k, t] = np.roll(u0, 2 * t) * (1 - 0.03 * t)
    f = (rng.standard_normal((N, 10, 128)) * 0.5).astype(np.float32)
    u_diff = (u_target + 0.05 * rng.standard_normal((N, 11, 128))).astype(np.float32)
    ut, ft, ud = torch.from_numpy(u_target), torch.from_numpy(f), torch.from_numpy(u_diff)
    arrays = dict(u_target=u_target, f=f, u_diffused=u_diff)
    with torch.no_grad():
        for tag, pc, po in (("full", "full", None), ("popc", "front_rear_quarter", "front_rear_quarter"),
                            ("fopc", "front_rear_quarter", None)):
            J, E = burgers_metric(ut, ft, target="final_u", partial_control=pc, report_all=True, partially_observed=po)
            for name, v in zip(("mse", "mse_median", "mae", "mae_median", "nmse", "nmae"), J):
                arrays[f"{tag}:J:{name}"] = v
            arrays[f"{tag}:energy"] = E
            Jd, _ = burgers_metric(ut, ft, target="final_u", partial_control=pc, report_all=True, partially_observed=po,
                                   diffused_u=ud, evaluate_u=True)
            arrays[f"{tag}:Jdiff:mse"] = Jd[0]
            arrays[f"{tag}:Jdiff:nmae"] = Jd[5]
            J1, _ = burgers_metric(ut, ft, target="final_u", partial_control=pc, report_all=False, partially_observed=po)
            arrays[f"{tag}:J1"] = J1
        for tag, po in (("full", None), ("po", "front_rear_quarter")):
            arrays[f"dev:{tag}"] = mse_deviation(ud, ut, partially_observed=po)
            for name, v in zip(("mse", "mae", "nmse", "nmae"), mse_deviation(ud, ut, partially_observed=po, report_all=True)):
                arrays[f"dev:{tag}:{name}"] = v
    save("metrics_burgers", **arrays)


def gen_burgers_fopc():
    from model.burgers_1d.unet import Unet2D
    from diffusion.diffusion_1d_burgers import (GaussianDiffusion, get_nablaJ, cosine_beta_J_schedule, sigmoid_schedule_flip)
    from utils import ddpm_guidance_loss, mse_dist_reg
    from oracle import unet2d as U

    c_uw = U.Unet2DConfig(dim=64, dim_mults=(1, 2, 4), resnet_block_groups=1)
    c_w = U.Unet2DConfig(dim=32, dim_mults=(1, 2, 4, 8), resnet_block_groups=1)
    kw = dict(init_dim=None, out_dim=2, channels=2, resnet_block_groups=1)
    m_uw = Unet2D(dim=64, dim_mults=(1, 2, 4), **kw).eval()
    m_w = Unet2D(dim=32, dim_mults=(1, 2, 4, 8), **kw).eval()
    m_uw.load_state_dict(U.synthetic_state_dict(c_uw, seed=41))
    m_w.load_state_dict(U.synthetic_state_dict(c_w, seed=42))
    B, T = 2, 200
    rng = np.random.default_rng(9)
    u_target = torch.from_numpy(rng.standard_normal((B, 11, 128)).astype(np.float32))
    u0, uT = u_target[:, 0] / 10, u_target[:, 10] / 10
    w = (0.7, 0.01, 0.05, None)                    # FOPC: fully observed; non-zero weights so the guidance terms are exercised

    def loss(x):
        return ddpm_guidance_loss(u_target / 10, x[:, 0, :11, :], x[:, 1, :10, :], wu=w[0], wf=w[1], wreg=w[2],
                                  dist_reg=mse_dist_reg, partially_observed=w[3])
    gd = GaussianDiffusion((m_uw, m_w), seq_length=(16, 128), timesteps=T, auto_normalize=False, use_conv2d=True,
                           temporal=True, is_condition_u0=True, is_condition_uT=True,
                           set_unobserved_to_zero_during_sampling=False, eval_two_models=True, prior_beta=1.5,
                           normalize_beta=False)
    arrays = dict(u_target=u_target, seed_uw=41, seed_w=42, timesteps=T, weights=np.array(w[:3]))
    kwargs = dict(clip_denoised=True, nablaJ=get_nablaJ(loss), J_scheduler=cosine_beta_J_schedule,
                  w_scheduler=sigmoid_schedule_flip, guidance_u0=True, u_init=u0, u_final=uT)
    g = torch.Generator().manual_seed(77)
    for t in (199, 120, 37, 0):
        x_in = torch.randn(B, 2, 16, 128, generator=g) * (0.3 + 0.7 * t / 199)
        # condition exactly as p_sample_loop does before the step (:539-553)
        x_in[:, 0, 0, :] = u0
        x_in[:, 0, 10, :] = uT
        torch.manual_seed(1000 + t)
        out = gd.p_sample(x_in.clone(), t, **kwargs)
        torch.manual_seed(1000 + t)
        z = torch.randn(B, 2, 16, 128) if t > 0 else torch.zeros(B, 2, 16, 128)
        arrays[f"t{t}:x_in"] = x_in
        arrays[f"t{t}:z"] = z
        arrays[f"t{t}:x_out"] = out[0].detach()
        arrays[f"t{t}:x0"] = out[1].detach()
        arrays[f"t{t}:pred_noise"] = out[2].detach()
        with torch.no_grad():
            tb = torch.full((B,), t, dtype=torch.long)
            arrays[f"t{t}:eps_uw"] = m_uw(x_in, tb)
            xw = x_in.clone()
            xw[..., 0, 1:10, :] = 0
            arrays[f"t{t}:eps_w"] = m_w(xw, tb)
    save("burgers_fopc", **arrays)


sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
from dataset_files import write_jellyfish_files, write_smoke_files  # noqa: E402


def gen_datasets():
    from dataset.data_2d import Jellyfish, Smoke
    arrays = dict(seed_jelly=3, seed_smoke=4)
    root = tempfile.mkdtemp()
    write_jellyfish_files(root, 3)
    for tag, ovp in (("full", False), ("pob", True)):
        ds = Jellyfish(dataset="jellyfish", dataset_path=root, time_steps=40, steps=20, time_interval=1, is_train=False,
                       is_testdata=True, only_vis_pressure=ovp)
        arrays[f"jelly:{tag}:len"] = len(ds)
        for i in (0, 2):
            state_0, thetas_0, bd_0, sim_id, thetas_gt = ds[i]
            arrays[f"jelly:{tag}:{i}:state_0"] = state_0
            arrays[f"jelly:{tag}:{i}:thetas_0"] = thetas_0
            arrays[f"jelly:{tag}:{i}:bd_0"] = bd_0
            arrays[f"jelly:{tag}:{i}:sim_id"] = sim_id
            arrays[f"jelly:{tag}:{i}:thetas_gt"] = thetas_gt
    root2 = tempfile.mkdtemp()
    write_smoke_files(root2, 4)
    ds = Smoke(dataset_path=root2, is_train=False)
    arrays["smoke:len"] = len(ds)
    state, sim_id = ds[1]
    arrays["smoke:1:state_frames"] = state[[0, 1, 8, 255]]            # 4 of the 256 frames keep the fixture small
    arrays["smoke:1:state_mean"] = state.double().mean((2, 3))
    arrays["smoke:1:sim_id"] = sim_id
    arrays["smoke:RESCALER"] = ds.RESCALER
    save("datasets", **arrays)


SECTIONS = {"metrics_smoke": gen_metrics_smoke, "run_model": gen_run_model, "metrics_burgers": gen_metrics_burgers,
            "burgers_fopc": gen_burgers_fopc, "datasets": gen_datasets}

if __name__ == "__main__":
    for s in (sys.argv[1:] or list(SECTIONS)):
        print("==", s)
        SECTIONS[s]()
