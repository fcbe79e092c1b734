#!/usr/bin/env python
"""Headline benchmark: guided trajectories/sec, 2-D smoke 64x64x32 @ 1000 DDPM steps (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (RANK /
LOCAL_RANK / WORLD_SIZE in the environment), or as a plain `python bench.py --gpus N`, which re-executes itself under
torch.distributed.run with one rank per GPU (RCCL, rendezvous on 127.0.0.1).

A "step" is ONE guided DDPM step of the whole local batch (config S64: 64 trajectories per GPU): joint U-Net forward + prior
U-Net forward + fused guidance/posterior update, i.e. 1/1000 of the sampling of each trajectory.
trajectories/s = (N * 64) / (1000 * seconds_per_step), seconds_per_step = MAX over ranks of (elapsed / K) between two
barrier + synchronize pairs.  Inputs and state are resident in HBM before the timed region; weights are seeded random
initialisations of the reference architecture (no checkpoints exist offline) and data is synthetic.

The JSON line also carries
  roofline     : achieved TFLOP/s of the dominant kernel class from HIP events recorded on the launch stream inside the timed
                 region, against the MFMA peak of the arithmetic mode the library reports for that op family;
  cpu_baseline : the CPU oracle (torch fp32, all usable host cores) executing the same step: 1 warm-up + up to 5 timed steps
                 at B=1 and at B=min(8, cores) inside a time budget, N=1 only;
  value_exact  : the same timed loop re-run with exact fp32 products (arithmetic mode x6: 3-way bf16 split) -- the default
                 mode (f16x3) carries 22 significant operand bits;
  burgers      : BASELINE.json configs[1] (Burgers POPC, B=256/GPU) measured in the same run, with its own roofline/cpu_baseline;
  smoke_evaluator : rollouts/s of the post-sampling PDE evaluator (N=1 only);
  roofline_step / roofline_exact : the whole step's algorithmic TFLOP/s against the same roof; the dominant class of the x6 leg;
  ddim100 / e2e   : ONE real pass each of the entry script's pipeline (sample() + PDE evaluator + metric rows) at B=64 per GPU: the
                    CLI default DDIM-100 and the full 1000-step DDPM; wall seconds, nothing extrapolated; e2e checks that its
                    measured ms per step agrees with the timed loop's within 2 % (--no-e2e skips both: ~5 minutes);
  every leg carries its per-rank min/max step time and the world size RCCL reported.
"""
import argparse
import json
import contextlib
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3         # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # same guide: v_mfma_f32_32x32x16_bf16 / _f16 dense peak
STEPS_PER_TRAJECTORY = 1000
LOCAL_BATCH = 64
FRAMES, SIZE = 32, 64
UNIT_GFLOP = 1794.5                   # algorithmic GFLOP per S64 trajectory-step (both U-Nets)
BURGERS_UNIT_GFLOP = 15.8             # Burgers POPC pair per trajectory-step


def family_of(kernel_class):
    if kernel_class.startswith("conv3"):
        return "conv"
    if kernel_class.startswith("igemm"):
        return "igemm"
    if kernel_class.startswith("stem"):
        return "stem"
    if "attention" in kernel_class:
        return "attn"
    return None


def mfma_roof(kernel_class, modes):
    """Roof for the ALGORITHMIC fp32 flops of a kernel class under the arithmetic mode the model handle reports
    (`modes` = 'conv=..,igemm=..,attn=..,stem=..' from dpc_unet*_modes): the split-operand kernels spend 3 (f16x3) or 6 (x6)
    16-bit MFMAs per fp32 product."""
    md = dict(kv.split("=") for kv in modes.split(",")) if modes else {}
    mode = md.get(family_of(kernel_class), "f32")
    if kernel_class == "conv3_wgrad_f16x3":           # the 3x3x3 weight gradient is always f16x3 (wgrad3.hip), the other geometries
        mode = "f16x3"                                # ("conv_wgrad") exact fp32 products on the fp32 MFMA (train.hip)
    elif kernel_class == "conv_wgrad":
        mode = "f32"
    if mode == "f32":
        return PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense"
    if mode == "x6":
        return PEAK_BF16_MFMA_TFLOPS / 6.0, "2500 TF bf16 dense / 6 MFMAs per fp32 product (bf16x6 split)"
    return PEAK_BF16_MFMA_TFLOPS / 3.0, "2500 TF fp16 dense / 3 MFMAs per fp32 product (f16x3 split)"


def dtype_label(modes):
    """What the `dtype` field says: tensors in HBM and every accumulation are fp32; the PRODUCTS of the GEMM-shaped kernels are formed
    from split 16-bit operands in the arithmetic mode the library reports (f16x3: 22 significant operand bits, 3 MFMAs per product;
    x6: exact 3-way bf16 split, 6 MFMAs; f32: the native fp32 MFMA)."""
    md = sorted(set(kv.split("=")[1] for kv in modes.split(","))) if modes else ["f32"]
    if md == ["f32"]:
        return "f32"
    return "f32 storage/accumulate; " + "/".join(md) + " products"


def step_roofline(modes, flop_per_step, sec_per_step, note):
    """Whole-step figure next to the dominant kernel's: ALGORITHMIC flop of one step / measured step time against the MFMA roof of the
    arithmetic mode of the convolution family (where > 90 % of the flop are)."""
    peak, peak_note = mfma_roof("conv3x6_bn64", modes)
    achieved = flop_per_step / sec_per_step / 1e12
    return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
            "peak_note": peak_note, "note": note}


def mfma_issue(kernel_class, modes, achieved_tflops):
    """What the matrix pipe actually ISSUES for the algorithmic rate `achieved`: products per fp32 product of the
    class's arithmetic mode (f16x3: 3, x6: 6 on the 16-bit pipe; f32: 1 on the fp32 pipe) x the fraction of the direct form's products
    the convolution algorithm keeps (3x3x3 convolutions: Winograd over frames F(4,3) 1/2, F(2,3) 2/3; everything else 1), against
    the DENSE peak of that pipe -- `frac` prices algorithmic flop against peak / products, this one prices issued MFMA flop."""
    fam = family_of(kernel_class)
    mode = dict(kv.split("=") for kv in modes.split(",")).get(fam, "f32") if modes else "f32"
    per, peak = {"f16x3": (3.0, PEAK_BF16_MFMA_TFLOPS), "x6": (6.0, PEAK_BF16_MFMA_TFLOPS)}.get(mode, (1.0, PEAK_FP32_MFMA_TFLOPS))
    keep, alg = 1.0, None
    if fam == "conv" and mode == "f16x3":
        from diffphycon_amd import _lib
        alg = _lib.lib().dpc_conv3d_algorithm().decode()
        keep = {"winograd_f43_frames": 0.5, "winograd_f23_frames": 2.0 / 3.0}.get(alg, 1.0)
    return {"mfma_issue_frac": achieved_tflops * per * keep / peak,
            "mfma_issue_note": f"{per:g} MFMA products per fp32 product ({mode})" + (f" x {keep:.3f} of the direct form's products ({alg})" if alg else "")
                               + f" / {peak:g} TFLOP/s dense"}


def pmc_sq(kernel_class, workload):
    """Matrix-pipe busy fraction and sustained shader clock of the class from the COMMITTED rocprofv3 --pmc SQ pass (profiles/pmc_sq.json,
    tools/gpu_evidence.sh; stamped with the kernel sources like pmc_traffic.json) -- null when absent or stale.  Not measured by this run."""
    out = {"mfma_busy_frac": None, "shader_clock_ghz": None}
    if workload != "smoke":
        return out
    try:
        row = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json"))).get(kernel_class, {})
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from pmc_summary import source_stamp
        if row.get("kernel_source_sha16") is not None and row.get("kernel_source_sha16") == source_stamp(kernel_class):
            out = {"mfma_busy_frac": row.get("mfma_busy_frac"), "shader_clock_ghz": row.get("shader_clock_ghz"),
                   "pmc_sq_source": "committed rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (profiles/pmc_sq.json), not re-measured by this run"}
    except (OSError, ValueError):
        pass
    return out


def pmc_traffic(kernel_class, workload="smoke"):
    """HBM bytes per launch of a kernel class from the COMMITTED rocprofv3 --pmc passes (profiles/pmc_traffic.json for the S64 headline
    loop, profiles/pmc_traffic_<workload>.json for the burgers / train legs -- tools/pmc_legs.sh: FETCH_SIZE doubled per the guide's
    gfx950 correction, + WRITE_SIZE), or None.  Not measured by this run: PMC needs rocprofv3."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json" if workload == "smoke" else f"pmc_traffic_{workload}.json")
    try:
        row = json.load(open(path)).get(kernel_class, {})
    except (OSError, ValueError):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        from pmc_summary import source_stamp
        stamp = source_stamp(kernel_class)
    except Exception:
        stamp = None
    # counters measured on another version of the kernel's source are stale: report null rather than a wrong number
    if row.get("kernel_source_sha16") is None or row.get("kernel_source_sha16") != stamp:
        return None
    return row.get("hbm_bytes_per_launch")


def synthetic_init(batch, traj0, size=SIZE):
    """zeros with a 5x5 block of ones at rows 10..25 / cols 12..52, init = density / 2."""
    init = torch.zeros(batch, size, size)
    g = torch.Generator().manual_seed(0)
    pos = torch.stack((torch.randint(10, 26, (4096,), generator=g), torch.randint(12, 53, (4096,), generator=g)), 1)
    for b in range(batch):
        r, c = pos[(traj0 + b) % 4096].tolist()
        init[b, r:r + 5, c:c + 5] = 0.5
    return init


def build_models(device, micro_batch, arithmetic=None, frames=FRAMES, size=SIZE):
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion
    torch.manual_seed(0)
    mj = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6, micro_batch=micro_batch, arithmetic=arithmetic)
    mw = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=2, micro_batch=micro_batch, arithmetic=arithmetic)
    sd_cpu = (mj.state_dict(), mw.state_dict())
    sd_cpu = tuple({k: v.clone() for k, v in sd.items()} for sd in sd_cpu)
    gd = GaussianDiffusion([mj.to(device), mw.to(device)], image_size=size, frames=frames, timesteps=1000,
                           sampling_timesteps=1000, loss_type="l2", objective="pred_noise", standard_fixed_ratio=1e5,
                           coeff_ratio=0.0, eval_2ddpm=True, w_prob_exp=0.97, device=device)
    return gd, sd_cpu


def usable_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box shows
    256 logical CPUs but a 16-CPU quota; oversubscribing oneDNN by 16x stalls for tens of minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _timed_cpu_steps(step, budget_s, max_steps=5):
    """1 untimed warm-up (oneDNN primitive creation) + up to `max_steps` timed steps while the leg stays inside its budget."""
    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    times = []
    while len(times) < max_steps and (not times or (time.perf_counter() - t0) + times[-1] < budget_s):
        t1 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t1)
    return warm, times


def cpu_baseline(sd_cpu, budget_s):
    """the CPU oracle (the reference's algorithm restated in torch fp32) on the same unit, 1 warm-up +
    up to 5 timed steps at B=1 and at B=min(8, cores) -- the plan of record since r06 (--cpu-budget defaults to 420 s of host time
    for all CPU legs: B = 8 costs ~45 s per step on the 16-thread host); `value` is the better of the two rates."""
    from oracle import unet3d as O
    from oracle import sampler_smoke as S
    cores = usable_cores()
    torch.set_num_threads(cores)
    cj = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6)
    cw = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=2)
    sched = S.make_schedule(1000, "sigmoid")
    legs = []
    for B, share in ((1, 0.12), (min(8, cores), 0.88)):
        if legs and B == legs[0]["B"]:
            break
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, FRAMES, 6, SIZE, SIZE, generator=g)
        z = torch.randn(B, FRAMES, 6, SIZE, SIZE, generator=g)
        init = synthetic_init(B, 0)
        x[:, 0, 0] = init
        t = torch.full((B,), 999, dtype=torch.long)

        def step():
            with torch.no_grad():
                ej = O.unet3d_forward(sd_cpu[0], cj, x, t)
                ew = O.unet3d_forward(sd_cpu[1], cw, x[:, :, 3:5], t)
                S.p_sample_step(sched, x.clone(), 999, ej, ew, z, init, S.rescaler_tensor())
        warm, times = _timed_cpu_steps(step, budget_s * share)
        mean = sum(times) / len(times)
        legs.append({"B": B, "warmup_s": round(warm, 3), "timed_steps": len(times), "mean_s_per_step": round(mean, 4),
                     "trajectories_per_s": B / (STEPS_PER_TRAJECTORY * mean)})
    best = max(legs, key=lambda l: l["trajectories_per_s"])
    return {"value": best["trajectories_per_s"], "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"guided DDPM steps (joint+prior U-Net forward + update), 64x64x32, torch fp32 on {cores} threads: "
                      + "; ".join(f"B={l['B']}: 1 warm-up + {l['timed_steps']} timed, {l['mean_s_per_step']} s/step" for l in legs)
                      + "; extrapolated x1000 steps.  Plan of record (BASELINE.md 3): 1 warm-up + 5 timed steps at B=1 and B=min(8, cores) "
                        f"(~5 minutes of host time); this run used a {budget_s:.0f} s budget (--cpu-budget): the step counts above",
            "legs": legs}


# ------------------------------------------------------------------------------------------------ Burgers POPC
def burgers_setup(device, batch, rank, arithmetic=None):
    """BASELINE.json configs[1]: Burgers POPC recipe of scripts/burgers_inference_partial_obs_partial_ctr.sh."""
    from diffphycon_amd.model.burgers_1d.unet import Unet2D
    from diffphycon_amd.diffusion import diffusion_1d_burgers as D
    torch.manual_seed(0)
    kw = dict(dim=64, out_dim=2, channels=2, resnet_block_groups=1, arithmetic=arithmetic)
    m_uw = Unet2D(dim_mults=(1, 2, 4, 8, 16), **kw)
    m_w = Unet2D(dim_mults=(1, 2, 4, 8), **kw)
    sd_cpu = tuple({k: v.clone() for k, v in m.state_dict().items()} for m in (m_uw, m_w))
    gd = D.GaussianDiffusion((m_uw.to(device), m_w.to(device)), seq_length=(16, 128), timesteps=1000, auto_normalize=False,
                             use_conv2d=True, temporal=True, is_condition_u0=True, is_condition_uT=True,
                             set_unobserved_to_zero_during_sampling=True, eval_two_models=True, prior_beta=0.9,
                             normalize_beta=False).to(device)
    # u0 = two Gaussians (generate_burgers.py:361-372), uT = u0 rolled by 16 cells, both / 10
    g = torch.Generator().manual_seed(1000 + rank)
    xg = torch.linspace(0, 1, 128)[None, :]

    def bump(lo, hi, alo, ahi):
        loc = lo + (hi - lo) * torch.rand(batch, 1, generator=g)
        amp = alo + (ahi - alo) * torch.rand(batch, 1, generator=g)
        sig = 0.05 + 0.10 * torch.rand(batch, 1, generator=g)
        return amp * torch.exp(-0.5 * ((xg - loc) / sig) ** 2)
    u0 = bump(0.2, 0.4, 0.0, 2.0) + bump(0.6, 0.8, -2.0, 0.0)
    uT = torch.roll(u0, 16, dims=1)
    ut = torch.zeros(batch, 11, 128)
    ut[:, 0], ut[:, 10] = u0, uT
    guide = D.BurgersGuidance(ut / 10, 0.0, 0.0, 0.0, "front_rear_quarter")       # shipped scripts: all-zero weights
    kwargs = dict(nablaJ=guide, J_scheduler=D.cosine_beta_J_schedule, w_scheduler=D.sigmoid_schedule_flip,
                  u_init=(u0 / 10).to(device), u_final=(uT / 10).to(device), clip_denoised=True)
    return gd, kwargs, sd_cpu, (u0 / 10, uT / 10)


def burgers_cpu_baseline(sd_cpu, cond, budget_s):
    """The POPC recipe on the CPU oracle, once under no_grad and once in the REFERENCE's mode: its sampling loop runs with autograd
    enabled (diffusion_1d_burgers.py:525 "removed no_grad decorator here"; the denoisers' parameters require grad, so both forwards
    record a graph, and get_nablaJ :34-49 differentiates the guidance loss with create_graph=True).  BASELINE.md section 3 plans
    B = 50 with 20 timed steps per mode; this runs B = 50 (the entry script's batch, inference_1d_burgers.py:341) with 1 warm-up + as
    many timed steps (<= 20) per mode as the bounded sample allows, and says how many.  `value` = the better of the two rates."""
    from oracle import unet2d as U
    from oracle import sampler_burgers as S
    cores = usable_cores()
    torch.set_num_threads(cores)
    B = min(50, int(cond[0].shape[0]))
    c_uw = U.Unet2DConfig(dim=64, dim_mults=(1, 2, 4, 8, 16), resnet_block_groups=1)
    c_w = U.Unet2DConfig(dim=64, dim_mults=(1, 2, 4, 8), resnet_block_groups=1)
    sched = S.make_schedule(1000, "cosine")
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(B, 2, 16, 128, generator=g)
    z = torch.randn(B, 2, 16, 128, generator=g)
    tb = torch.full((B,), 999, dtype=torch.long)
    ut = torch.zeros(B, 11, 128)
    ut[:, 0], ut[:, 10] = cond[0][:B], cond[1][:B]
    kw = dict(prior_beta=0.9, eta_w=S.scheduler_table("sigmoid_flip")[999], eta_J=S.scheduler_table("cosine")[999])

    def step_nograd():
        x = x0.clone()
        with torch.no_grad():
            S.set_conditions(x, cond[0][:B], cond[1][:B], True)
            e_uw = U.unet2d_forward(sd_cpu[0], c_uw, x, tb)
            e_w = U.unet2d_forward(sd_cpu[1], c_w, S.w_model_input(x), tb)
            S.p_sample_step(sched, x, 999, e_uw, e_w, z, **kw)

    sd_grad = tuple({k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()} for sd in sd_cpu)

    def step_grad():
        # what the reference's loop pays: both forwards under autograd (graphs of a 136 M-parameter and a 36 M-parameter U-Net
        # recorded and dropped every step), the guidance gradient by autograd.grad(create_graph=True) on the predicted x0
        x = x0.clone()
        S.set_conditions(x, cond[0][:B], cond[1][:B], True)
        e_uw = U.unet2d_forward(sd_grad[0], c_uw, x, tb)
        e_w = U.unet2d_forward(sd_grad[1], c_w, S.w_model_input(x), tb)

        def nablaJ(x_start):                                  # get_nablaJ :34-49 over ddpm_guidance_loss (utils.py:1289-1328)
            x_start.requires_grad_(True)                      # (already part of the denoisers' graph, as in the reference)
            u, f = x_start[:, 0, :11], x_start[:, 1, :10]
            m = torch.ones(128)
            m[32:96] = 0
            J = 0.0 * ((((u[:, 0] - ut[:, 0]) ** 2 + (u[:, 10] - ut[:, 10]) ** 2) * m).mean(-1)) + 0.0 * (f ** 2).sum((-1, -2)) \
                + 0.0 * ((u[:, 1:] - u[:, :-1]) ** 2).sum((-1, -2))                          # shipped scripts: all-zero weights
            return torch.autograd.grad(J, x_start, grad_outputs=torch.ones_like(J), retain_graph=True, create_graph=True,
                                       allow_unused=True)[0].detach()
        S.p_sample_step(sched, x, 999, e_uw, e_w, z, grad_fn=nablaJ, **kw)

    legs = {}
    for name, fn, share in (("no_grad", step_nograd, 0.4), ("grad_enabled", step_grad, 0.6)):
        warm, times = _timed_cpu_steps(fn, budget_s * share, max_steps=20)
        mean = sum(times) / len(times)
        legs[name] = {"B": B, "warmup_s": round(warm, 3), "timed_steps": len(times), "mean_s_per_step": round(mean, 4),
                      "trajectories_per_s": B / (STEPS_PER_TRAJECTORY * mean)}
    return {"value": max(l["trajectories_per_s"] for l in legs.values()), "unit": "trajectories/s", "cores": cores, "kind": "port",
            "value_no_grad": legs["no_grad"]["trajectories_per_s"], "value_grad_enabled": legs["grad_enabled"]["trajectories_per_s"],
            "sample": f"guided DDPM steps (joint+prior Unet2D forward + update) at B={B}, 16x128, torch fp32 on {cores} threads, "
                      + "; ".join(f"{k}: 1 warm-up ({l['warmup_s']} s) + {l['timed_steps']} timed, {l['mean_s_per_step']} s/step" for k, l in legs.items())
                      + "; extrapolated x1000 steps.  Plan of record (BASELINE.md 3): B=50, 20 timed steps per mode -- the step counts above are what "
                        f"the {budget_s:.0f} s budget allowed; grad_enabled = the reference's own mode (diffusion_1d_burgers.py:525)",
            "legs": legs}


def burgers_fd_leg(ctx, with_cpu, budget_s):
    """Burgers finite-difference evaluator (generate_burgers.py:207): N = 50 trajectories (the entry script's batch) and
    N = 256, 10 000 explicit Euler steps in ONE launch, bit-exact vs the reference; CPU = the NumPy oracle, full 10 000 steps at N = 50
    (BASELINE.md section 3), on one core (NumPy does not thread these element-wise ops)."""
    import numpy as np
    from oracle import burgers as OB
    from diffphycon_amd.evaluators import burgers_numeric_solve_free
    out = {"metric": "Burgers FD rollouts/s (128 cells, 10 000 Euler steps)", "unit": "rollouts/s", "dtype": "f32 (bit-exact vs the reference)"}
    for n in (50, 256):
        u0, f = OB.synthetic_inputs(n, seed=3)
        ud, fd = torch.from_numpy(u0).to(ctx.device), torch.from_numpy(f).to(ctx.device)
        burgers_numeric_solve_free(ud, fd, visc=0.01, T=1.0, dt=1e-4, num_t=10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = burgers_numeric_solve_free(ud, fd, visc=0.01, T=1.0, dt=1e-4, num_t=10)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"n{n}"] = {"seconds": dt, "rollouts_per_s": n / dt}
        if n == 50:
            keep = (u0, f, got.cpu().numpy())
    out["value"] = out["n256"]["rollouts_per_s"]
    if with_cpu:
        u0, f, got = keep
        t0 = time.perf_counter()
        ref = OB.burgers_numeric_solve_free(u0, f, 0.01, 1.0, 1e-4, 10)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 50 / dt, "unit": "rollouts/s", "cores": 1, "kind": "port",
                               "sample": f"N=50, the full 10 000 steps (BASELINE.md 3) on the NumPy oracle: {dt:.2f} s",
                               "gpu_equals_cpu_bit_for_bit": bool(np.array_equal(ref, got))}
    else:
        out["cpu_baseline"] = None
        out["cpu_baseline_null_reason"] = "CPU legs run on rank 0 at N=1 only (or --no-cpu-baseline)"
    return out


def _worker_ready(i):
    return i


def _smoke_oracle_rollout(args):
    """One rollout on the NumPy oracle (a worker of smoke_evaluator's `cores`-process CPU leg)."""
    import numpy as np
    from oracle import smoke_solver as O
    d0, c1, c2, T = args
    t0 = time.perf_counter()
    O.solver(O.init_sim_128(), O.init_velocity_(), d0, c1, c2, per_timelength=T)
    return time.perf_counter() - t0


class Ctx:
    """rank / world / device / optional torch.distributed handle + the barrier-bracketed timing helpers."""

    def __init__(self, rank, world, device, dist, stub=False):
        self.rank, self.world, self.device, self.dist, self.stub = rank, world, device, dist, stub
        self.seen_world = dist.get_world_size() if dist is not None else 1      # what the backend (RCCL / gloo) itself reports

    def sync(self):
        if not self.stub:
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            if not self.stub:
                torch.cuda.synchronize()

    def reduce(self, seconds):
        """(max, min) of a per-rank duration over the ranks."""
        if self.dist is None:
            return seconds, seconds
        hi = torch.tensor([seconds], device=self.device, dtype=torch.float64)
        lo = hi.clone()
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        return hi.item(), lo.item()

    def log(self, t_start, msg):
        if self.rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)


def timed_loop(ctx, step, steps, warmup, profile=True):
    """W untimed warm-up steps (the last one with every kernel class bracketed by events: that picks the dominant class), then
    EXACTLY K timed steps between barrier + synchronize pairs with only the dominant class instrumented (two events per launch
    cost ~4 % of a step when every launch is bracketed), then -- OUTSIDE the timed region -- ONE more step with every class
    bracketed: the per-class breakdown and `kernel_time_fraction_of_step` come from that steady-state step (r04 took them on
    the last warm-up step, which with --warmup 1 is the very first step of the process: module loads, first-touch allocations;
    VERDICT r04 weak #3).  -> (max-over-ranks seconds per step, min, prof_all, prof_dom, ms of the breakdown step)."""
    from diffphycon_amd import _lib
    prof_warm = None
    for i in range(warmup):
        last = profile and i == warmup - 1
        if last:
            _lib.profile_begin()
        step()
        ctx.sync()
        if last:
            prof_warm = _lib.profile_end()
    ctx.sync()                               # barrier + device sync on both sides of the timed region (also when --warmup 0)
    dom = max(prof_warm.items(), key=lambda kv: kv[1]["total_ms"])[0] if prof_warm else None
    if profile:
        _lib.profile_begin([dom] if dom else None)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_end() if profile else None
    hi, lo = ctx.reduce(elapsed)
    prof_all, steady_ms = None, None
    if profile:
        _lib.profile_begin()
        tw0 = time.perf_counter()
        step()
        ctx.sync()
        steady_ms = (time.perf_counter() - tw0) * 1e3
        prof_all = _lib.profile_end()
    return hi / steps, lo / steps, prof_all, prof, steady_ms


def roofline_of(prof, prof_all, modes, sec_per_step, steps, warm_ms, unit_tflop_per_step, traffic_scale=1.0, pmc_workload=None):
    """pmc_workload: which committed PMC passes apply -- "smoke" (profiles/pmc_traffic.json: the S64 headline loop in the default
    arithmetic), "burgers" / "train" (profiles/pmc_traffic_<leg>.json: that leg's own launch shapes), None: no counters were taken for
    this leg's launches (another shape, or another kernel behind the same class name): traffic null."""
    name, d = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
    pmc = (lambda n: pmc_traffic(n, pmc_workload)) if pmc_workload else (lambda _name: None)
    if d["flops"] > 0:
        achieved = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
        peak, peak_note = mfma_roof(name, modes)
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": pmc(name), "peak_note": peak_note}
    else:
        achieved = d["bytes"] / (d["total_ms"] * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": pmc(name)}
    if roof["traffic"] is not None:
        roof["traffic"] *= traffic_scale
    if roof["traffic"] is None:
        roof["traffic_source"] = None
    elif pmc_workload == "smoke":
        roof["traffic_source"] = ("committed rocprofv3 --pmc passes (profiles/pmc_traffic.json: per launch of an 8-trajectory micro-batch, "
                                  f"scaled x{traffic_scale:g} to this run's launch size), not re-measured by this run")
    else:
        roof["traffic_source"] = (f"committed rocprofv3 --pmc passes of this leg at this batch (profiles/pmc_traffic_{pmc_workload}.json, "
                                  "tools/pmc_legs.sh: average over the class's launches), not re-measured by this run")
    if roof["bound"] == "mfma":
        roof.update(mfma_issue(name, modes, roof["achieved"]))
        roof.update(pmc_sq(name, pmc_workload))
    roof["kernel"], roof["launches"] = name, d["launches"]
    roof["avg_launch_ms"] = d["total_ms"] / max(d["launches"], 1)
    if prof_all:
        roof["breakdown_ms_per_step"] = {k: round(v["total_ms"], 3) for k, v in sorted(prof_all.items())}
        roof["breakdown_note"] = ("all classes bracketed by events on ONE extra steady-state step run after (outside) the timed region; "
                                  "the timed steps bracket only the roofline kernel class")
        roof["kernel_time_fraction_of_step"] = sum(v["total_ms"] for v in prof_all.values()) / warm_ms
    else:
        roof["breakdown_ms_per_step"] = {k: round(v["total_ms"] / steps, 3) for k, v in sorted(prof.items())}
    roof["step_flops_fraction_of_fp32_peak"] = (unit_tflop_per_step / sec_per_step) / PEAK_FP32_MFMA_TFLOPS
    return roof


def run_burgers(ctx, args, t_start, batch=256, with_cpu=True, exact=True):
    """Burgers POPC, 1000-step DDPM, batch 256 per GPU (BASELINE.json configs[1]); HIP-graph replay of the step when the
    per-kernel events are off (--burgers-graph, default on for the exact/graph legs)."""
    gd, kwargs, sd_cpu, cond = burgers_setup(ctx.device, batch, ctx.rank)
    gd.noise_seed, gd.traj_offset, gd.guidance_batch = 0, ctx.rank * batch, batch
    guide = kwargs["nablaJ"]
    state = {"t": 999}

    def make_step(g):
        img = g.sample_noise([batch, 2, 16, 128], ctx.device)
        x_w = torch.empty_like(img)

        def step():
            t = state["t"]
            g._prepare(img, x_w, kwargs["u_init"], kwargs["u_final"])
            e_uw, e_w = g._denoise_step(img, x_w, t)          # HIP-graph replay when g.use_graph
            z = g.sample_noise([batch, 2, 16, 128], ctx.device)
            g._update(img, e_uw, e_w, z, None, img, g._coef(t, guide, kwargs["J_scheduler"], kwargs["w_scheduler"], True, batch))
            state["t"] = t - 1 if t > 1 else 999
        return step, img
    step, img = make_step(gd)
    # leg 1: eager launches with per-kernel events (roofline + per-class breakdown); leg 2 (the reported value): the production
    # form -- the two denoiser forwards replayed from a HIP graph, no events
    gd.use_graph = False
    sec_e, _, prof_all, prof, warm_ms = timed_loop(ctx, step, args.steps, max(args.warmup, 1))
    assert torch.isfinite(img).all()
    gd.use_graph = os.environ.get("DPC_BURGERS_GRAPH", "1") != "0"
    state["t"] = 999
    sec, sec_min, _, _, _ = timed_loop(ctx, step, args.steps, 2, profile=False)
    assert torch.isfinite(img).all()
    modes = gd.model_uw.modes
    out = {"metric": "guided trajectories/sec, 1D Burgers POPC 128 cells x 10 steps @1000 DDPM steps",
           "value": ctx.world * batch / (STEPS_PER_TRAJECTORY * sec), "unit": "trajectories/s", "n_gpus": ctx.world,
           "steps": args.steps, "ms_per_step": sec * 1e3, "ms_per_step_min_rank": sec_min * 1e3, "dtype": dtype_label(modes),
           "launch_mode": ("denoiser forwards replayed from a HIP graph (torch.cuda.CUDAGraph), update kernels eager"
                           if gd.use_graph else "eager launches"),
           "ms_per_step_eager_profiled": sec_e * 1e3, "world_size_seen_by_rccl": ctx.seen_world,
           "arithmetic": modes,
           "config": {"workload": "Burgers POPC (BASELINE.json configs[1]): 128 cells x 10 steps (16x128 padded), "
                                  f"1000-step guided DDPM, batch={batch} per GPU; one step = prepare + joint Unet2D(dim 64, "
                                  "mults 1-2-4-8-16) + prior Unet2D(1-2-4-8) + fused update",
                      "global_batch": ctx.world * batch, "parallelism": f"batch-shard x{ctx.world}"}}
    if ctx.rank == 0:
        out["roofline"] = roofline_of(prof, prof_all, modes, sec_e, args.steps, warm_ms, batch * BURGERS_UNIT_GFLOP / 1e3,
                                      pmc_workload="burgers" if batch == 256 else None)
        out["roofline"]["note"] = "per-kernel events need eager launches: measured on the eager leg (ms_per_step_eager_profiled)"
        out["roofline_step"] = step_roofline(modes, batch * BURGERS_UNIT_GFLOP * 1e9, sec,
                                             f"{batch} x {BURGERS_UNIT_GFLOP} GFLOP (SURVEY.md 8d) per step / ms_per_step")
    ctx.log(t_start, f"burgers: {sec * 1e3:.2f} ms/step ({sec_e * 1e3:.2f} eager with events)")
    if exact:
        gd_x, _, _, _ = burgers_setup(ctx.device, batch, ctx.rank, arithmetic="x6")
        gd_x.noise_seed, gd_x.traj_offset, gd_x.guidance_batch = 0, ctx.rank * batch, batch
        step_x, img_x = make_step(gd_x)
        state["t"] = 999
        sec_x, _, _, _, _ = timed_loop(ctx, step_x, args.steps, 1, profile=False)
        assert torch.isfinite(img_x).all()
        out["value_exact"] = ctx.world * batch / (STEPS_PER_TRAJECTORY * sec_x)
        out["ms_per_step_exact"] = sec_x * 1e3
        out["arithmetic_exact"] = gd_x.model_uw.modes
        del gd_x
    if with_cpu and ctx.rank == 0 and ctx.world == 1:
        out["cpu_baseline"] = burgers_cpu_baseline(sd_cpu, cond, args.cpu_budget * 0.5)
    else:
        out["cpu_baseline"] = None
        out["cpu_baseline_null_reason"] = "CPU legs run on rank 0 at N=1 only (or --no-cpu-baseline)"
    return out


TRAIN_FWD_GFLOP = 897.3               # one joint-denoiser forward of a 64x64x32 sample (half of UNIT_GFLOP's pair,
                                      # stem included); a training step costs ~3x (forward + backward-data + weight gradient)


def run_train(ctx, args, t_start, batch=16, with_cpu=True):
    """one optimizer step of the smoke joint denoiser (train_2d_smoke.py: Unet3D dim 64, mults (1,2,4), 6
    channels, 64x64x32, lr 1e-3) = q_sample + p_losses forward + hand-written backward + gradient all-reduce (N > 1) + gradient
    norm + fused clip/Adam/EMA, batch 16 per GPU (Trainer's default train_batch_size)."""
    from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion, Trainer
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    torch.manual_seed(0)
    m = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6)
    sd_cpu = {k: v.clone() for k, v in m.state_dict().items()}
    gd = GaussianDiffusion(m, image_size=SIZE, frames=FRAMES, timesteps=1000, sampling_timesteps=250, loss_type="l2",
                           objective="pred_noise", device=ctx.device)
    # the Trainer's defaults (r04): backward-data convolutions f16x3 under the dynamic loss scale; DPC_TRAIN_BWD_MODE=x6 times the exact mode alone
    bwd_mode = os.environ.get("DPC_TRAIN_BWD_MODE", "f16x3")
    ls_env = os.environ.get("DPC_TRAIN_LOSS_SCALE", "dynamic" if bwd_mode == "f16x3" else "1")
    loss_scale = ls_env if ls_env == "dynamic" else float(ls_env)
    tr = Trainer(gd, "Smoke", None, train_batch_size=batch * ctx.world, train_lr=1e-3, is_w_model=False, bwd_mode=bwd_mode,
                 loss_scale=loss_scale)
    g = torch.Generator().manual_seed(100 + ctx.rank)
    state = (torch.randn(batch, FRAMES, 6, SIZE, SIZE, generator=g) * 0.5).to(ctx.device)
    losses = []

    def step():
        losses.append(tr.train_step([state]))
    sec, sec_min, prof_all, prof, warm_ms = timed_loop(ctx, step, args.steps, max(args.warmup, 1))
    first, last = float(losses[0].item()), float(losses[-1].item())
    assert torch.isfinite(tr._t.w).all() and last == last, "non-finite weights / loss after the timed steps"
    tr.check_gradient_range()                 # (raises if an operand of the f16x3 kernels left its window during the timed steps)
    out = {"metric": "training samples/sec, 2D smoke 64x64x32 joint denoiser (p_losses forward + backward + clip + Adam + EMA)",
           "value": ctx.world * batch / sec, "unit": "samples/s", "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": sec * 1e3, "ms_per_step_min_rank": sec_min * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": dtype_label(getattr(gd.model, "modes", "")), "data": "synthetic",
           "world_size_seen_by_rccl": ctx.seen_world,
           "arithmetic": f"forward {gd.model.modes if hasattr(gd.model, 'modes') else ''}; backward-data convolutions {bwd_mode}"
                         f" (loss scale {loss_scale if isinstance(loss_scale, str) else format(loss_scale, 'g')}: 2^{int(__import__('math').log2(tr.loss_scale))} "
                         f"after the timed steps, {tr.skipped_steps} skipped); weight gradients: 3x3x3 convolutions f16x3 (wgrad3.hip), every "
                         "other geometry exact fp32 products on the native fp32 MFMA",
           "loss_first_last": [first, last],
           "config": {"workload": "S64 training step (SURVEY 8 f-4; scripts/smoke_train_joint.sh): Unet3D(dim 64, mults 1-2-4, 6 "
                                  f"channels) on 64x64 x 32 frames, batch={batch} per GPU, one optimizer step per bench step",
                      "global_batch": ctx.world * batch, "parallelism": f"data-parallel x{ctx.world} (one flat-gradient all-reduce)"}}
    if ctx.rank == 0:
        out["roofline"] = roofline_of(prof, prof_all, "", sec, args.steps, warm_ms, 3 * batch * TRAIN_FWD_GFLOP / 1e3,
                                      pmc_workload="train" if batch == 16 else None)
    ctx.log(t_start, f"train: {sec * 1e3:.1f} ms per optimizer step, loss {first:.4f} -> {last:.4f}")
    if bwd_mode == "f16x3":
        # the same step with EXACT backward-data products (bf16x6, loss scale 1: Trainer(bwd_mode="x6"), train_2d_smoke.py --bwd_mode x6;
        # DESIGN.md 8), as `value_exact` does for the sampling loop: reported beside the default, never as `value`
        del tr
        torch.cuda.empty_cache()
        tr6 = Trainer(gd, "Smoke", None, train_batch_size=batch * ctx.world, train_lr=1e-3, is_w_model=False, bwd_mode="x6")
        l6 = []
        sec6, sec6_min, _, _, _ = timed_loop(ctx, lambda: l6.append(tr6.train_step([state])), args.steps, max(args.warmup, 1), profile=False)
        f6, b6 = float(l6[0].item()), float(l6[-1].item())
        assert torch.isfinite(tr6._t.w).all() and b6 == b6, "non-finite weights / loss in the exact-backward leg"
        tr6.check_gradient_range()
        out["exact_backward"] = {"ms_per_step": sec6 * 1e3, "ms_per_step_min_rank": sec6_min * 1e3, "value": ctx.world * batch / sec6,
                                 "unit": "samples/s", "loss_first_last": [f6, b6],
                                 "note": "backward-data convolutions with exact fp32 products (bf16x6, direct 3x3x3 kernel), loss scale 1"}
        ctx.log(t_start, f"train (exact backward, x6): {sec6 * 1e3:.1f} ms per optimizer step")
        del tr6
    out["cpu_baseline"] = None
    out["cpu_baseline_null_reason"] = ("folded into the default line without its CPU leg (one oracle forward + autograd backward at B=1 is ~30 s); "
                                       "`python bench.py --workload train` times it")
    if with_cpu and ctx.rank == 0 and ctx.world == 1:
        out.pop("cpu_baseline_null_reason")
        from oracle import train_smoke as TS
        from oracle import unet3d as O
        cores = usable_cores()
        torch.set_num_threads(cores)
        cfg = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6)
        sched = TS.schedule(1000)
        x0 = state[:1].cpu()
        noise = torch.randn(x0.shape, generator=g)
        t0 = time.perf_counter()
        TS.loss_and_grads(sd_cpu, cfg, sched, x0, torch.tensor([500]), noise)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "samples/s", "cores": cores, "kind": "port",
                               "sample": f"ONE p_losses forward + torch-autograd backward of the CPU oracle at B=1, 64x64x32, fp32 on "
                                         f"{cores} threads ({dt:.1f} s; no optimizer step)"}
    return out


def run_s128_leg(ctx, args, t_start, batch=64):
    """BASELINE.json configs[4] (S128: 128 x 128 x 64 frames, prior-reweighted dual diffusion, batch 512 over 8 GPUs = 64 per GPU),
    micro-batch 8: the headline step at that extent and at the config's per-GPU batch (r04 ran B = 8)."""
    from diffphycon_amd.diffusion.diffusion_2d_smoke import SmokeGuidance
    mb = int(os.environ.get("DPC_S128_MICRO_BATCH", "8"))
    gd, _ = build_models(ctx.device, mb, frames=64, size=128)     # micro-batch 8 (B = 8, r03: 1: 361 ms per step, 2: 326, 4: 308; B = 16, r05: 4: 616.6, 8: 609.5)
    guide = SmokeGuidance((2.0, 18.0, 20.0, 16.0, 20.0, 1.0), 0.0)
    gd.noise_seed, gd.traj_offset = 0, ctx.rank * batch
    init = torch.nn.functional.interpolate(synthetic_init(batch, ctx.rank * batch)[:, None], scale_factor=2)[:, 0].to(ctx.device)
    x = gd.sample_noise([batch, 64, 6, 128, 128], ctx.device)
    x[:, 0, 0] = init
    st = {"t": 999}

    def step():
        gd.p_sample(None, x, st["t"], design_fn=guide, design_guidance="standard", init=init)
        st["t"] = st["t"] - 1 if st["t"] > 1 else 999
    sec, sec_min, _, _, _ = timed_loop(ctx, step, args.steps, args.warmup, profile=False)
    assert torch.isfinite(x).all()
    ctx.log(t_start, f"s128: {sec * 1e3:.1f} ms/step")
    return {"metric": "guided trajectories/sec, 2D smoke 128x128x64 @1000 DDPM steps", "value": ctx.world * batch / (STEPS_PER_TRAJECTORY * sec),
            "unit": "trajectories/s", "n_gpus": ctx.world, "steps": args.steps, "ms_per_step": sec * 1e3,
            "ms_per_step_min_rank": sec_min * 1e3, "world_size_seen_by_rccl": ctx.seen_world, "dtype": dtype_label(gd.model_joint.modes),
            "roofline_step": step_roofline(gd.model_joint.modes, batch * 14534.0e9, sec, f"{batch} x 14534 GFLOP (SURVEY.md 8d) per step / ms_per_step"),
            "config": {"workload": f"S128 (BASELINE.json configs[4]): 2D smoke 128x128 x 64 frames, batch={batch} per GPU (512 / 8), micro-batch {mb}",
                       "global_batch": ctx.world * batch, "parallelism": f"batch-shard x{ctx.world}"},
            "roofline": None, "roofline_null_reason": "per-class events are taken on the S64 headline only; roofline_step covers this leg",
            "cpu_baseline": None,
            "cpu_baseline_null_reason": "one S128 step of one trajectory is ~8x the S64 unit (14.5 TFLOP: about a minute on the host "
                                        "cores); the S64 cpu_baseline of this line is the CPU figure of record (BASELINE.md 3 plans none for S128)"}


def run_j128(ctx, args, t_start, batch=16, image_size=128, frames=20):
    """BASELINE.json configs[3] (J128): jellyfish full observation, 128 x 128 x 20 frames, joint (7 -> 4) + prior (7 -> 1) video
    U-Nets reweighted, design gradient through the two 2-D surrogates, batch 16 per GPU.  The entry script's own pipeline
    (inference/inference_2d_jellyfish.py --synthetic) runs at two chain lengths; the difference is the per-step time."""
    sys.path.insert(0, os.path.join(ROOT, "inference"))
    import inference_2d_jellyfish as J
    times = {}
    for T in (4, 4, 20):
        a = J.build_parser().parse_args(["--synthetic", "True", "--batch_size", str(batch), "--num_batches", "1", "--timesteps", str(T),
                                         "--sampling_timesteps", str(T), "--image_size", str(image_size), "--frames", str(frames),
                                         "--surrogate_dim", "64",      # the released surrogates' width (ForceUnet's head is Linear(512, .))
                                         "--inference_result_path", "/tmp/dpc_bench_j128"])
        a.device = ctx.device
        torch.manual_seed(0)
        J.load_normalization(a)
        force_model, diffusion, bd_updater, design_fn = J.load_model(a)
        ppl = J.InferencePipeline(diffusion, {"design_fn": design_fn, "design_guidance": a.design_guidance, "bd_updater": bd_updater},
                                  results_path=a.inference_result_path, args_general=a)
        ctx.sync()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sys.stderr):      # the entry script prints its results: stdout carries the ONE JSON line only
            ppl.run(J.synthetic_batches(a))
        ctx.sync()
        times[T] = time.perf_counter() - t0
        del ppl, diffusion, force_model, bd_updater, design_fn
        torch.cuda.empty_cache()
    sec, sec_min = ctx.reduce((times[20] - times[4]) / 16)
    ctx.log(t_start, f"j128: {sec * 1e3:.1f} ms per guided step")
    return {"metric": f"guided trajectories/sec, 2D jellyfish {image_size}x{image_size}x{frames} @1000 DDPM steps",
            "value": ctx.world * batch / (STEPS_PER_TRAJECTORY * sec), "unit": "trajectories/s", "n_gpus": ctx.world,
            "ms_per_step": sec * 1e3, "ms_per_step_min_rank": sec_min * 1e3, "world_size_seen_by_rccl": ctx.seen_world,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 storage/accumulate; f16x3 products", "data": "synthetic",
            "config": {"workload": f"J128 (BASELINE.json configs[3]): 2D jellyfish full-obs {image_size}x{image_size} x {frames} "
                                   f"frames, joint+prior Unet3D reweighted + design gradient on libdpc, batch={batch} per GPU; "
                                   "per-step time = (20-step run - 4-step run) / 16 of the entry script's pipeline",
                       "global_batch": ctx.world * batch, "parallelism": f"batch-shard x{ctx.world}"},
            "cpu_baseline": None,
            "cpu_baseline_null_reason": "no CPU restatement of the J128 step is timed: one guided step of one trajectory (two 20 x 128 x 128 video "
                                        "U-Nets + autograd through both surrogates) is minutes of host time; BASELINE.md 3 plans none"}


def run_e2e(ctx, args, t_start, ddim, ms_per_step=None, batch=LOCAL_BATCH, batches=1, overlap=True):
    """ONE real pass of the entry script's pipeline (inference/inference_2d_smoke.py main() on the synthetic test split, the
    reference's :179-197 + :317-427): `GaussianDiffusion.sample()` of `batch` trajectories per GPU -- 1000 guided DDPM steps
    (ddim=False) or the script's CLI default DDIM-100 (ddim=True, :511-517) -- then the PDE evaluator (solver_batch) + multi_evaluate,
    metric rows gathered over the ranks as the script does.  Nothing is extrapolated: seconds are wall seconds of that pass."""
    sys.path.insert(0, os.path.join(ROOT, "inference"))
    import inference_2d_smoke as S
    a = S.build_parser().parse_args(["--synthetic", "True", "--batch_size", str(batch), "--n_test", str(batch * ctx.world * batches),
                                     "--using_ddim", str(bool(ddim)), "--ddim_sampling_steps", "100", "--overlap_evaluator", str(bool(overlap)),
                                     "--inference_result_path", "/tmp/dpc_bench_e2e"])
    a.device, a.rank, a.world_size = ctx.device, ctx.rank, ctx.world
    a.inference_result_subpath = os.path.join(a.inference_result_path, f"{'ddim' if ddim else 'ddpm'}_{os.getpid()}")
    torch.manual_seed(0)                      # the same two random-init denoisers as the headline loop (build_models)
    with contextlib.redirect_stdout(sys.stderr):
        loader, rescaler = S.load_data(a)
        diffusion, design_fn = S.load_model(a, rescaler, a.w_energy, w_init=a.w_init)
        ppl = S.InferencePipeline(diffusion, {"design_fn": design_fn, "design_guidance": a.design_guidance}, rescaler,
                                  results_path=a.inference_result_subpath, args_general=a)
        # warm-up outside the timed pass: weight upload + packing, workspace allocation, module load of the evaluator
        st0 = next(iter(loader))[0][:2]
        tiny = diffusion[0].sampling_timesteps, diffusion[0].is_ddim_sampling
        diffusion[0].sampling_timesteps, diffusion[0].is_ddim_sampling = 2, True
        ppl.multi_evaluate(ppl.run_model(st0), st0)
        diffusion[0].sampling_timesteps, diffusion[0].is_ddim_sampling = tiny
        spent = {"sample": 0.0, "evaluate": 0.0}
        sample_events = []
        overlapped = bool(overlap) and batches > 1

        def timed(name, fn):
            def wrapped(*x, **kw):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = fn(*x, **kw)
                torch.cuda.synchronize()
                spent[name] += time.perf_counter() - t0
                return out
            return wrapped

        def timed_on_stream(fn):
            # overlapped schedule: a device-wide synchronize would serialise the side stream's rollouts with the sampling; the sampling
            # time is taken with events on the sampling stream instead (it INCLUDES what the co-resident rollouts cost the sampling)
            def wrapped(*x, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*x, **kw)
                e1.record()
                sample_events.append((e0, e1))
                return out
            return wrapped
        if overlapped:
            ppl.run_model = timed_on_stream(ppl.run_model)
        else:
            ppl.run_model = timed("sample", ppl.run_model)
            ppl.multi_evaluate = timed("evaluate", ppl.multi_evaluate)
        ctx.sync()
        t0 = time.perf_counter()
        J = ppl.run(loader)
        ctx.sync()
        total = time.perf_counter() - t0
        if overlapped:
            spent["sample"] = sum(e0.elapsed_time(e1) for e0, e1 in sample_events) * 1e-3
            spent["evaluate"] = max(total - spent["sample"], 0.0)         # what the evaluator still adds to the wall clock
    total, _ = ctx.reduce(total)
    samp, samp_min = ctx.reduce(spent["sample"])
    evl, _ = ctx.reduce(spent["evaluate"])
    steps = 100 if ddim else STEPS_PER_TRAJECTORY
    out = {"pipeline": "inference/inference_2d_smoke.py main(): sample() + solver_batch + multi_evaluate, synthetic test split, "
                       f"random-init denoisers, {'DDIM-100, eta 1 (the CLI default)' if ddim else '1000-step guided DDPM'}",
           "batch_per_gpu": batch, "batches": batches, "n_gpus": ctx.world, "world_size_seen_by_rccl": ctx.seen_world, "sampling_steps": steps,
           "schedule": ("evaluator of batch i on a side stream under the sampling of batch i + 1 (InferencePipeline.run, r05); sample_seconds = "
                        "event time on the sampling stream, evaluate_seconds = wall - sample_seconds = what the evaluator still adds"
                        if overlapped else "serial: sample, evaluate, next batch (the reference's loop, inference_2d_smoke.py:259-271)"),
           "sample_seconds": samp, "sample_seconds_min_rank": samp_min, "evaluate_seconds": evl, "wall_seconds": total,
           "trajectories_per_s_sampling": ctx.world * batch * batches / samp,
           "trajectories_per_s_end_to_end": ctx.world * batch * batches / total,
           "end_to_end_over_sampling_only": samp / total,
           "ms_per_step_measured": samp / (steps * batches) * 1e3, "unit": "trajectories/s", "measured": "wall clock of one complete pass, not extrapolated",
           "metric_row": {k: float(v.reshape(-1)[0]) for k, v in J.items()},
           "cpu_baseline": None,
           "cpu_baseline_null_reason": "a whole pass on the host cores would take hours (cpu_baseline of the headline: per-step rate x 1000); "
                                       "the evaluator's share is timed in smoke_evaluator.cpu_baseline"}
    if ms_per_step is not None:
        ratio = (samp / (steps * batches) * 1e3) / ms_per_step
        out["vs_headline_ms_per_step"] = ratio
        out["agrees_with_headline_within_2pct"] = bool(abs(ratio - 1.0) <= 0.02)
        if not out["agrees_with_headline_within_2pct"]:
            ctx.log(t_start, f"WARNING: e2e ms per step is {ratio:.4f} x the timed-loop ms_per_step (expected within 2 %)")
    ctx.log(t_start, f"e2e {'ddim100' if ddim else 'ddpm1000'}: sample {samp:.1f} s + evaluate {evl:.1f} s")
    del ppl, diffusion
    torch.cuda.empty_cache()
    return out


def run_smoke_evaluator(ctx, B=64, T=256, with_cpu=True):
    """Post-sampling PDE evaluator: B rollouts x T frames in one launch, as multi_evaluate consumes them.
    cpu_baseline (evaluate_solver.py:205): the NumPy oracle of the same rollout -- bit-identical to the
    reference's phi run -- on ONE process (the plan's "31 steps extrapolated to 255" is run in full: all 255 steps of one rollout) and
    on `cores` processes, one rollout each (the reference's own parallelism: inference_2d_smoke.py:339-364 forks one process per
    trajectory)."""
    import numpy as np
    from diffphycon_amd.dataset.apps import evaluate_solver as E
    rng = np.random.default_rng(0)
    c1 = (rng.standard_normal((B, 32, 64, 64)) * 0.3).astype(np.float32)
    c2 = (rng.standard_normal((B, 32, 64, 64)) * 0.3).astype(np.float32)
    c1[:, :, 8:56, 8:56] = 0
    c2[:, :, 8:56, 8:56] = 0
    d0 = synthetic_init(B, 0).numpy() * 2
    sim = E.init_sim_128()
    c1d, c2d, d0d = (torch.from_numpy(a).to(ctx.device) for a in (c1, c2, d0))
    kw = dict(frame_stride=8, space_stride=2, density_dtype=torch.float32)
    E.solver_batch(sim, E.init_velocity_(), d0d[:2], c1d[:2, :2], c2d[:2, :2], 2, **kw)          # warm-up / module load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = E.solver_batch(sim, E.init_velocity_(), d0d, c1d, c2d, T, return_cg_iterations=True, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = out[4].double()
    res = {"rollouts_per_s": B / dt, "seconds": dt, "batch": B, "frames": T, "mean_cg_iterations_per_frame": its.mean().item(),
           "us_per_cg_iteration": dt * 1e6 / max(its.sum().item() / B, 1.0),
           "note": "smoke PDE rollouts (phi advect + masked CG pressure solve, fp64, bit-exact vs the reference) of 64 "
                   "sampled control sequences; end-to-end figure next to trajectories/s (SURVEY.md 8d)"}
    if with_cpu:
        import multiprocessing as mp
        cores = usable_cores()
        one = _smoke_oracle_rollout((d0[0], c1[0], c2[0], T))
        # "spawn", not "fork" (ADVICE r05): the parent has initialised torch and the HIP runtime by now, and a forked child can inherit a
        # locked allocator / runtime mutex.  Spawned workers re-import this module (NumPy oracle only, no GPU context); one untimed
        # round-trip per worker first, so that their start-up is not in the timed region.
        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_worker_ready, range(cores))
            t0 = time.perf_counter()
            per = pool.map(_smoke_oracle_rollout, [(d0[i % B], c1[i % B], c2[i % B], T) for i in range(cores)])
            wall = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": cores / wall, "unit": "rollouts/s", "cores": cores, "kind": "port",
                               "value_one_process": 1.0 / one,
                               "sample": f"NumPy oracle (bit-identical to the reference's phi rollout), {T} frames at 128^2: one process, one "
                                         f"rollout: {one:.2f} s; {cores} processes, one rollout each: {wall:.2f} s wall "
                                         f"(slowest worker {max(per):.2f} s)"}
    else:
        res["cpu_baseline"] = None
        res["cpu_baseline_null_reason"] = "--no-cpu-baseline"
    return res


# ------------------------------------------------------------------------------------------------ launch plumbing
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this host driver (RCCL needs it)
    env["DPC_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="smoke", choices=["smoke", "burgers", "s128", "train", "j128", "e2e", "ddim100"],
                    help="smoke = BASELINE.json's headline metric S64 (default); burgers = configs[1]; s128 = configs[4] shape "
                         "(128x128x64 frames; builder-side line, batch 8 per GPU by default)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (S64 = 64)")
    ap.add_argument("--micro-batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=420.0,
                    help="seconds of host time for all cpu_baseline legs (420: BASELINE.md section 3's step counts -- 5 timed steps at B = 1 and "
                         "B = min(8, cores), 20 Burgers steps per mode -- fit; 60 gives the short r05 sample)")
    ap.add_argument("--no-extras", action="store_true", help="headline loop only: no exact-mode / burgers / evaluator legs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the two real end-to-end passes (DDIM-100: ~30 s, DDPM-1000: ~270 s)")
    ap.add_argument("--stub", action="store_true",
                    help="TEST ONLY: CPU + gloo, the step is a sleep; exercises the launcher / barrier / reduction plumbing")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or omit the launcher)")
    dist = None
    if args.stub:
        device = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL over xGMI
    ctx = Ctx(rank, world, device, dist, stub=args.stub)
    seen_world = ctx.seen_world
    launcher = "self (bench.py -> torch.distributed.run)" if os.environ.get("DPC_BENCH_SELF_LAUNCHED") else \
        ("external torch.distributed.run" if world > 1 else "single process")

    if args.stub:
        B = args.batch or LOCAL_BATCH

        def step():
            time.sleep(0.002 * (1 + rank))           # ranks differ on purpose: the reduction must report the slowest
        for _ in range(args.warmup):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        ctx.sync()
        hi, lo = ctx.reduce(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"metric": "STUB (no GPU work): launcher plumbing test", "stub": True, "n_gpus": world,
                              "value": world * B / (STEPS_PER_TRAJECTORY * hi / args.steps), "unit": "trajectories/s",
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": hi / args.steps * 1e3,
                              "ms_per_step_min_rank": lo / args.steps * 1e3, "world_size_seen_by_backend": seen_world,
                              "launcher": launcher, "config": {"global_batch": world * B}}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload == "burgers":
        out = run_burgers(ctx, args, t_start, batch=args.batch or 256, with_cpu=not args.no_cpu_baseline,
                          exact=not args.no_extras)
        if rank == 0:
            out.update({"warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                        "data": "synthetic", "world_size_seen_by_rccl": seen_world, "launcher": launcher})
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload == "j128":
        out = run_j128(ctx, args, t_start, batch=args.batch or 16)
        if rank == 0:
            out.update({"steps": 16, "warmup": 4, "world_size_seen_by_rccl": seen_world, "launcher": launcher, "roofline": None,
                        "cpu_baseline": None})
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload in ("e2e", "ddim100"):
        nb = 2 if args.workload == "ddim100" else 1
        out = run_e2e(ctx, args, t_start, ddim=args.workload == "ddim100", batch=args.batch or LOCAL_BATCH, batches=nb)
        if nb > 1:
            ser = run_e2e(ctx, args, t_start, ddim=True, batch=args.batch or LOCAL_BATCH, batches=nb, overlap=False)
            out["serial_schedule"] = {k: ser[k] for k in ("schedule", "sample_seconds", "evaluate_seconds", "wall_seconds",
                                                            "trajectories_per_s_sampling", "trajectories_per_s_end_to_end", "metric_row")}
            out["metric_rows_equal_serial"] = out["metric_row"] == ser["metric_row"]
        if rank == 0:
            out.update({"launcher": launcher, "roofline": None, "cpu_baseline": None})
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.workload == "train":
        out = run_train(ctx, args, t_start, batch=args.batch or 16, with_cpu=not args.no_cpu_baseline)
        if rank == 0:
            out.update({"world_size_seen_by_rccl": seen_world, "launcher": launcher})
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from diffphycon_amd import _lib
    from diffphycon_amd.diffusion.diffusion_2d_smoke import SmokeGuidance
    s128 = args.workload == "s128"
    frames, size = (64, 128) if s128 else (FRAMES, SIZE)
    B = args.batch or (8 if s128 else LOCAL_BATCH)
    # micro-batch 32: 2 forwards per net and step; the persistent conv kernel then walks 64 tiles per workgroup on the largest layers
    # (r02, one box: 4: 311.1 ms per step, 8: 289.3, 16: 286.1; r03, one box, interleaved: 16: 282.9 / 284.2, 32: 279.4 / 280.0,
    # 64: 279.1); the activation workspace is ~20 GB of the 288; outputs are bit-identical for every micro-batch
    # (tests/test_gpu_unet3d.py: test_unet3d_bench_micro_batches_are_bit_identical, B = 32 at micro-batch 4 / 16 / 32)
    mbatch = args.micro_batch or (4 if s128 else 32)      # (s128, r03 one box, interleaved: 1: 361.1 ms per step, 2: 326.3, 4: 308.4)
    unit_gflop = 14534.0 if s128 else UNIT_GFLOP
    gd, sd_cpu = build_models(device, mbatch, frames=frames, size=size)
    guide = SmokeGuidance((2.0, 18.0, 20.0, 16.0, 20.0, 1.0), 0.0)

    def make_step(g):
        g.noise_seed, g.traj_offset = 0, rank * B            # batch-sharded: rank r owns trajectories [r*B, (r+1)*B)
        init = (synthetic_init(B, rank * B, size=size) if not s128 else
                torch.nn.functional.interpolate(synthetic_init(B, rank * B)[:, None], scale_factor=2)[:, 0]).to(device)
        x = g.sample_noise([B, frames, 6, size, size], device)
        x[:, 0, 0] = init
        st = {"t": 999}

        def step():
            g.p_sample(None, x, st["t"], design_fn=guide, design_guidance="standard", init=init)
            st["t"] = st["t"] - 1 if st["t"] > 1 else 999
        return step, x

    step, x = make_step(gd)
    ctx.log(t_start, "models built, starting warmup")
    sec, sec_min, prof_all, prof, warm_ms = timed_loop(ctx, step, args.steps, args.warmup)
    ctx.log(t_start, f"timed steps done: {sec * 1e3:.2f} ms/step")
    assert torch.isfinite(x).all(), "non-finite state after the timed steps"
    modes = gd.model_joint.modes
    out = None
    if rank == 0:
        roof = roofline_of(prof, prof_all, modes, sec, args.steps, warm_ms, B * unit_gflop / 1e3,
                           traffic_scale=(min(mbatch, B) / 8.0) if not s128 else 1.0, pmc_workload=None if s128 else "smoke")
        cfg_name = ("S128 shape (BASELINE.json configs[4]): 2D smoke 128x128 x 64 frames" if s128 else
                    "S64 (BASELINE.json configs[2]): 2D smoke 64x64 x 32 frames")
        out = {
            "metric": ("guided trajectories/sec, 2D smoke 128x128x64 @1000 DDPM steps" if s128 else
                       "guided trajectories/sec, 2D smoke 64x64x32 @1000 DDPM steps"),
            "value": world * B / (STEPS_PER_TRAJECTORY * sec), "unit": "trajectories/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": sec * 1e3, "ms_per_step_min_rank": sec_min * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype_label(modes), "data": "synthetic",
            "arithmetic": modes + " (reported by the library: dpc_unet3d_modes). fp32 tensors in HBM and fp32 accumulation "
                          "everywhere; f16x3 = each fp32 operand split into 2 fp16 terms (22 significant bits), 3 MFMAs per "
                          "product; x6 = exact 3-way bf16 split, 6 MFMAs; f32 = native fp32 MFMA. value_exact re-times the same "
                          "loop in x6",
            "conv3d_algorithm": _lib.lib().dpc_conv3d_algorithm().decode() + " (reported by the library: dpc_conv3d_algorithm; "
                                "winograd_f43_frames = F(4,3) minimal filtering along the frame axis, 54 tap products per four output "
                                "frames where the direct form has 108 and F(2,3) 72; roofline.achieved counts algorithmic direct-form "
                                "FLOP either way, roofline.mfma_issue_frac what is issued)",
            "config": {"workload": f"{cfg_name}, 1000-step guided DDPM, batch={B} per GPU; one step = joint+prior Unet3D(dim 64, "
                                   "mults 1-2-4) forward + fused guidance/posterior update; trajectories/s = batch/(1000*s_per_step)",
                       "global_batch": world * B, "micro_batch": mbatch, "parallelism": f"batch-shard x{world}"},
            "world_size_seen_by_rccl": seen_world, "launcher": launcher,
            "roofline": roof,
            "roofline_step": step_roofline(modes, B * unit_gflop * 1e9, sec,
                                           f"{B} x {unit_gflop} GFLOP (SURVEY.md 8d: both U-Nets, direct-form flop) per step / ms_per_step"),
        }
    if not args.no_extras:
        # ---- exact-product leg: same loop, arithmetic mode x6 (3-way bf16 split of both operands, 6 MFMAs per fp32 product)
        del step, x
        gd_x, _ = build_models(device, mbatch, arithmetic="x6", frames=frames, size=size)
        step_x, x_x = make_step(gd_x)
        steps_x = min(args.steps, 5)
        sec_x, _, prof_all_x, prof_x, warm_ms_x = timed_loop(ctx, step_x, steps_x, 1)
        assert torch.isfinite(x_x).all()
        if rank == 0:
            out["value_exact"] = world * B / (STEPS_PER_TRAJECTORY * sec_x)
            out["ms_per_step_exact"] = sec_x * 1e3
            out["arithmetic_exact"] = gd_x.model_joint.modes
            out["dtype_exact"] = dtype_label(gd_x.model_joint.modes)
            out["roofline_exact"] = roofline_of(prof_x, prof_all_x, gd_x.model_joint.modes, sec_x, steps_x, warm_ms_x, B * unit_gflop / 1e3)
            out["roofline_exact"]["traffic"], out["roofline_exact"]["traffic_source"] = None, None   # (PMC passes cover the default mode)
            out["roofline_step_exact"] = step_roofline(gd_x.model_joint.modes, B * unit_gflop * 1e9, sec_x, "as roofline_step, x6 leg")
        del gd_x, step_x, x_x
        torch.cuda.empty_cache()
        ctx.log(t_start, f"exact-mode leg done: {sec_x * 1e3:.1f} ms/step")
        if not s128:
            bo = run_burgers(ctx, args, t_start, with_cpu=not args.no_cpu_baseline)
            if rank == 0:
                out["burgers"] = bo
            if rank == 0 and world == 1:
                out["smoke_evaluator"] = run_smoke_evaluator(ctx, with_cpu=not args.no_cpu_baseline)
                ctx.log(t_start, "evaluator leg done")
                out["burgers_fd"] = burgers_fd_leg(ctx, not args.no_cpu_baseline, args.cpu_budget)
                ctx.log(t_start, "Burgers FD leg done")
            # the other BASELINE configs and the training step, short legs folded into the same line
            del gd
            torch.cuda.empty_cache()
            extra = argparse.Namespace(**vars(args))
            extra.steps, extra.warmup = min(args.steps, 3), 1
            legs = {}
            legs["s128"] = run_s128_leg(ctx, extra, t_start)
            legs["j128"] = run_j128(ctx, extra, t_start)
            legs["train"] = run_train(ctx, extra, t_start, with_cpu=False)
            # the CLI default (DDIM-100) and ONE real 1000-step trajectory batch through the entry script's pipeline, evaluator included
            if not args.no_e2e:
                legs["ddim100"] = run_e2e(ctx, args, t_start, ddim=True, batches=2)        # 2 batches: the evaluator overlap is in the schedule
                legs["e2e"] = run_e2e(ctx, args, t_start, ddim=False, ms_per_step=sec * 1e3)
            if rank == 0:
                out.update(legs)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and not s128:
            out["cpu_baseline"] = cpu_baseline(sd_cpu, args.cpu_budget * 0.8)
        else:
            out["cpu_baseline"] = None
        # short scalars first (a truncated line still carries both arithmetic modes' values), then the two contract objects, then the legs
        front = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_min_rank", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "value_exact", "ms_per_step_exact", "dtype_exact", "data", "config", "roofline", "cpu_baseline"]
        out = {**{k: out[k] for k in front if k in out}, **{k: v for k, v in out.items() if k not in front}}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
