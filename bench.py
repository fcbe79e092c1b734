#!/usr/bin/env python
"""Headline benchmark: guided trajectories/sec, 2-D smoke 64x64x32 @ 1000 DDPM steps (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is ONE guided DDPM step of the whole local batch (config S64: 64 trajectories per GPU):
joint U-Net forward + prior U-Net forward + fused guidance/posterior update — i.e. 1/1000 of the sampling of
each trajectory.  trajectories/s = (N * 64) / (1000 * seconds_per_step).  Inputs and state are resident in
HBM before the timed region; weights are seeded random initialisations of the reference architecture
(no checkpoints exist offline) and data is synthetic.

The JSON line also carries
  roofline     : achieved fp32 TFLOP/s of the dominant kernel class (the implicit-GEMM conv kernel), from HIP
                 events recorded on the launch stream inside the timed region, against the 157.3 TF fp32 MFMA peak;
  cpu_baseline : the CPU oracle (torch fp32, all host cores) executing the same step at B=1, timed once.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3         # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0        # same guide: v_mfma_f32_32x32x16_bf16 / _f16 dense peak


def mfma_roof(name):
    """Roof for the ALGORITHMIC fp32 flops of a kernel class under the active arithmetic mode: the split-operand kernels
    spend 3 (f16x3: 2-way fp16 split, default) or 6 (x6: exact 3-way bf16 split) 16-bit MFMAs per fp32 product."""
    if name.startswith("conv3x6"):
        mode = os.environ.get("DPC_CONV_MODE", "f16x3").lower()
    elif name.startswith("igemm"):
        mode = os.environ.get("DPC_IGEMM_MODE", "f16x3").lower()
    elif name.startswith("stem"):
        mode = os.environ.get("DPC_STEM_MODE", "f16x3").lower()
    else:
        mode = "f32"
    if mode.startswith("f3"):
        return PEAK_FP32_MFMA_TFLOPS, "fp32 MFMA dense"
    if mode.startswith("x") or mode.startswith("b"):
        return PEAK_BF16_MFMA_TFLOPS / 6.0, "2500 TF bf16 dense / 6 MFMAs per fp32 product (bf16x6 split)"
    return PEAK_BF16_MFMA_TFLOPS / 3.0, "2500 TF fp16 dense / 3 MFMAs per fp32 product (f16x3 split)"


def pmc_traffic(kernel_class):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per
    the guide's gfx950 correction, + WRITE_SIZE), or None when no PMC summary exists for that kernel class."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path)).get(kernel_class, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None
STEPS_PER_TRAJECTORY = 1000
LOCAL_BATCH = 64
FRAMES, SIZE = 32, 64


def synthetic_init(batch, traj0, size=SIZE):
    """zeros with a 5x5 block of ones at rows 10..25 / cols 12..52, init = density / 2."""
    init = torch.zeros(batch, size, size)
    g = torch.Generator().manual_seed(0)
    pos = torch.stack((torch.randint(10, 26, (4096,), generator=g), torch.randint(12, 53, (4096,), generator=g)), 1)
    for b in range(batch):
        r, c = pos[(traj0 + b) % 4096].tolist()
        init[b, r:r + 5, c:c + 5] = 0.5
    return init


def build_models(device, micro_batch):
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion
    torch.manual_seed(0)
    mj = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6, micro_batch=micro_batch)
    mw = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=2, micro_batch=micro_batch)
    sd_cpu = (mj.state_dict(), mw.state_dict())
    sd_cpu = tuple({k: v.clone() for k, v in sd.items()} for sd in sd_cpu)
    gd = GaussianDiffusion([mj.to(device), mw.to(device)], image_size=SIZE, frames=FRAMES, timesteps=1000,
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


def cpu_baseline(sd_cpu, init1):
    """One guided DDPM step, B=1, on the CPU oracle (the reference's algorithm restated in torch fp32)."""
    from oracle import unet3d as O
    from oracle import sampler_smoke as S
    cores = usable_cores()
    torch.set_num_threads(cores)
    cj = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6)
    cw = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=2)
    sched = S.make_schedule(1000, "sigmoid")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, FRAMES, 6, SIZE, SIZE, generator=g)
    z = torch.randn(1, FRAMES, 6, SIZE, SIZE, generator=g)
    x[:, 0, 0] = init1
    t = torch.full((1,), 999, dtype=torch.long)
    t0 = time.perf_counter()
    with torch.no_grad():
        ej = O.unet3d_forward(sd_cpu[0], cj, x, t)
        ew = O.unet3d_forward(sd_cpu[1], cw, x[:, :, 3:5], t)
        S.p_sample_step(sched, x, 999, ej, ew, z, init1, S.rescaler_tensor())
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (STEPS_PER_TRAJECTORY * dt), "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"1 guided DDPM step (joint+prior U-Net forward + update) at B=1, 64x64x32, torch fp32 on {cores} "
                      f"threads: {dt:.2f} s/step, extrapolated x1000 steps"}


def burgers_setup(device, batch, rank):
    """BASELINE.json configs[1]: Burgers POPC recipe of scripts/burgers_inference_partial_obs_partial_ctr.sh."""
    from diffphycon_amd.model.burgers_1d.unet import Unet2D
    from diffphycon_amd.diffusion import diffusion_1d_burgers as D
    torch.manual_seed(0)
    kw = dict(dim=64, out_dim=2, channels=2, resnet_block_groups=1)
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


def burgers_cpu_baseline(sd_cpu, cond):
    """One guided DDPM step of the POPC recipe at B=8 on the CPU oracle (torch fp32, all usable cores)."""
    from oracle import unet2d as U
    from oracle import sampler_burgers as S
    cores = usable_cores()
    torch.set_num_threads(cores)
    B = 8
    c_uw = U.Unet2DConfig(dim=64, dim_mults=(1, 2, 4, 8, 16), resnet_block_groups=1)
    c_w = U.Unet2DConfig(dim=64, dim_mults=(1, 2, 4, 8), resnet_block_groups=1)
    sched = S.make_schedule(1000, "cosine")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 2, 16, 128, generator=g)
    z = torch.randn(B, 2, 16, 128, generator=g)
    tb = torch.full((B,), 999, dtype=torch.long)
    t0 = time.perf_counter()
    with torch.no_grad():
        S.set_conditions(x, cond[0][:B], cond[1][:B], True)
        e_uw = U.unet2d_forward(sd_cpu[0], c_uw, x, tb)
        e_w = U.unet2d_forward(sd_cpu[1], c_w, S.w_model_input(x), tb)
        S.p_sample_step(sched, x, 999, e_uw, e_w, z, prior_beta=0.9, eta_w=S.scheduler_table("sigmoid_flip")[999],
                        eta_J=S.scheduler_table("cosine")[999])
    dt = time.perf_counter() - t0
    return {"value": B / (STEPS_PER_TRAJECTORY * dt), "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": f"1 guided DDPM step (joint+prior Unet2D forward + update) at B={B}, 16x128, torch fp32 on {cores} "
                      f"threads: {dt:.2f} s/step, extrapolated x1000 steps"}


def main_burgers(args, rank, world, device, dist):
    """`--workload burgers`: Burgers POPC, 1000-step DDPM, batch 256 per GPU (BASELINE.json configs[1])."""
    from diffphycon_amd import _lib
    B = args.batch if args.batch != LOCAL_BATCH else 256
    gd, kwargs, sd_cpu, cond = burgers_setup(device, B, rank)
    gd.noise_seed, gd.traj_offset, gd.guidance_batch = 0, rank * B, B
    guide = kwargs["nablaJ"]
    img = gd.sample_noise([B, 2, 16, 128], device)
    x_w = torch.empty_like(img)

    def step(t):
        gd._prepare(img, x_w, kwargs["u_init"], kwargs["u_final"])
        t_b = torch.full((B,), t, device=device, dtype=torch.long)
        e_uw, e_w = gd._denoise(img, x_w, t_b)
        z = gd.sample_noise([B, 2, 16, 128], device)
        gd._update(img, e_uw, e_w, z, None, img, gd._coef(t, guide, kwargs["J_scheduler"], kwargs["w_scheduler"], True, B))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    t_cur = 999
    prof_all = None
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if last:
            _lib.profile_begin()             # every class on an untimed step (see main())
        tw0 = time.perf_counter()
        step(t_cur)
        t_cur -= 1
        sync()
        warm_ms = (time.perf_counter() - tw0) * 1e3
        if last:
            prof_all = _lib.profile_end()
    sync()
    dom_name = max(prof_all.items(), key=lambda kv: kv[1]["total_ms"])[0] if prof_all else None
    _lib.profile_begin([dom_name] if dom_name else None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(t_cur)
        t_cur -= 1
    sync()
    elapsed = time.perf_counter() - t0
    prof = _lib.profile_end()
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    assert torch.isfinite(img).all()
    if rank == 0:
        sec = elapsed / args.steps
        name, d = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        achieved = d["flops"] / (d["total_ms"] * 1e-3) / 1e12 if d["flops"] > 0 else d["bytes"] / (d["total_ms"] * 1e-3) / 1e9
        peak, peak_note = mfma_roof(name)
        roof = ({"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                 "peak_note": peak_note}
                if d["flops"] > 0 else
                {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0})
        roof.update({"traffic": pmc_traffic(name), "kernel": name, "launches": d["launches"],
                     "avg_launch_ms": d["total_ms"] / max(d["launches"], 1),
                     "breakdown_ms_per_step": ({k: round(v["total_ms"], 3) for k, v in sorted(prof_all.items())} if prof_all else
                                               {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(prof.items())}),
                     "breakdown_note": "all classes on the last untimed warm-up step; timed steps bracket the roofline class only",
                     "kernel_time_fraction_of_step": (sum(v["total_ms"] for v in prof_all.values()) / warm_ms if prof_all else
                                                      sum(v["total_ms"] for v in prof.values()) / (elapsed * 1e3)),
                     "step_flops_fraction_of_fp32_peak": (B * 15.8 / 1e3 / sec) / PEAK_FP32_MFMA_TFLOPS})
        out = {"metric": "guided trajectories/sec, 1D Burgers POPC 128 cells x 10 steps @1000 DDPM steps",
               "value": world * B / (STEPS_PER_TRAJECTORY * sec), "unit": "trajectories/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "Burgers POPC (BASELINE.json configs[1]): 128 cells x 10 steps (16x128 padded), "
                                      f"1000-step guided DDPM, batch={B} per GPU; one step = prepare + joint Unet2D(dim 64, "
                                      "mults 1-2-4-8-16) + prior Unet2D(1-2-4-8) + fused update",
                          "global_batch": world * B, "parallelism": f"batch-shard x{world}"},
               "roofline": roof,
               "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else burgers_cpu_baseline(sd_cpu, cond)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="smoke", choices=["smoke", "burgers"],
                    help="smoke = BASELINE.json's headline metric (default); burgers = configs[1]")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=LOCAL_BATCH, help="trajectories per GPU (S64 = 64)")
    ap.add_argument("--micro-batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL over xGMI

    if args.workload == "burgers":
        return main_burgers(args, rank, world, device, dist)
    from diffphycon_amd import _lib
    from diffphycon_amd.diffusion.diffusion_2d_smoke import SmokeGuidance
    gd, sd_cpu = build_models(device, args.micro_batch)
    guide = SmokeGuidance((2.0, 18.0, 20.0, 16.0, 20.0, 1.0), 0.0)
    B = args.batch
    gd.noise_seed, gd.traj_offset = 0, rank * B            # batch-sharded: rank r owns trajectories [r*B, (r+1)*B)
    init_cpu = synthetic_init(B, rank * B)
    init = init_cpu.to(device)
    x = gd.sample_noise([B, FRAMES, 6, SIZE, SIZE], device)
    x[:, 0, 0] = init

    def step(t):
        gd.p_sample(None, x, t, design_fn=guide, design_guidance="standard", init=init)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:.1f}s] {msg}", file=sys.stderr, flush=True)

    t_cur = 999
    log("models built, starting warmup")
    prof_all = None
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if last:
            _lib.profile_begin()             # every kernel class, on an UNTIMED step: per-class breakdown + dominant class
        tw0 = time.perf_counter()
        step(t_cur)
        t_cur -= 1
        sync()
        warm_ms = (time.perf_counter() - tw0) * 1e3
        if last:
            prof_all = _lib.profile_end()
        log("warmup step done")
    sync()                                   # barrier + device sync on both sides of the timed region (also when --warmup 0)
    # Two HIP events per launch cost ~4 % of the step when every launch is bracketed (tools/profile_overhead.py), so the
    # timed steps instrument only the dominant kernel class (the roofline kernel); the other classes come from the warm-up pass.
    dom_name = max(prof_all.items(), key=lambda kv: kv[1]["total_ms"])[0] if prof_all else None
    _lib.profile_begin([dom_name] if dom_name else None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(t_cur)
        t_cur -= 1
    sync()
    elapsed = time.perf_counter() - t0
    log(f"timed steps done: {elapsed:.3f}s")
    prof = _lib.profile_end()
    log("profile read")
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    assert torch.isfinite(x).all(), "non-finite state after the timed steps"

    if rank == 0:
        sec_per_step = elapsed / args.steps
        value = world * B / (STEPS_PER_TRAJECTORY * sec_per_step)
        # dominant kernel class by total time
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"])
        name, d = dom
        if d["flops"] > 0:
            achieved = d["flops"] / (d["total_ms"] * 1e-3) / 1e12
            peak, peak_note = mfma_roof(name)
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                    "frac": achieved / peak, "traffic": pmc_traffic(name), "peak_note": peak_note}
        else:
            achieved = d["bytes"] / (d["total_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                    "traffic": None}
        roof["kernel"] = name
        roof["launches"] = d["launches"]
        roof["avg_launch_ms"] = d["total_ms"] / max(d["launches"], 1)
        if prof_all:
            roof["breakdown_ms_per_step"] = {k: round(v["total_ms"], 3) for k, v in sorted(prof_all.items())}
            roof["breakdown_note"] = ("all classes bracketed by events on the last (untimed) warm-up step; the timed steps bracket "
                                      "only the roofline kernel class")
            roof["kernel_time_fraction_of_step"] = sum(v["total_ms"] for v in prof_all.values()) / warm_ms   # of that warm-up step
        else:
            roof["breakdown_ms_per_step"] = {k: round(v["total_ms"] / args.steps, 3) for k, v in sorted(prof.items())}
            roof["kernel_time_fraction_of_step"] = sum(v["total_ms"] for v in prof.values()) / (elapsed * 1e3)
        unit_gflop = 1794.5        # algorithmic GFLOP per trajectory-step (both U-Nets)
        roof["step_flops_fraction_of_fp32_peak"] = (B * unit_gflop / 1e3 / sec_per_step) / PEAK_FP32_MFMA_TFLOPS
        out = {
            "metric": "guided trajectories/sec, 2D smoke 64x64x32 @1000 DDPM steps",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": sec_per_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 tensors in HBM and fp32 accumulation everywhere. 3x3x3 convs and the implicit-GEMM ops split each "
                          "operand into 2 fp16 terms (22 significant bits) and sum 3 partial products per product (f16x3); the "
                          "stem and the fused attention blocks at C = 64 use the same f16x3 scheme; the C = 128 temporal "
                          "attention uses the exact 3-way bf16 split with 6 partial products (bf16x6), the C = 128 linear attention "
                          "the native fp32 MFMA. "
                          "Measured U-Net forward deviation from the reference's fp32 CPU output: 2.3e-6 (f16x3) vs 3.1e-6 "
                          "(bf16x6) vs 2.7e-6 (native fp32 MFMA) of the output range (tools/mode_error.py); tolerance 1e-4. "
                          "DPC_CONV_MODE / DPC_IGEMM_MODE = x6 | f32 select the other kernels",
            "config": {"workload": "S64 (BASELINE.json configs[2]): 2D smoke 64x64 x 32 frames, 1000-step guided DDPM, "
                                   f"batch={B} per GPU; one step = joint+prior Unet3D(dim 64, mults 1-2-4) forward + "
                                   "fused guidance/posterior update; trajectories/s = batch/(1000*s_per_step)",
                       "global_batch": world * B, "micro_batch": args.micro_batch, "parallelism": f"batch-shard x{world}"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(sd_cpu, init_cpu[:1])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
