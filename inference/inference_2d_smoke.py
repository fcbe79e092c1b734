"""2-D smoke control inference with the reference's entry surface (inference/inference_2d_smoke.py): same flags,
same `load_ddpm_model / load_model / InferencePipeline.{run_model, multi_evaluate, run} / inference / load_data / main`
structure, running on libdpc (HIP) through the `diffphycon_amd` mirrors.

Differences from the reference, all host-side:
  * `guidance_fn` is the closed-form `SmokeGuidance` (same gradient; evaluated inside the fused update kernel);
  * `multi_evaluate` runs all trajectories of a batch in ONE GPU launch (`solver_batch`) instead of one CPU process per
    trajectory, and keeps only the sub-sampled frames the metrics read (:388-390);
  * `--synthetic True` (extra flag) fabricates the test split and random-initialises the two U-Nets when no dataset /
    checkpoint is mounted; one process per GPU shards the batches (torchrun), metrics are gathered with RCCL.
"""
import argparse
import datetime
import os
import sys
import time

import numpy as np
import torch

sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffphycon_amd.dataset.data_2d import Smoke, SyntheticSmoke  # noqa: E402
from diffphycon_amd.dataset.apps.evaluate_solver import init_sim_128, init_velocity_, solver_batch  # noqa: E402
from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion, SmokeGuidance, Trainer  # noqa: E402
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D  # noqa: E402
from diffphycon_amd import _lib, parallel  # noqa: E402
from filepath import SMOKE_DATA_PATH, SMOKE_RESULTS_PATH  # noqa: E402


def guidance_fn(x, args, RESCALER, w_energy=0, w_init=0, low=None, init=None, init_u=None):
    """dJ/d(x*RESCALER) of the reference's objective (:30-44), closed form."""
    return SmokeGuidance(RESCALER.reshape(-1), w_energy, w_init)(x)


def _ddpm(model, args, eval_2ddpm=False, **kw):
    return GaussianDiffusion(
        model, image_size=args.image_size, frames=32, timesteps=1000,
        sampling_timesteps=args.ddim_sampling_steps if args.using_ddim else 1000, ddim_sampling_eta=args.ddim_eta,
        loss_type="l2", objective="pred_noise", standard_fixed_ratio=args.standard_fixed_ratio,
        coeff_ratio=args.coeff_ratio, eval_2ddpm=eval_2ddpm, **kw)


def load_ddpm_model(args, RESCALER):
    model_joint = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6).to(args.device)
    diffusion_joint = _ddpm(model_joint, args)
    model_w = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=2).to(args.device)
    diffusion_w = _ddpm(model_w, args)
    if not args.synthetic:
        Trainer(diffusion_joint, dataset=args.dataset, dataset_path=args.dataset_path,
                results_path=args.diffusion_model_joint_path, amp=False).load(args.diffusion_joint_checkpoint)
        Trainer(diffusion_w, dataset=args.dataset, dataset_path=args.dataset_path,
                results_path=args.diffusion_model_w_path, amp=False).load(args.diffusion_w_checkpoint)
    diffusion = _ddpm([diffusion_joint.model, diffusion_w.model], args, eval_2ddpm=True, w_prob_exp=args.w_prob_exp,
                      device=args.device)
    return diffusion, args.device


def load_model(args, RESCALER, w_energy=0, w_init=0):
    if args.inference_method != "DDPM":
        raise NotImplementedError("only --inference_method DDPM is implemented by the reference script as well")
    diffusion, device = load_ddpm_model(args, RESCALER)
    design_fn = SmokeGuidance(RESCALER.reshape(-1).float(), w_energy, w_init)
    return [diffusion], design_fn


class InferencePipeline(object):
    def __init__(self, model, args=None, RESCALER=1, results_path=None, args_general=None):
        self.model, self.args, self.results_path, self.args_general = model, args, results_path, args_general
        self.image_size, self.device, self.upsample = args_general.image_size, args_general.device, args_general.upsample
        self.RESCALER = RESCALER
        self.sim = init_sim_128()
        os.makedirs(self.results_path, exist_ok=True)

    def prepare_inputs(self, state):
        """The sampler's conditioning of a batch (:179-186) as device tensors: torch element-wise kernels on the sampling stream.  The
        overlapped schedule calls this for batch i + 1 BEFORE batch i's rollouts are enqueued on the side stream, so that none of
        these kernels runs beside a rollout (torch's kernels contain packed fp32 instructions)."""
        state = state[:, ::8].to(self.args_general.device)          # (slice on the host: 0.2 GB over PCIe instead of 1.6 GB; same values)
        return dict(batch_size=state.shape[0], init=(state[:, 0, 0] / self.RESCALER[:, 0, 0]).contiguous(), init_u=state[:, 0, 0].contiguous(),
                    control=(state[:, :, 3:5] / self.RESCALER[:, :, 3:5]).contiguous())

    def run_model(self, state, prepared=None):
        """state: not rescaled [B, 256, 6, 64, 64] -> sampled + rescaled [B, 32, 6, 64, 64]   (:179-197)"""
        kw = prepared if prepared is not None else self.prepare_inputs(state)
        output = self.model[0].sample(design_fn=self.args["design_fn"], design_guidance=self.args["design_guidance"], low=None, **kw)
        if getattr(self, "_side_stream", None) is not None:
            # overlapped schedule: the torch element-wise kernels below must not run beside the previous batch's rollouts (torch's kernels
            # contain packed fp32 instructions: DESIGN.md 6.2).  Normally the rollouts ended ~24 s ago; a very short chain waits here.
            torch.cuda.current_stream().wait_stream(self._side_stream)
        output = output * self.RESCALER
        output[:, :, -1] = output[:, :, -1].mean((-2, -1)).unsqueeze(-1).unsqueeze(-1).expand(-1, -1, 64, 64)
        return output

    def _rollouts_enqueue(self, pred, data):
        """First half of multi_evaluate's device work (:317-427) on the CURRENT stream, no host read: the sampled controls rolled through the
        PDE solver (csrc/smoke_rollout.hip).  Returns what _metric_rows needs."""
        k = int(data.shape[-1] / pred.shape[-1])
        d00 = data[:, 0, 0].to(pred.device)          # (only the initial density is read: 1 MB instead of the batch's 1.6 GB)
        pred[:, 0, 0] = d00[:, ::k, ::k]
        pred_ = pred.detach().clone()
        pred_[:, :, 3:5, 8:56, 8:56] = 0                                         # indirect control (:330)
        dens, _, vel, smoke = solver_batch(self.sim, init_velocity_(), d00, pred_[:, :, 3], pred_[:, :, 4],
                                           per_timelength=256, frame_stride=8, space_stride=2, want_zero_density=False)
        return pred, pred_, dens, vel, smoke

    def _metric_rows(self, pred, pred_, dens, vel, smoke):
        """Second half: the per-trajectory metric rows [B, 5] = (J_total, J_target, J_energy, mse, n_l2) -- torch element-wise kernels.  In
        the overlapped schedule this runs on the sampling stream AFTER the side stream has drained, with nothing else resident: torch's own
        kernels are built with packed fp32 instructions, which must not run beside another kernel."""
        B = pred.shape[0]
        cur = torch.empty(B, 32, 6, 64, 64, dtype=torch.float64, device=pred.device)     # data_current (:388-390)
        cur[:, :, 0] = dens
        cur[:, :, 1], cur[:, :, 2] = vel[..., 0], vel[..., 1]
        cur[:, :, 3], cur[:, :, 4] = pred_[:, :, 3].double(), pred_[:, :, 4].double()
        cur[:, :, 5] = smoke[:, :, None, None]
        mask = torch.ones_like(pred)
        mask[:, 0] = 0
        p, d = pred * mask, cur * mask
        diff = p - d
        # (diff[:, :, -1:] and not the reference's diff[:, :, [-1]]: a list index becomes an index tensor that is copied to the device with a
        #  BLOCKING copy on this stream -- the host then waits for whatever is queued in front of it; same elements, same order)
        mse = torch.cat((diff[:, :, :3], diff[:, :, -1:]), dim=2).square().mean((1, 2, 3, 4))
        n_l2 = diff[:, :, :3].square().sum((1, 2, 3, 4)).sqrt() / d[:, :, :3].square().sum((1, 2, 3, 4)).sqrt()
        J_target = -d[:, -1, -1, 0, 0]
        J_energy = d[:, :, 3:5].square().mean((1, 2, 3, 4))
        J_total = J_target + self.args_general.w_energy * J_energy
        return torch.stack((J_total, J_target, J_energy, mse, n_l2), dim=1)          # [B, 5] per-trajectory metric rows

    def _evaluate_enqueue(self, pred, data):
        return self._metric_rows(*self._rollouts_enqueue(pred, data))

    def _evaluate_report(self, rows, start, elapsed=None):
        """The host side of multi_evaluate: ONE read of the batch means, the reference's print lines.  `elapsed`: the evaluator's own
        time when it ran overlapped (device events around the rollouts + the metric rows), else wall time since `start` (:317, :425)."""
        self.last_rows = rows               # run() gathers them ONCE after its loop (ranks may own different batch counts)
        m = rows.mean(0).cpu().numpy()
        print(f"Time cost: {time.time() - start if elapsed is None else elapsed}")
        print("J_total=J_target+w*J_energy=", m[1], "+", self.args_general.w_energy, "*", m[2], "=", m[0])
        print("mse=", m[3], "normalized_l2=", m[4])
        return tuple(np.array([v]) for v in m)

    def multi_evaluate(self, pred, data, plot=False, method="DDPM"):
        """pred [B,32,6,64,64], data [B,256,6,64,64]: roll the sampled controls through the PDE solver and score
        (:317-427).  Returns per-batch means (J_total, J_target, J_energy, mse, n_l2) as arrays of length 1."""
        start = time.time()
        return self._evaluate_report(self._evaluate_enqueue(pred, data), start)

    def run(self, dataloader):
        """The reference's loop (:259-271) samples a batch, evaluates it, samples the next.  Here the evaluation of batch i -- 64 CUs
        for ~2 s: one persistent workgroup per rollout -- is enqueued on a SIDE stream and runs under batch i + 1's sampling (r05;
        `--overlap_evaluator False` restores the serial schedule).  Same kernels, same inputs: the metric rows are bit-identical to
        the serial run (tests/test_gpu_inference_scripts.py).  While a rollout is in flight the persistent sampling kernels are
        told to size their grids for the CUs that are left (include/dpc.h: dpc_set_cu_budget -- a PROCESS-WIDE library setting: it applies
        to every handle, thread and device of the process, which is one rank = one GPU = one pipeline here; reset in `finally`)."""
        J = {k: [] for k in ("J_total", "J_target", "J_energy", "mse", "n_l2")}
        rows = []
        overlap = bool(getattr(self.args_general, "overlap_evaluator", True)) and len(dataloader) > 1
        side = torch.cuda.Stream(device=self.device) if overlap else None
        self._side_stream = side
        budget = int(os.environ.get("DPC_EVALUATOR_CU_BUDGET", "192"))
        L = _lib.lib()
        pending = None                      # (rollout outputs on the side stream, start time, pred, the rollouts' completion event)
        tlog = os.environ.get("DPC_PIPELINE_LOG") == "1"
        t_run0 = time.time()

        def note(msg):
            if tlog:
                print(f"[pipeline +{time.time() - t_run0:8.3f} s] {msg}", file=sys.stderr, flush=True)

        def release_budget():
            # Called at the top of every sampling step while rollouts are in flight.  Grid sizes are fixed when a launch is ENQUEUED, and
            # the host runs ahead of the GPU: it first waits for the sampling stream (a ~0.1 ms bubble per step, for the ~9 steps the
            # rollouts last), then asks whether the rollouts are done and, if so, gives the persistent kernels the whole device back.
            torch.cuda.current_stream().synchronize()
            done_now = pending is not None and pending[3].query()
            note(f"sampling step begins; rollouts of the previous batch {'DONE' if done_now else 'still running'}")
            if done_now:
                L.dpc_set_cu_budget(0)
                self.model[0].step_callback = None

        def finish(pend):
            side.synchronize()               # the rollouts are done; the sampling stream is idle too (sample() ends with a host read)
            L.dpc_set_cu_budget(0)
            self.model[0].step_callback = None
            t_rows = time.time()
            rows_i = self._metric_rows(*pend[0])
            torch.cuda.current_stream().synchronize()
            # "Time cost" as the reference means it -- the evaluator's time -- not the ~25 s that passed since it was enqueued (the next
            # batch was sampled meanwhile): rollouts by device events on the side stream + the metric-row kernels and their host read
            out = self._evaluate_report(rows_i, pend[1], elapsed=pend[4].elapsed_time(pend[3]) * 1e-3 + (time.time() - t_rows))
            rows.append(self.last_rows)
            for key, v in zip(J, out):
                J[key].append(v)

        try:
            batches = iter(dataloader)
            nxt = next(batches, None)
            prepared = self.prepare_inputs(nxt[0]) if (overlap and nxt is not None) else None
            i = -1
            while nxt is not None:
                (state, sim_id), i = nxt, i + 1
                print(f"Batch No.{i}")
                note(f"batch {i}: loader returned")
                ids = [int(v) for v in sim_id]
                assert ids == list(range(ids[0], ids[0] + len(ids))), "batches must hold consecutive simulation ids"
                # noise keyed by the global simulation id (Philox): independent of batch size and of the sharding over ranks
                self.model[0].traj_offset, self.model[0].noise_epoch = ids[0], 0
                pred = self.run_model(state, prepared)
                note(f"batch {i}: run_model returned")
                print("pred shape: ", pred.shape)
                nxt = next(batches, None)
                if not overlap:
                    out = self.multi_evaluate(pred, state, plot=False, method=self.args_general.inference_method)
                    rows.append(self.last_rows)
                    for key, v in zip(J, out):
                        J[key].append(v)
                    continue
                if pending is not None:
                    finish(pending)              # batch i - 1's rollouts ran beside this batch's sampling
                # the NEXT batch's conditioning tensors are formed now, while nothing else is resident (see prepare_inputs)
                prepared = self.prepare_inputs(nxt[0]) if nxt is not None else None
                done = torch.cuda.Event()
                done.record()                    # pred (and the prepared inputs) are complete on the sampling stream
                start = time.time()
                with torch.cuda.stream(side):
                    side.wait_event(done)
                    began = torch.cuda.Event(enable_timing=True)
                    began.record()
                    r = self._rollouts_enqueue(pred, state)
                    rolled = torch.cuda.Event(enable_timing=True)
                    rolled.record()              # on the side stream: the rollouts are complete (the metric rows follow in finish())
                note(f"batch {i}: evaluator enqueued on the side stream")
                pending = (r, start, pred, rolled, began)
                if budget > 0:
                    L.dpc_set_cu_budget(budget)  # the next batch's persistent kernels leave the rollouts' CUs alone ...
                    self.model[0].step_callback = release_budget          # ... until the rollouts are done
            if pending is not None:
                finish(pending)
        finally:
            # (an exception inside the loop must not leave the library's process-wide CU budget lowered or the sampler's hook installed)
            L.dpc_set_cu_budget(0)
            self.model[0].step_callback = None
        if getattr(self.args_general, "world_size", 1) > 1:
            # one RCCL all_gather pair for the whole run (variable row counts per rank, global trajectory order); the summary
            # is then the mean over ALL trajectories (= the reference's mean of batch means when batches are equal-sized)
            dev = self.args_general.device
            local = torch.cat(rows) if rows else torch.zeros(0, 5, dtype=torch.float64, device=dev)
            allrows = parallel.gather_metric_rows(local).mean(0).cpu().numpy()
            J = {k: [np.array([v])] for k, v in zip(J, allrows)}
        self.all_rows = rows                 # per-batch [B, 5] metric rows of this rank, in batch order
        summary = ",\n".join(f"{k}: {np.stack(v).mean(0)}" for k, v in J.items())
        print("Final results!\nNumber of upsampling times: 0\n" + summary)
        with open(os.path.join(self.results_path, "results.txt"), "a") as f:
            f.write(datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S") + "\n" + str(self.args_general) + "\n")
            f.write("Number of upsampling times: 0\n" + summary + "\n" + "-" * 89 + "\n")
        return {k: np.stack(v).mean(0) for k, v in J.items()}


def inference(dataloader, diffusion, design_fn, args, RESCALER):
    ppl = InferencePipeline(diffusion, {"design_fn": design_fn, "design_guidance": args.design_guidance}, RESCALER,
                            results_path=args.inference_result_subpath, args_general=args)
    return ppl.run(dataloader)


def load_data(args):
    assert args.dataset == "Smoke"
    if args.synthetic:
        dataset = SyntheticSmoke(n_simu=args.n_test, size=args.image_size)
    else:
        dataset = Smoke(dataset_path=args.dataset_path, is_train=False)
    RESCALER = dataset.RESCALER.unsqueeze(0).to(args.device)
    # batch-shard the test split over ranks (one process per GPU); a rank's trajectories keep their global index
    start, stop = parallel.shard_range(len(dataset), args.rank, args.world_size)
    subset = torch.utils.data.Subset(dataset, range(start, stop))
    loader = torch.utils.data.DataLoader(subset, batch_size=args.batch_size, shuffle=False, pin_memory=True,
                                         num_workers=0 if args.synthetic else 8)
    print("number of batch in test_loader: ", len(loader))
    return loader, RESCALER


def main(args):
    dataloader, RESCALER = load_data(args)
    diffusion, design_fn = load_model(args, RESCALER, args.w_energy, w_init=args.w_init)
    return inference(dataloader, diffusion, design_fn, args, RESCALER)


def build_parser():
    parser = argparse.ArgumentParser(description="inference 2d inverse design model")
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--dataset", default="Smoke", type=str)
    parser.add_argument("--dataset_path", default=SMOKE_DATA_PATH, type=str)
    parser.add_argument("--w_energy", default=0, type=float)
    parser.add_argument("--image_size", type=int, default=64)
    parser.add_argument("--upsample", default=0, type=int)
    parser.add_argument("--is_condition_pad", default=True, type=eval)
    parser.add_argument("--is_condition_reward", default=False, type=eval)
    parser.add_argument("--batch_size", default=50, type=int)
    parser.add_argument("--inference_result_path", default=os.path.join(SMOKE_RESULTS_PATH, "inference_results"), type=str)
    parser.add_argument("--inference_method", default="DDPM", type=str)
    parser.add_argument("--diffusion_model_joint_path", default=os.path.join(SMOKE_RESULTS_PATH, "checkpoints/joint_models"), type=str)
    parser.add_argument("--diffusion_joint_checkpoint", default=50, type=int)
    parser.add_argument("--diffusion_model_w_path", default=os.path.join(SMOKE_RESULTS_PATH, "checkpoints/w_models"), type=str)
    parser.add_argument("--diffusion_w_checkpoint", default=17, type=int)
    parser.add_argument("--using_ddim", default=True, type=eval)
    parser.add_argument("--overlap_evaluator", default=True, type=eval,
                        help="(not in the reference) run batch i's PDE rollouts on a side stream under batch i + 1's sampling; results are bit-identical")
    parser.add_argument("--ddim_eta", default=1., type=float)
    parser.add_argument("--w_prob_exp", default=0.97, type=float)
    parser.add_argument("--ddim_sampling_steps", default=100, type=int)
    parser.add_argument("--design_guidance", default="standard", type=str)
    parser.add_argument("--standard_fixed_ratio", default=100000, type=float)
    parser.add_argument("--coeff_ratio", default=0, type=float)
    parser.add_argument("--w_init", default=0, type=float)
    # extra (not in the reference): run without datasets / checkpoints
    parser.add_argument("--synthetic", default=False, type=eval, help="synthetic test split + random-init U-Nets")
    parser.add_argument("--n_test", default=50, type=int, help="synthetic test-set size")
    return parser


if __name__ == "__main__":
    args = build_parser().parse_args()
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    assert torch.cuda.is_available(), "the HIP path needs a GPU"
    args.rank, args.world_size = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    args.device = torch.device("cuda", local)
    if args.world_size > 1:
        parallel.init_process_group(args.rank, args.world_size, args.device)
    args.inference_result_subpath = os.path.join(args.inference_result_path,
                                                 datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
    print("args: ", args)
    main(args)
