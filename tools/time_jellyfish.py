#!/usr/bin/env python
"""Per-step time of the jellyfish control sampler (inference/inference_2d_jellyfish.py --synthetic) on one GPU: runs the
script's own pipeline at two chain lengths and differences out the fixed costs.
    python tools/time_jellyfish.py [batch]      -> ms per guided DDPM step, trajectories/s at 1000 steps"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "inference"))
import inference_2d_jellyfish as J  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if os.environ.get("JELLY_CUDNN_BENCHMARK"):
    torch.backends.cudnn.benchmark = True
times = {}
for T in (8, 8, 40):
    args = J.build_parser().parse_args(["--synthetic", "True", "--batch_size", str(B), "--num_batches", "1", "--timesteps", str(T),
                                        "--sampling_timesteps", str(T), "--inference_result_path", "/tmp/jelly_out"])
    args.device = torch.device("cuda", 0)
    torch.cuda.set_device(args.device)
    torch.manual_seed(0)
    J.load_normalization(args)
    force_model, diffusion, bd_updater, design_fn = J.load_model(args)
    ppl = J.InferencePipeline(diffusion, {"design_fn": design_fn, "design_guidance": args.design_guidance, "bd_updater": bd_updater},
                              results_path=args.inference_result_path, args_general=args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ppl.run(J.synthetic_batches(args))
    torch.cuda.synchronize()
    times[T] = time.perf_counter() - t0
ms = (times[40] - times[8]) / 32 * 1e3
print(f"jellyfish {args.image_size}x{args.image_size} x {args.frames} frames, batch {B}: {ms:.1f} ms per guided DDPM step "
      f"-> {B / ms:.4f} trajectories/s at 1000 steps")
