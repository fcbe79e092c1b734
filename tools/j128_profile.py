"""Where a guided J128 step goes (BASELINE.json configs[3]: 128 x 128 x 20 frames, batch 16): per-class event times of ONE design-gradient
call (forward + backward of both 2-D surrogates, model/surrogates_hip.py) and of ONE pair of denoiser forwards, on the GPU box:
    gpurun -- 'python tools/j128_profile.py [image_size] [batch]'"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "inference"))
import inference_2d_jellyfish as J  # noqa: E402
from diffphycon_amd import _lib  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
a = J.build_parser().parse_args(["--synthetic", "True", "--batch_size", str(B), "--num_batches", "1", "--timesteps", "4", "--sampling_timesteps", "4",
                                 "--image_size", str(size), "--frames", "20", "--surrogate_dim", "64", "--inference_result_path", "/tmp/j128_prof"])
a.device = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.manual_seed(0)
J.load_normalization(a)
force_model, diffusion, bd_updater, design_fn = J.load_model(a)
x = torch.randn(B, 20, 4, size, size, device=a.device)
bd = torch.rand(B, 20, 3, size, size, device=a.device)


def timed(fn, name):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    _lib.profile_begin()
    fn()
    torch.cuda.synchronize()
    prof = _lib.profile_end()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"== {name}: {ms:.1f} ms per call; classes of one call (events, every launch bracketed):")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]):
        if v["total_ms"] > 0.3:
            tf = v["flops"] / (v["total_ms"] * 1e-3) / 1e12 if v["flops"] else 0.0
            print(f"   {k:28s} {v['total_ms']:8.2f} ms  x{v['launches']:4d}  {tf:7.1f} TF/s")


timed(lambda: design_fn(x.clone(), bd), "design gradient (both surrogates, forward + backward)")
t = torch.full((B,), 500, device=a.device, dtype=torch.long)
xs = torch.randn(B, 20, 7, size, size, device=a.device)
cond = torch.randn(B, 20, 3, size, size, device=a.device)
timed(lambda: diffusion._denoise(xs, cond, t), "the two denoisers (7 -> 4 state net, 7 -> 1 angle net)")
