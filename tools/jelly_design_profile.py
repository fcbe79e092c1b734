import sys, os, torch, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/inference'); sys.path.insert(0, '/root/repo/tests')
import inference_2d_jellyfish as J
from torch_force_fn import force_fn
args = J.build_parser().parse_args(["--synthetic", "True", "--batch_size", "16", "--timesteps", "8", "--sampling_timesteps", "8",
                                    "--inference_result_path", "/tmp/jelly_out"])
args.device = torch.device("cuda", 0); torch.cuda.set_device(0); torch.manual_seed(0)
J.load_normalization(args)
force_model, diffusion, bd_updater, design_fn = J.load_model(args)
B, Fr, s = 16, 20, 64
x = torch.randn(B, Fr, 4, s, s, device=args.device)
bd_0 = torch.rand(B, Fr, 3, s, s, device=args.device)
for _ in range(2):
    g = design_fn(x.clone(), bd_0)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    g = design_fn(x.clone(), bd_0); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=40, max_shapes_column_width=70))
t0 = time.perf_counter()
for _ in range(3): g = design_fn(x.clone(), bd_0)
torch.cuda.synchronize(); print("design_fn ms", (time.perf_counter() - t0) / 3 * 1e3)
