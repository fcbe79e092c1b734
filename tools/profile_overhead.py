import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from diffphycon_amd import _lib
from diffphycon_amd.diffusion.diffusion_2d_smoke import SmokeGuidance
dev = torch.device('cuda:0')
gd, _ = bench.build_models(dev, 8)
guide = SmokeGuidance((2.0, 18.0, 20.0, 16.0, 20.0, 1.0), 0.0)
B = 64
gd.noise_seed, gd.traj_offset = 0, 0
init = bench.synthetic_init(B, 0).to(dev)
x = gd.sample_noise([B, bench.FRAMES, 6, bench.SIZE, bench.SIZE], dev)
x[:, 0, 0] = init
def step(t): gd.p_sample(None, x, t, design_fn=guide, design_guidance="standard", init=init)
step(999); torch.cuda.synchronize()
for prof in (False, True, False, True):
    if prof: _lib.profile_begin()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(3): step(998 - i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    if prof: _lib.profile_end()
    print("profiling", prof, "ms/step %.2f" % (dt * 1e3))
