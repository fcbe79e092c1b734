import sys, torch
sys.path.insert(0, '/root/repo')
from oracle import unet3d as O
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
dev = torch.device('cuda:0')
cfg = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6)
sd = O.synthetic_state_dict(cfg, seed=11)
gen = torch.Generator().manual_seed(11)
x = torch.randn(4, 32, 6, 64, 64, generator=gen).to(dev)
t = torch.tensor([999, 500, 10, 0]).to(dev)
ms = {}
for mb in (4, 1):
    m = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6, micro_batch=mb)
    m.load_state_dict(sd); m = m.to(dev); m.debug_taps(True)
    y = m(x, t); ms[mb] = (m, y)
print("out equal:", torch.equal(ms[4][1], ms[1][1]), (ms[4][1] - ms[1][1]).abs().max().item())
names = ["init_conv", "init_temporal_attn"] + [f"downs.{i}.{j}" for i in range(3) for j in range(5)] + ["mid_block1", "mid_spatial_attn", "mid_temporal_attn", "mid_block2"] + [f"ups.{i}.{j}" for i in range(2) for j in (0, 2, 3, 4)] + ["final_conv.0"]
dims = {"init_conv": (64, 64), "init_temporal_attn": (64, 64)}
def shape_of(name):
    if name.startswith("init") or name == "final_conv.0": return 64, 64
    if name.startswith("downs.0"): return (64, 32) if name.endswith(".4") else (64, 64)
    if name.startswith("downs.1"): return (128, 16) if name.endswith(".4") else (128, 32)
    if name.startswith("downs.2"): return (256, 16)
    if name.startswith("mid"): return 256, 16
    if name.startswith("ups.0"): return (128, 32) if name.endswith(".4") else (128, 16)
    if name.startswith("ups.1"): return (64, 64) if name.endswith(".4") else (64, 32)
for n in names:
    try:
        C, S = shape_of(n)
        a = ms[4][0].get_tap(n, (4, C, 32, S, S), dev)[3]
        b = ms[1][0].get_tap(n, (1, C, 32, S, S), dev)[0]
        print(f"{n:22s} equal={torch.equal(a, b)} maxdiff={(a - b).abs().max().item():.3e}")
    except Exception as e:
        print(n, "ERR", str(e)[:80])
