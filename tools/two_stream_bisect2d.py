"""The Unet2D (Burgers) form of tools/two_stream_bisect.py: two small Burgers denoisers (the widths tools/rank_stress.py's entry-script
runs use: dim 16, mults (1,2,4) and (1,2), batch 3, 16 x 128 images) run concurrently on two streams of one process; the second net's
output is compared with its serial result.    gpurun -- 'python tools/two_stream_bisect2d.py [rounds] [dim] [batch]'"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet2d as U  # noqa: E402
from diffphycon_amd.model.burgers_1d.unet import Unet2D  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
nets = []
for mults, seed in (((1, 2, 4), 0), ((1, 2), 1)):
    cfg = U.Unet2DConfig(dim=dim, dim_mults=mults, resnet_block_groups=1)
    m = Unet2D(dim=dim, dim_mults=mults, channels=2, out_dim=2, resnet_block_groups=1)
    m.load_state_dict(U.synthetic_state_dict(cfg, seed=seed))
    nets.append(m.to(dev))
torch.manual_seed(0)
x = torch.randn(B, 2, 16, 128, device=dev)
ta = torch.randint(0, 1000, (B,), device=dev)
tb = torch.randint(0, 1000, (B,), device=dev)
TAPS = os.environ.get("BISECT_TAPS", "0") == "1"
shapes = {}
if TAPS:
    cfg1 = U.Unet2DConfig(dim=dim, dim_mults=(1, 2), resnet_block_groups=1)
    tr = {}
    with torch.no_grad():
        U.unet2d_forward(U.synthetic_state_dict(cfg1, seed=1), cfg1, x.cpu(), tb.cpu(), taps=tr)
    shapes = {k: tuple(v.shape) for k, v in tr.items()}
    nets[1].debug_taps(True)


def grab():
    out = {}
    for k, shp in shapes.items():
        try:
            out[k] = nets[1].get_tap(k, shp, dev).clone()
        except RuntimeError:
            pass
    return out


y0 = [nets[0](x, ta).clone(), nets[1](x, tb).clone()]
serial = grab()
rep = [torch.equal(nets[0](x, ta), y0[0]), torch.equal(nets[1](x, tb), y0[1])]
print("serial repeats equal:", rep, flush=True)
torch.cuda.synchronize()
side = torch.cuda.Stream()
bad = [0, 0]
for r in range(rounds):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        yb = nets[1](x, tb)
    ya = nets[0](x, ta)
    cur.wait_stream(side)
    torch.cuda.synchronize()
    if TAPS:
        now = grab()
        for k in shapes:
            if k in now and not torch.equal(now[k], serial[k]):
                d = (now[k] - serial[k]).abs()
                print(f"round {r}: first differing tap of net 1: {k} shape {shapes[k]}: {int((d > 0).sum())} elements, max |diff| {d.max().item():.3e}", flush=True)
                break
    for i, y in enumerate((ya, yb)):
        if not torch.equal(y, y0[i]):
            bad[i] += 1
            d = (y - y0[i]).abs()
            print(f"round {r}: net {i} differs in {int((d > 0).sum())} of {d.numel()} elements, max |diff| {d.max().item():.3e} (range "
                  f"{y0[i].abs().max().item():.3e}); per trajectory {[int((d[b] > 0).sum()) for b in range(B)]}", flush=True)
print(f"net 0 (main stream): {bad[0]} of {rounds} rounds differ; net 1 (side stream): {bad[1]} of {rounds}")
