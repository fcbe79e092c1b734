"""Which operator of a U-Net forward changes its result when ANOTHER kernel stream runs on the GPU at the same time?
Two Unet3D handles (as tests/test_gpu_unet3d.py::test_two_denoisers_on_two_streams_equal_the_serial_forwards): the prior net runs on a
side stream while the joint net runs on the main stream; every debug tap of the prior net is compared with its serial run and the FIRST
tap that differs is reported with the number / positions of the differing elements.
    gpurun -- 'python tools/two_stream_bisect.py [rounds] [dim] [frames] [hw]'      (A/B switches: DPC_DEBUG=1 DPC_...=.. in the environment)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet3d as O  # noqa: E402  (tool: seeded synthetic weights + the tap names)
from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D  # noqa: E402

TAPS = os.environ.get("BISECT_TAPS", "0") == "1"        # per-tap comparison perturbs the overlap (extra copy kernels): off by default
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 32
Fr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
HW = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0")
nets, cfgs = [], []
for ch, seed in ((6, 0), (2, 1)):
    cfg = O.Unet3DConfig(dim=dim, dim_mults=(1, 2), channels=ch)
    m = Unet3D_with_Conv3D(dim=dim, dim_mults=(1, 2), channels=ch)
    m.load_state_dict(O.synthetic_state_dict(cfg, seed=seed))
    nets.append(m.to(dev))
    cfgs.append(cfg)
torch.manual_seed(0)
x = torch.randn(4, Fr, 6, HW, HW, device=dev)
tj = torch.tensor([900, 10, 500, 3], device=dev)
tw = torch.tensor([1, 999, 42, 700], device=dev)
# tap names and shapes from one oracle forward of the prior net on a tiny input of the same geometry
taps_ref = {}
with torch.no_grad():
    O.unet3d_forward(O.synthetic_state_dict(cfgs[1], seed=1), cfgs[1], x[:, :, 3:5].cpu(), tw.cpu(), taps=taps_ref)
shapes = {k: tuple(v.shape) for k, v in taps_ref.items()}
if TAPS:
    nets[1].debug_taps(True)


def grab():
    out = {}
    if not TAPS:
        return out
    for k, shp in shapes.items():
        try:
            out[k] = nets[1].get_tap(k, shp, dev).clone()
        except RuntimeError:
            pass
    return out


yw0 = nets[1](x[:, :, 3:5], tw).clone()
serial = grab()
yw1 = nets[1](x[:, :, 3:5], tw).clone()
print("serial repeat equal:", torch.equal(yw0, yw1), flush=True)
nets[0](x, tj)
torch.cuda.synchronize()
side = torch.cuda.Stream()
bad_rounds = 0
for r in range(rounds):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        yw = nets[1](x[:, :, 3:5], tw)
    nets[0](x, tj)
    cur.wait_stream(side)
    torch.cuda.synchronize()
    now = grab()
    first = None
    for k in shapes:                       # (insertion order = forward order)
        if k in now and not torch.equal(now[k], serial[k]):
            first = k
            break
    if first is None and torch.equal(yw, yw0):
        continue
    bad_rounds += 1
    if first is None:
        d = (yw - yw0).abs()
        nz = (d > 0).nonzero()
        per_b = [int((d[b] > 0).sum()) for b in range(d.shape[0])]
        per_f = [int((d[:, f] > 0).sum()) for f in range(d.shape[1])]
        print(f"round {r}: output differs in {nz.shape[0]} of {d.numel()} elements, max |diff| {d.max().item():.3e} (range {yw0.abs().max().item():.3e}); "
              f"per trajectory {per_b}; per frame {per_f}; first {nz[:4].tolist()}", flush=True)
        continue
    d = (now[first] - serial[first]).abs()
    nz = (d > 0).nonzero()
    print(f"round {r}: first differing tap {first} shape {shapes[first]}: {nz.shape[0]} elements, max |diff| {d.max().item():.3e} "
          f"(range {serial[first].abs().max().item():.3e}); first positions {nz[:6].tolist()}; last {nz[-3:].tolist()}", flush=True)
print(f"{bad_rounds} of {rounds} concurrent rounds differ from the serial forward")
