import sys, ctypes as C, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from diffphycon_amd import _lib as L
dev = torch.device('cuda:0')
def run(B, Fr, H, W, Ci, Co):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Ci, Fr, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, 1, generator=g) / Ci ** 0.5
    b = torch.randn(Co, generator=g)
    ref = F.conv3d(x.double(), w.double(), b.double()).float()
    xd = x.permute(0, 2, 3, 4, 1).contiguous().to(dev); wd = w.to(dev).contiguous(); bd = b.to(dev)
    out = torch.empty(B, Fr, H, W, Co, device=dev)
    ws = L.workspace(L.lib().dpc_conv_workspace_bytes(Ci, Co, 1), dev)
    L.check(L.lib().dpc_conv3d_cl(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), B, Fr, H, W, Ci, Co, 1, 1, 1, 1, 1, 1, 0, 0, 0,
                                  C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    got = out.cpu().permute(0, 4, 1, 2, 3)
    err = (got - ref).abs()
    print(f"M={B*Fr*H*W} {Ci}->{Co}: max err {err.max().item():.3e}  bad rows frac {(err.amax(1) > 1e-3).float().mean().item():.4f}")
    return err
for (B, Fr, H, W, Ci, Co) in [(1, 32, 16, 16, 256, 384), (4, 32, 16, 16, 256, 384), (4, 32, 16, 16, 256, 64), (4, 32, 16, 16, 256, 128), (8, 32, 32, 32, 128, 64), (2, 32, 16, 16, 128, 384)]:
    e = run(B, Fr, H, W, Ci, Co)
    if e.max() > 1e-3:
        bad = (e > 1e-3).nonzero()
        print("  first bad idx (b,c,f,h,w):", bad[0].tolist(), " last:", bad[-1].tolist(), " count", len(bad))
        print("  bad channels:", sorted(set(bad[:, 1].tolist()))[:20], " bad b:", sorted(set(bad[:, 0].tolist())))
