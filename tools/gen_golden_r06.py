"""Fixtures of r06 (build container only: imports the reference through tools/refshim.py).

  unet3d_dim64.npz   : the REFERENCE's Unet3D_with_Conv3D(dim=64, dim_mults=(1,2,4), channels=6) -- the width of every BASELINE smoke /
                       jellyfish config -- on a 2 x 8 x 16 x 16 input: output + six taps (a 64-wide ResnetBlock with 8-channels-per-
                       group GroupNorm, the 128-wide spatial linear attention, the 256-wide mid spatial / temporal attention, a
                       concatenated up block, the final block).  Weights are NOT stored: they are oracle.unet3d.synthetic_state_dict
                       (seed in the file; NumPy PCG64, identical on any box) loaded strictly into the reference module.
  unet2d_popc.npz    : the reference's Unet2D at the POPC joint width (dim 64, mults (1,2,4,8,16), one GroupNorm group) on a
                       2 x 2 x 16 x 128 input: output + the init / mid / final taps; weights = oracle.unet2d.synthetic_state_dict(seed).
  ckpt_<task>/...pt  : checkpoint files in the layout the reference's Trainer.save writes (diffusion_2d_smoke.py:942-955,
                       diffusion_1d_burgers.py:934-949, diffusion_2d_jellyfish.py Trainer.save): {'step', 'model', 'opt', 'ema', 'scaler'} with
                       'model' = the REFERENCE GaussianDiffusion.state_dict() of tiny nets (all schedule buffers + both denoisers, the
                       rotary `freqs` buffers included), 'opt' = a real torch.optim.Adam.state_dict() over the reference's parameters
                       after one step, 'ema' = the state_dict of an EMA wrapper in ema_pytorch's published layout (`initted`, `step`,
                       `online_model.*`, `ema_model.*`; the package itself is not available offline -- layout unpinned, values = the model's).
                       tests/test_checkpoint_layout.py loads them through the readers the inference scripts use.

    python tools/gen_golden_r06.py [unet3d] [unet2d] [ckpt]
"""
import copy
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refshim  # noqa: E402

refshim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

from gen_golden import save  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def _hooked(m, names):
    taps, hooks = {}, []
    named = dict(m.named_modules())
    for name in names:
        hooks.append(named[name].register_forward_hook(lambda _m, _i, o, name=name: taps.__setitem__("tap:" + name, o.detach().clone())))
    return taps, hooks


def gen_unet3d():
    from model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from oracle import unet3d as O
    seed = 60
    cfg = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6)
    sd = O.synthetic_state_dict(cfg, seed=seed)
    m = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6).eval()
    own = m.state_dict()
    missing = [k for k in own if k not in sd and not k.endswith("rotary_emb.freqs")]
    extra = [k for k in sd if k not in own]
    assert not missing and not extra, (missing, extra)
    m.load_state_dict(sd, strict=False)                     # (only the rotary `freqs` buffers keep their constructor values)
    x = torch.randn(2, 8, 6, 16, 16, generator=torch.Generator().manual_seed(seed))
    t = torch.tensor([999, 3])
    taps, hooks = _hooked(m, ("downs.0.0", "downs.1.2", "mid_spatial_attn", "mid_temporal_attn", "ups.0.0", "ups.2.1"))
    with torch.no_grad():
        y = m(x, t)
    for h in hooks:
        h.remove()
    save("unet3d_dim64", x=x, t=t, y=y, seed=seed, dim=64, dim_mults=np.array((1, 2, 4)), channels=6, **taps)


def gen_unet2d():
    from model.burgers_1d.unet import Unet2D
    from oracle import unet2d as U
    seed = 61
    mults = (1, 2, 4, 8, 16)
    cfg = U.Unet2DConfig(dim=64, dim_mults=mults, resnet_block_groups=1)
    sd = U.synthetic_state_dict(cfg, seed=seed)
    m = Unet2D(dim=64, init_dim=None, out_dim=2, dim_mults=mults, channels=2, resnet_block_groups=1).eval()
    m.load_state_dict(sd)
    x = torch.randn(2, 2, 16, 128, generator=torch.Generator().manual_seed(seed))
    t = torch.tensor([999, 0])
    taps, hooks = _hooked(m, ("init_conv", "mid_block2", "final_res_block"))
    with torch.no_grad():
        y = m(x, t)
    for h in hooks:
        h.remove()
    save("unet2d_popc", x=x, t=t, y=y, seed=seed, dim=64, dim_mults=np.array(mults), groups=1, **taps)


class _EMA(nn.Module):
    """ema_pytorch.EMA's state_dict layout as published (0.2.x): buffers `initted`, `step`, sub-modules `online_model`, `ema_model`"""

    def __init__(self, model):
        super().__init__()
        self.online_model = model
        self.ema_model = copy.deepcopy(model)
        self.register_buffer("initted", torch.tensor(True))
        self.register_buffer("step", torch.tensor(10))


def _checkpoint(diffusion, step):
    """what Trainer.save stores (diffusion_2d_smoke.py:945-953), with one real Adam step behind `opt`"""
    opt = torch.optim.Adam(diffusion.parameters(), lr=1e-4, betas=(0.9, 0.99))
    # gradients as a backward pass leaves them: every TRAINABLE parameter has one, the rotary table (an nn.Parameter with
    # requires_grad False in rotary-embedding-torch, shared by all temporal attention blocks) has none -> no Adam state at its index.
    # Value = 1e-3 x (1 + position in parameters()): the first moment then names the position it belongs to.
    for i, p in enumerate(diffusion.parameters()):
        if p.requires_grad:
            p.grad = torch.full_like(p, 1e-3 * (1 + i))
    opt.step()
    opt.zero_grad(set_to_none=True)
    return {"step": step, "model": diffusion.state_dict(), "opt": opt.state_dict(), "ema": _EMA(diffusion).state_dict(), "scaler": None,
            # (not part of the reference's file: the names behind Adam's indices, for the test that pins the index space)
            "_param_names": [k for k, _ in diffusion.named_parameters()]}


def gen_ckpt():
    from model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    # ---- smoke (diffusion_2d_smoke.py: one denoiser per trainer -- smoke_train_joint.sh / smoke_train_w.sh)
    from diffusion.diffusion_2d_smoke import GaussianDiffusion as GDS
    torch.manual_seed(70)
    m = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2), channels=6)
    gd = GDS(m, image_size=16, frames=4, timesteps=1000, sampling_timesteps=100, loss_type="l2", objective="pred_noise")
    d = os.path.join(GOLDEN, "ckpt_smoke")
    os.makedirs(d, exist_ok=True)
    torch.save(_checkpoint(gd, 7), os.path.join(d, "model-1.pt"))
    # ---- Burgers (diffusion_1d_burgers.py:934-949: cos10000-model-<milestone>.pt, + 'version')
    from model.burgers_1d.unet import Unet2D
    from diffusion.diffusion_1d_burgers import GaussianDiffusion as GDB
    torch.manual_seed(71)
    mb = Unet2D(dim=8, init_dim=None, out_dim=2, dim_mults=(1, 2), channels=2, resnet_block_groups=1)
    gb = GDB(mb, seq_length=(16, 32), timesteps=1000, auto_normalize=False, use_conv2d=True, temporal=True)
    d = os.path.join(GOLDEN, "ckpt_burgers")
    os.makedirs(d, exist_ok=True)
    ck = _checkpoint(gb, 11)
    ck["version"] = "1.0"
    torch.save(ck, os.path.join(d, "cos10000-model-1.pt"))
    # ---- jellyfish (diffusion_2d_jellyfish.py: one 7-channel denoiser per trainer, out_dim 4 (joint) here)
    import diffusion.diffusion_2d_jellyfish as DJ
    torch.manual_seed(72)
    mj = Unet3D_with_Conv3D(dim=8, out_dim=4, dim_mults=(1, 2), channels=7)
    gj = DJ.GaussianDiffusion(mj, image_size=16, frames=4, cond_steps=1, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                              objective="pred_noise", device="cpu")
    d = os.path.join(GOLDEN, "ckpt_jellyfish")
    os.makedirs(d, exist_ok=True)
    torch.save(_checkpoint(gj, 13), os.path.join(d, "model-1.pt"))
    for sub in ("ckpt_smoke/model-1.pt", "ckpt_burgers/cos10000-model-1.pt", "ckpt_jellyfish/model-1.pt"):
        print("wrote", sub, f"{os.path.getsize(os.path.join(GOLDEN, sub)) / 1024:.0f} KiB")


if __name__ == "__main__":
    todo = sys.argv[1:] or ["unet3d", "unet2d", "ckpt"]
    for s in todo:
        {"unet3d": gen_unet3d, "unet2d": gen_unet2d, "ckpt": gen_ckpt}[s]()
