"""GPU parity of the training step: TrainableUnet3D (hand-written HIP forward-with-tape + backward through
the C ABI) and the fused optimizer against the REFERENCE's own records -- loss, every parameter gradient, the clip norm and
the post-Adam weights of tests/golden/train_*.npz (tools/gen_golden_train.py: diffusion_2d_smoke.py p_losses :809-831,
Trainer.train :998-1054) -- and against torch autograd through the CPU oracle at the real width."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    return torch.device("cuda:0")


def _net(g, dev, bwd_mode, **kw):
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffphycon_amd.model.video_diffusion_pytorch.unet3d_train import TrainableUnet3D
    m = Unet3D_with_Conv3D(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=int(g["channels"]))
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w0:")})
    return TrainableUnet3D(m, dev, bwd_mode=bwd_mode, **kw)


def _sched(dev):
    from oracle import train_smoke as T
    s = T.schedule(1000)
    return s["sqrt_alphas_cumprod"].to(dev).contiguous(), s["sqrt_one_minus_alphas_cumprod"].to(dev).contiguous()


def _check_grads(T, g, step, tol, scale=1.0):
    names = [k[len(f"s{step}:g:"):] for k in g.files if k.startswith(f"s{step}:g:")]
    assert sorted(names) == sorted(T.names)
    G = max(float(np.abs(g[f"s{step}:g:{k}"]).max()) for k in names)
    bad = []
    for k in names:
        ref = torch.from_numpy(g[f"s{step}:g:{k}"])
        got = (T.ctx.G[k].cpu() / scale).reshape(ref.shape)
        err = (got - ref).abs().max().item()
        # (tensors whose true gradient is zero -- conv biases in front of a one-channel-per-group GroupNorm -- hold rounding
        #  noise on both sides: the floor is relative to the largest gradient of the net)
        if not err < tol * ref.abs().max().item() + 2e-6 * G:
            bad.append((k, err, ref.abs().max().item()))
    assert not bad, bad[:8]


@pytest.mark.parametrize("bwd_mode", ["x6", "f32"])
@pytest.mark.parametrize("tag", ["joint", "w", "wide"])
def test_loss_and_every_gradient_match_the_reference(tag, bwd_mode, dev):
    g = load_golden(f"train_{tag}")
    T = _net(g, dev, bwd_mode)
    a, b = _sched(dev)
    coff = 3 if int(g["channels"]) == 2 else 0
    x0 = torch.from_numpy(g["s0:state"]).to(dev)
    loss = T.p_losses(x0, torch.from_numpy(g["s0:t"]).to(dev), torch.from_numpy(g["s0:noise"]).to(dev), a, b, channel_offset=coff)
    assert abs(loss.item() - float(g["s0:loss"])) < 1e-5 * float(g["s0:loss"])
    _check_grads(T, g, 0, 1e-4)
