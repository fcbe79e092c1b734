"""Checkpoints in the layout the REFERENCE's Trainer.save writes (tools/gen_golden_r06.py builds them from the imported reference's own
GaussianDiffusion.state_dict() and a real torch.optim.Adam.state_dict(); diffusion_2d_smoke.py:942-955, diffusion_1d_burgers.py:934-949,
diffusion_2d_jellyfish.py Trainer.save) are read by the readers the inference scripts use -- `Trainer(...).load(milestone)` -- with every
key matched strictly, and the smoke trainer's index space for the optimizer state (position in diffusion_model.parameters()) is the
reference's.  CPU only: the readers need no GPU state.  (VERDICT r05 item 4b / SURVEY 8 row f-3.)"""
import os

import pytest
import torch

from conftest import GOLDEN

CK = {"smoke": ("ckpt_smoke", "model-1.pt"), "burgers": ("ckpt_burgers", "cos10000-model-1.pt"), "jellyfish": ("ckpt_jellyfish", "model-1.pt")}


def _file(task):
    d, f = CK[task]
    return os.path.join(GOLDEN, d), torch.load(os.path.join(GOLDEN, d, f), map_location="cpu")


def _assert_loaded(module, ref_sd):
    own = module.state_dict()
    ref = {k: v for k, v in ref_sd.items() if not k.endswith("rotary_emb.freqs")}
    # (the reference registers the rotary table as a buffer per attention block; here it is a host-side constant of the kernel:
    # the only keys of a reference checkpoint that have no counterpart)
    assert sorted(own) == sorted(ref), (sorted(set(own) ^ set(ref))[:8])
    for k, v in ref.items():
        assert own[k].shape == v.shape and torch.equal(own[k].cpu().to(v.dtype), v), k
    assert len(ref_sd) - len(ref) > 0 or not any("temporal" in k for k in ref_sd)


def test_smoke_trainer_reads_a_reference_checkpoint():
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion, Trainer
    d, data = _file("smoke")
    assert sorted(k for k in data if not k.startswith("_")) == ["ema", "model", "opt", "scaler", "step"]
    m = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2), channels=6)
    gd = GaussianDiffusion(m, image_size=16, frames=4, timesteps=1000, sampling_timesteps=100, loss_type="l2", objective="pred_noise")
    tr = Trainer(gd, results_path=d)
    tr.load(1)
    assert tr.step == 7
    _assert_loaded(gd, data["model"])
    # Optimizer state: index i of Adam's state_dict = position in the REFERENCE's diffusion_model.parameters().  The trainer fills its
    # flat buffers through _param_order(): the names behind the reference's indices (stored next to the checkpoint by the generator)
    # must be the trainer's, slot by slot -- `ups` before the mid blocks, and the rotary table's slot (a requires_grad=False Parameter of
    # the reference: it owns an index, no gradient, hence no state) where the reference has it.
    order = tr._param_order()
    names = data["_param_names"]
    assert len(order) == len(names) == len(data["opt"]["param_groups"][0]["params"])
    st = data["opt"]["state"]
    for i, ((k, shape), ref_name) in enumerate(zip(order, names)):
        if k is None:
            assert ref_name.endswith("rotary_emb.freqs") and i not in st, (i, ref_name)
            continue
        assert "model." + k == ref_name, (i, k, ref_name)
        assert tuple(st[i]["exp_avg"].shape) == shape == tuple(st[i]["exp_avg_sq"].shape), (i, k)
        assert int(st[i]["step"]) == 1
        # the generator's gradient of parameter i was 1e-3 (1 + i) everywhere: exp_avg = (1 - 0.9) x that names the position
        assert torch.allclose(st[i]["exp_avg"], torch.full_like(st[i]["exp_avg"], 1e-4 * (1 + i)), rtol=1e-5), (i, k)
    assert sum(1 for k, _ in order if k is None) == 1
    # EMA: the keys the trainer reads when its buffers exist
    opt, ema = tr._pending
    for k, _ in order:
        assert k is None or "ema_model.model." + k in ema
    assert bool(ema["initted"]) and int(ema["step"]) == 10


def test_burgers_trainer_reads_a_reference_checkpoint():
    from diffphycon_amd.model.burgers_1d.unet import Unet2D
    from diffphycon_amd.diffusion.diffusion_1d_burgers import GaussianDiffusion, Trainer
    d, data = _file("burgers")
    assert data["version"] == "1.0"
    m = Unet2D(dim=8, out_dim=2, dim_mults=(1, 2), channels=2, resnet_block_groups=1)
    gd = GaussianDiffusion(m, seq_length=(16, 32), timesteps=1000, auto_normalize=False, use_conv2d=True, temporal=True)
    tr = Trainer(gd, None, results_folder=d)
    tr.load(1)
    assert tr.step == 11
    _assert_loaded(gd, data["model"])


def test_jellyfish_trainer_reads_a_reference_checkpoint():
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffphycon_amd.diffusion.diffusion_2d_jellyfish import GaussianDiffusion, Trainer
    d, data = _file("jellyfish")
    m = Unet3D_with_Conv3D(dim=8, out_dim=4, dim_mults=(1, 2), channels=7)
    gd = GaussianDiffusion(m, image_size=16, frames=4, cond_steps=1, timesteps=1000, sampling_timesteps=1000, loss_type="l2",
                           objective="pred_noise", device="cpu")
    tr = Trainer(gd, results_path=d)
    tr.load(1)
    assert tr.step == 13
    _assert_loaded(gd, data["model"])


def test_rotary_table_matches_the_wheel_when_it_is_installed():
    """SURVEY 8 row A15 is `parity unpinned`: rotary-embedding-torch 0.8.4 is not in this image (no network).  This is the test that
    flips the row: where the wheel IS importable, the host-side rotary table of the fused temporal attention and the oracle's
    restatement are compared with RotaryEmbedding(dim=32).rotate_queries_or_keys."""
    ret = pytest.importorskip("rotary_embedding_torch")
    from oracle import unet3d as O
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import _rotary_tables
    torch.manual_seed(0)
    q = torch.randn(2, 4, 20, 32)                      # (..., heads, frames, dim_head): the temporal attention's layout
    want = ret.RotaryEmbedding(32).rotate_queries_or_keys(q)
    assert torch.allclose(O.rotary(q), want, atol=1e-6), (O.rotary(q) - want).abs().max()
    cos, sin = _rotary_tables(20, 32)                  # the table the fused temporal attention kernel multiplies by
    x = q.reshape(*q.shape[:-1], 16, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(q.shape)
    got = q * torch.as_tensor(cos) + rot * torch.as_tensor(sin)
    assert torch.allclose(got, want, atol=1e-6), (got - want).abs().max()
