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


@pytest.mark.parametrize("bwd_mode,loss_scale", [("x6", 1.0), ("f32", 1.0), ("f16x3", 2.0 ** 12)])
@pytest.mark.parametrize("tag", ["joint", "w", "wide"])
def test_loss_and_every_gradient_match_the_reference(tag, bwd_mode, loss_scale, dev):
    """(r04) 'f16x3' = backward-data convolutions on 22-bit split operands under a power-of-two loss scale: same records, same
    tolerance.  (The fixture nets are tiny -- d loss / d eps = 2 (eps - noise) / numel is ~1e-3 per element, 1e4 times the S64
    net's -- so the scale that suits them is 2^12, not S64's 2^20, which the sentinel rejects: next test.)"""
    g = load_golden(f"train_{tag}")
    T = _net(g, dev, bwd_mode, loss_scale=loss_scale)
    a, b = _sched(dev)
    coff = 3 if int(g["channels"]) == 2 else 0
    x0 = torch.from_numpy(g["s0:state"]).to(dev)
    loss = T.p_losses(x0, torch.from_numpy(g["s0:t"]).to(dev), torch.from_numpy(g["s0:noise"]).to(dev), a, b, channel_offset=coff)
    assert abs(loss.item() - float(g["s0:loss"])) < 1e-5 * float(g["s0:loss"])
    _check_grads(T, g, 0, 1e-4, scale=loss_scale)
    _lib_status = __import__("diffphycon_amd._lib", fromlist=["x"])
    assert _lib_status.lib().dpc_train_range_status(1, _lib_status.stream()) == 0        # nothing was clamped on the way


@pytest.mark.parametrize("tag", ["w"])
def test_a_loss_scale_the_gradients_cannot_take_raises_the_sentinel(tag, dev):
    """r04: 2^20 on the tiny prior-net fixture scales its output gradients past 4094, where the f16x3 backward-data convolutions
    clamp -- silently before r04 (gradients off by 5 %).  Every layer's weight-gradient launch now watches its gradient operand
    (dpc_conv_wgrad_cl: dy_abs_limit), so the step is reported: dpc_train_range_status != 0, Trainer.check_gradient_range raises,
    and the dynamic scaler (next test) skips it."""
    from diffphycon_amd import _lib as L
    g = load_golden(f"train_{tag}")
    T = _net(g, dev, "f16x3", loss_scale=2.0 ** 20)
    a, b = _sched(dev)
    coff = 3 if int(g["channels"]) == 2 else 0
    L.lib().dpc_train_range_status(1, L.stream())
    T.p_losses(torch.from_numpy(g["s0:state"]).to(dev), torch.from_numpy(g["s0:t"]).to(dev), torch.from_numpy(g["s0:noise"]).to(dev), a, b,
               channel_offset=coff)
    assert L.lib().dpc_train_range_status(1, L.stream()) != 0
    assert "output gradient" in L.lib().dpc_last_error().decode()


def _trainer(g, dev, bwd_mode="x6", **kw):
    from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion, Trainer
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    ch = int(g["channels"])
    m = Unet3D_with_Conv3D(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=ch)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w0:")})
    gd = GaussianDiffusion(m, image_size=int(g["hw"]), frames=int(g["frames"]), timesteps=1000, sampling_timesteps=250, loss_type="l2",
                           objective="pred_noise", device=dev)
    return Trainer(gd, "Smoke", None, train_batch_size=2, train_lr=float(g["lr"]), is_w_model=(ch == 2), bwd_mode=bwd_mode, **kw)


@pytest.mark.parametrize("tag", ["joint", "w", "wide"])
def test_optimizer_steps_match_the_reference(tag, dev):
    """Trainer.train's step (:1011-1043): backward, clip_grad_norm_(1.0), Adam(lr 1e-3, betas (0.9, 0.99)), on the reference's
    recorded batches; the gradient norm and the post-Adam weights of step 1 and step 2 against the reference's."""
    g = load_golden(f"train_{tag}")
    tr = _trainer(g, dev)
    lr = float(g["lr"])
    for step in range(1 if tag == "wide" else 2):
        loss = tr.loss_and_gradients(torch.from_numpy(g[f"s{step}:state"]).to(dev), torch.from_numpy(g[f"s{step}:t"]).to(dev),
                                     torch.from_numpy(g[f"s{step}:noise"]).to(dev))
        assert abs(loss.item() - float(g[f"s{step}:loss"])) < 2e-5 * float(g[f"s{step}:loss"]), (step, loss.item())
        _check_grads(tr._t, g, step, 2e-4 if step else 1e-4)
        tr.optimizer_step()
        assert abs(tr.norm.item() - float(g[f"s{step}:grad_norm"])) < 1e-4 * float(g[f"s{step}:grad_norm"])
        G = max(float(np.abs(g[f"s0:g:{k}"]).max()) for k in tr._t.names)
        sd = tr.model.model.state_dict()
        for k in tr._t.names:
            ref = torch.from_numpy(g[f"s{step}:w:{k}"])
            d = (sd[k].cpu() - ref).abs()
            # Adam's first steps move a weight by ~lr sign(g): elements whose recorded gradient is rounding noise take an
            # arbitrary +-lr step on either side; everything else must agree to a fraction of lr
            live = torch.from_numpy(np.abs(g[f"s0:g:{k}"]) > 1e-4 * G)
            assert d[live].numel() == 0 or d[live].max().item() < 3e-2 * lr, (step, k, d[live].max().item())
            assert d.max().item() < 2.1 * lr * (step + 1), (step, k)


def test_dynamic_loss_scale_matches_the_reference_steps_and_skips_on_overflow(dev):
    """r04: Trainer(bwd_mode="f16x3", loss_scale="dynamic") -- backward-data convolutions on the forward's 22-bit split operands, the
    gradients scaled by a power of two that follows accelerate's GradScaler rule (the reference's Trainer(fp16=True), :871-874): the
    two recorded steps match the reference like the exact mode does; an overflowing output gradient (forced here by a loss scale the
    gradients cannot take) skips the step on the device-resident norm, halves the scale and leaves weights / moments / EMA / step
    counters untouched."""
    g = load_golden("train_joint")
    tr = _trainer(g, dev, bwd_mode="f16x3", loss_scale="dynamic")
    lr = float(g["lr"])
    assert tr.loss_scale == 2.0 ** 20
    w0 = None
    for step in range(2):
        for attempt in range(16):            # (a tiny net's gradients can be 1e4 x S64's: the scaler may walk down from 2^20 first)
            loss = tr.loss_and_gradients(torch.from_numpy(g[f"s{step}:state"]).to(dev), torch.from_numpy(g[f"s{step}:t"]).to(dev),
                                         torch.from_numpy(g[f"s{step}:noise"]).to(dev))
            assert abs(loss.item() - float(g[f"s{step}:loss"])) < 2e-5 * float(g[f"s{step}:loss"])
            if w0 is None:
                w0 = tr._t.w.clone()
            if tr.optimizer_step():
                break
            assert step == 0 and torch.equal(tr._t.w, w0) and tr.opt_step == 0          # a skipped step changes nothing
        _check_grads(tr._t, g, step, 2e-4 if step else 1e-4, scale=tr._t.loss_scale)
        assert abs(tr.norm.item() - float(g[f"s{step}:grad_norm"])) < 1e-4 * float(g[f"s{step}:grad_norm"])
    assert 2.0 ** 8 <= tr.loss_scale <= 2.0 ** 20 and tr.opt_step == 2
    print(f"dynamic loss scale settled at 2^{int(np.log2(tr.loss_scale))} after {tr.skipped_steps} skipped steps")
    G = max(float(np.abs(g[f"s0:g:{k}"]).max()) for k in tr._t.names)
    sd = tr.model.model.state_dict()
    for k in tr._t.names:
        d = (sd[k].cpu() - torch.from_numpy(g[f"s1:w:{k}"])).abs()
        live = torch.from_numpy(np.abs(g[f"s0:g:{k}"]) > 1e-4 * G)
        assert d[live].numel() == 0 or d[live].max().item() < 3e-2 * lr, (k, d[live].max().item())
    # ---- overflow: 2^44 puts d loss / d eps ~ 1e-4 * 1.8e13 far outside the fp16 window of the weight-gradient operands
    before = (tr._t.w.clone(), tr.m.clone(), tr.v.clone(), tr.ema.clone(), tr.opt_step, tr.ema_sched.step, tr._t.loss_scale)
    tr._t.set_loss_scale(2.0 ** 44)
    tr.loss_and_gradients(torch.from_numpy(g["s0:state"]).to(dev), torch.from_numpy(g["s0:t"]).to(dev), torch.from_numpy(g["s0:noise"]).to(dev))
    assert tr.optimizer_step() is False
    assert tr._t.loss_scale == 2.0 ** 43 and tr.good_steps == 0
    after = (tr._t.w, tr.m, tr.v, tr.ema, tr.opt_step, tr.ema_sched.step)
    assert all(torch.equal(a, b) if torch.is_tensor(a) else a == b for a, b in zip(before[:6], after))
    tr.check_gradient_range()                                     # the poison kernel consumed the gradient bit; no activation bit was raised
    # growth: after `scale_growth_interval` clean steps the scale doubles
    settled = float(before[-1])
    # ---- the checkpoint carries the scaler in torch.cuda.amp.GradScaler.state_dict()'s layout (the reference stores accelerate's, :951)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        tr.results_path = __import__("pathlib").Path(d)
        tr._t.set_loss_scale(2.0 ** 17)
        tr.loss_scale, tr.good_steps = 2.0 ** 17, 5
        tr.save(3)
        sc = torch.load(str(tr.results_path / "model-3.pt"), weights_only=False)["scaler"]
        assert sc["scale"] == 2.0 ** 17 and sc["_growth_tracker"] == 5 and sc["backoff_factor"] == 0.5 and sc["growth_interval"] == 2000
        tr2 = _trainer(g, dev, bwd_mode="f16x3", loss_scale="dynamic", results_path=d)
        tr2.load(3)
        assert tr2.loss_scale == 2.0 ** 17 and tr2.good_steps == 5
        tr2._ensure()
        assert tr2._t.loss_scale == 2.0 ** 17
    tr._t.set_loss_scale(settled / 2)
    tr.scale_growth_interval, tr.good_steps = 2, 0
    for _ in range(2):
        tr.loss_and_gradients(torch.from_numpy(g["s0:state"]).to(dev), torch.from_numpy(g["s0:t"]).to(dev), torch.from_numpy(g["s0:noise"]).to(dev))
        assert tr.optimizer_step() is True
    assert tr._t.loss_scale == settled


def test_training_step_is_bit_reproducible_and_inference_sees_the_trained_weights(dev):
    g = load_golden("train_joint")
    outs = []
    for _ in range(2):
        tr = _trainer(g, dev)
        for step in range(2):
            tr.loss_and_gradients(torch.from_numpy(g[f"s{step}:state"]).to(dev), torch.from_numpy(g[f"s{step}:t"]).to(dev),
                                  torch.from_numpy(g[f"s{step}:noise"]).to(dev))
            tr.optimizer_step()
        outs.append((tr._t.w.clone(), tr._t.g.clone(), tr.ema.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))         # fixed-order reductions everywhere: bit-identical
    # the denoiser's inference path (dpc_unet3d_forward) runs on the updated weights: compare with the oracle on them
    from oracle import unet3d as O
    cfg = O.Unet3DConfig(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=int(g["channels"]))
    sd = {k: v.detach().cpu().clone() for k, v in tr.model.model.state_dict().items()}
    x = torch.from_numpy(g["s0:state"])
    t = torch.from_numpy(g["s0:t"])
    with torch.no_grad():
        ref = O.unet3d_forward(sd, cfg, x, t)
    y = tr.model.model(x.to(dev), t.to(dev)).cpu()
    assert ((y - ref).abs().max() / ref.abs().max()).item() < 1e-4
    assert (sd["init_conv.weight"] - torch.from_numpy(g["w0:init_conv.weight"])).abs().max() > 1e-4      # they did move


def test_checkpoint_resume_is_bit_identical_to_the_uninterrupted_run(dev, tmp_path):
    """Trainer.save / load (:942-985): Trainer(...); load(); train_step() -- the usual order, training buffers built AFTER the
    load -- continues with Adam's moments and step count, the LR-schedule position and the EMA (average, step, initted) of the
    file: weights, moments and average after 2 + 2 steps are bit-equal to 4 uninterrupted steps.  `opt` / `ema` are written
    in the layouts of torch.optim.Adam.state_dict() / the EMA module's state_dict()."""
    g = load_golden("train_joint")

    def run(tr, steps):
        tr.ema_sched.update_every, tr.ema_sched.update_after_step = 1, 1      # the average must leave its copy phase inside the test
        for step in steps:
            k = step % 2
            tr.loss_and_gradients(torch.from_numpy(g[f"s{k}:state"]).to(dev), torch.from_numpy(g[f"s{k}:t"]).to(dev),
                                  torch.from_numpy(g[f"s{k}:noise"]).to(dev))
            tr.optimizer_step()
            tr.step += 1
        return tr
    full = run(_trainer(g, dev, results_path=str(tmp_path)), range(4))
    first = run(_trainer(g, dev, results_path=str(tmp_path)), range(2))
    first.save(7)
    ck = torch.load(str(tmp_path / "model-7.pt"), map_location="cpu")
    n = len(list(first.model.parameters()))
    # (index space = the reference's parameters(): one more slot than trainable parameters -- its rotary table -- which holds no state)
    assert set(ck["opt"]) == {"state", "param_groups"} and len(ck["opt"]["state"]) == n and ck["opt"]["param_groups"][0]["params"] == list(range(n + 1))
    assert {"initted", "step", "ema_model.model.init_conv.weight", "online_model.model.init_conv.weight", "ema_model.betas"} <= set(ck["ema"])
    resumed = _trainer(g, dev, results_path=str(tmp_path))
    resumed.load(7)                                  # no training buffers yet: the state is applied when they are built
    assert resumed._t is None and resumed.step == 2
    run(resumed, range(2, 4))
    assert resumed.opt_step == full.opt_step == 4 and resumed.ema_sched.step == full.ema_sched.step
    for name in ("m", "v", "ema"):
        assert torch.equal(getattr(resumed, name), getattr(full, name)), name
    assert torch.equal(resumed._t.w, full._t.w)
    assert not torch.equal(full.ema, full._t.w)      # (the average is a real average by now, not the copy of the first 100 steps)
    # a trainer whose buffers already exist takes the state at once; an optimizer state in another layout is refused
    again = run(_trainer(g, dev, results_path=str(tmp_path)), range(1))
    again.load(7)
    run(again, range(2, 4))
    assert torch.equal(again._t.w, full._t.w) and torch.equal(again.ema, full.ema)
    ck["opt"] = {"step": 2, "exp_avg": torch.zeros(3)}
    torch.save(ck, str(tmp_path / "model-8.pt"))
    with pytest.raises(ValueError, match="Adam state_dict"):
        _trainer(g, dev, results_path=str(tmp_path)).load(8)


def test_resume_from_a_checkpoint_in_the_reference_layout(dev):
    """tests/golden/ckpt_smoke/model-1.pt: the imported reference's GaussianDiffusion.state_dict() + a real torch.optim.Adam.state_dict()
    over ITS parameters() (tools/gen_golden_r06.py; CPU side of the same file: tests/test_checkpoint_layout.py).  Trainer.load() + the
    first use of the training buffers must put the moment stored at the reference's index i into the flat buffer of the parameter that
    owns index i there: the generator's first moment is 1e-4 (1 + i) everywhere in parameter i."""
    import os
    from conftest import GOLDEN
    from diffphycon_amd.diffusion.diffusion_2d_smoke import GaussianDiffusion, Trainer
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    m = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2), channels=6)
    gd = GaussianDiffusion(m, image_size=16, frames=4, timesteps=1000, sampling_timesteps=100, loss_type="l2", objective="pred_noise", device=dev)
    tr = Trainer(gd, "Smoke", None, train_batch_size=2, results_path=os.path.join(GOLDEN, "ckpt_smoke"), bwd_mode="x6")
    tr.load(1)
    T = tr._ensure()                                  # builds the flat buffers and applies the pending optimizer / EMA state
    ck = torch.load(os.path.join(GOLDEN, "ckpt_smoke", "model-1.pt"), map_location="cpu")
    assert tr.opt_step == 1 and tr.step == 7
    names = ck["_param_names"]
    checked = 0
    for i, name in enumerate(names):
        if name.endswith("rotary_emb.freqs"):
            continue
        k = name[len("model."):]
        o, n = T.offsets[k], T.ctx.W[k].numel()
        assert torch.allclose(tr.m[o:o + n].cpu(), torch.full((n,), 1e-4 * (1 + i)), rtol=1e-5), (i, k)
        assert torch.equal(tr.ema[o:o + n].cpu(), ck["ema"]["ema_model.model." + k].reshape(-1)), k
        checked += 1
    assert checked == len(names) - 1 == len(T.names)


@pytest.mark.parametrize("bwd_mode,loss_scale", [("x6", 1.0), ("f16x3", 2.0 ** 20)])
def test_full_width_gradients_vs_autograd_on_the_oracle(bwd_mode, loss_scale, dev):
    """dim 64, mults (1, 2, 4) -- the widths train_2d_smoke.py:43-47 builds -- at 8 frames x 16 x 16, B = 2: every parameter
    gradient against torch autograd through the CPU oracle.  'f16x3' runs the backward-data convolutions on 22-bit split operands
    with a power-of-two loss scale (undone by the optimizer kernel)."""
    from oracle import train_smoke as TS
    from oracle import unet3d as O
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffphycon_amd.model.video_diffusion_pytorch.unet3d_train import TrainableUnet3D
    cfg = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6)
    sd = O.synthetic_state_dict(cfg, seed=61)
    gen = torch.Generator().manual_seed(62)
    x0 = torch.randn(2, 8, 6, 16, 16, generator=gen) * 0.5
    noise = torch.randn(2, 8, 6, 16, 16, generator=gen)
    t = torch.tensor([700, 40])
    sched = TS.schedule(1000)
    loss_ref, grads_ref = TS.loss_and_grads(sd, cfg, sched, x0, t, noise)
    m = Unet3D_with_Conv3D(dim=64, dim_mults=(1, 2, 4), channels=6)
    m.load_state_dict(sd)
    T = TrainableUnet3D(m, dev, bwd_mode=bwd_mode, loss_scale=loss_scale)
    a, b = _sched(dev)
    loss = T.p_losses(x0.to(dev), t.to(dev), noise.to(dev), a, b)
    assert abs(loss.item() - loss_ref.item()) < 2e-5 * loss_ref.item()
    G = max(v.abs().max().item() for v in grads_ref.values())
    worst, bad = 0.0, []
    for k, ref in grads_ref.items():
        got = T.ctx.G[k].cpu().reshape(ref.shape) / loss_scale
        err = (got - ref).abs().max().item()
        rel = err / (ref.abs().max().item() + 1e-5 * G)
        worst = max(worst, rel)
        if rel > 5e-4:
            bad.append((k, rel))
    print(f"full-width gradients ({bwd_mode}): worst relative deviation {worst:.2e} over {len(grads_ref)} tensors")
    assert not bad, bad[:8]


# ------------------------------------------------------------------------------------------------ operator level
def _cl(x):            # [B, C, F, H, W] -> channels-last rows [B F H W, C]
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


WGRAD_CASES = [
    # B, F, Hi, Wi, C, N, k (kd, kh, kw), stride (h, w), pad (d, h, w), f16x3 path?
    (2, 5, 16, 64, 64, 64, (3, 3, 3), (1, 1), (1, 1, 1), True),
    (1, 4, 8, 64, 32, 128, (3, 3, 3), (1, 1), (1, 1, 1), True),
    (2, 3, 32, 32, 64, 64, (3, 3, 3), (1, 1), (1, 1, 1), True),
    (3, 2, 16, 16, 128, 128, (3, 3, 3), (1, 1), (1, 1, 1), True),
    (1, 9, 4, 16, 256, 64, (3, 3, 3), (1, 1), (1, 1, 1), True),
    (2, 3, 8, 8, 64, 64, (3, 3, 3), (1, 1), (1, 1, 1), True),          # W = 8: exact fp32 kernel (flag ignored)
    (2, 4, 16, 16, 8, 16, (3, 3, 3), (1, 1), (1, 1, 1), False),
    (1, 3, 6, 10, 64, 96, (3, 3, 3), (1, 1), (1, 1, 1), False),
    (2, 2, 8, 8, 128, 384, (1, 1, 1), (1, 1), (0, 0, 0), False),
    (2, 3, 16, 16, 64, 64, (1, 4, 4), (2, 2), (0, 1, 1), False),
    (1, 8, 16, 16, 8, 64, (7, 7, 7), (1, 1), (3, 3, 3), False),
    (2, 4, 12, 12, 4, 16, (7, 7, 7), (1, 1), (3, 3, 3), False),
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_weight_gradient(case, dev):
    """dpc_conv_wgrad_cl against torch autograd in fp64: every conv geometry of the denoiser (3x3x3 on the fp16 matrix cores and
    on the exact fp32 kernel, 1x1x1, the strided (1,4,4) downsample, the 7x7x7 stem with zero-padded channels), with the gradient
    operand at the magnitudes a mean-reduced loss produces (1e-6) so that the operand scaling is exercised."""
    import ctypes as C
    import torch.nn.functional as F_
    from diffphycon_amd import _lib as L
    B, Fr, H, W, Ci, N, k, st, pd, f16 = case
    g = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    x = torch.randn(B, Ci, Fr, H, W, generator=g)
    wref = torch.zeros(N, Ci, *k, dtype=torch.float64, requires_grad=True)
    y = F_.conv3d(x.double(), wref, None, stride=(1,) + st, padding=pd)
    dy = torch.randn(y.shape, generator=g) * 1e-6
    y.backward(dy.double())
    ref = wref.grad.float()
    Ho, Wo = y.shape[3], y.shape[4]
    xd, dyd = _cl(x).to(dev), _cl(dy).to(dev)
    dw = torch.full((N, Ci + 3, *k), 7.0, device=dev)            # written as a channel slice [2, 2 + Ci) of a wider weight
    lib = L.lib()
    nb = lib.dpc_conv_wgrad_workspace_bytes(Ci, N, *k, B * Fr * Ho)
    ws = L.workspace(nb, dev)
    L.check(lib.dpc_conv_wgrad_cl(L.ptr(xd), L.ptr(dyd), L.ptr(dw), B, Fr, H, W, Ci, Ho, Wo, N, *k, *st, *pd, 0, Ci + 3, 2, 1.0,
                                  2.0 ** 20 if f16 else 0.0, 0.0, 0, C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    got = dw.cpu()
    assert torch.all(got[:, :2] == 7.0) and torch.all(got[:, 2 + Ci:] == 7.0)      # nothing outside the slice is touched
    err = (got[:, 2:2 + Ci] - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6, err


def test_f16x3_weight_gradient_saturation_is_reported(dev):
    """the f16x3 weight-gradient kernel clamps dy * f16_dy_scale at 65504 and x * 2^4 likewise; a clamped (or non-finite)
    operand raises the device word that dpc_train_range_status reads.  In range: status OK.  One dy element past the window, one
    activation past 4094, one NaN: each is reported (and cleared by the read), and the clamped launch still returns finite numbers."""
    import ctypes as C
    from diffphycon_amd import _lib as L
    lib = L.lib()
    B, Fr, H, W, Ci, N = 1, 3, 16, 16, 32, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B * Fr * H * W, Ci, generator=g).to(dev)
    dy = (torch.randn(B * Fr * H * W, N, generator=g) * 1e-6).to(dev)
    ws = L.workspace(lib.dpc_conv_wgrad_workspace_bytes(Ci, N, 3, 3, 3, B * Fr * H), dev)
    dw = torch.empty(N, Ci, 3, 3, 3, device=dev)

    def run(xx, dd):
        L.check(lib.dpc_conv_wgrad_cl(L.ptr(xx), L.ptr(dd), L.ptr(dw), B, Fr, H, W, Ci, H, W, N, 3, 3, 3, 1, 1, 1, 1, 1, 0, Ci, 0, 1.0,
                                      2.0 ** 20, 0.0, 0, C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
        return lib.dpc_train_range_status(1, L.stream())
    lib.dpc_train_range_status(1, L.stream())                     # clear whatever an earlier test left
    assert run(x, dy) == 0
    big = dy.clone()
    big[100, 7] = 0.5                                             # 0.5 * 2^20 > 65504
    assert run(x, big) != 0 and b"output gradient" in lib.dpc_last_error()
    assert torch.isfinite(dw).all()
    assert lib.dpc_train_range_status(1, L.stream()) == 0         # the read cleared it
    xb = x.clone()
    xb[5, 3] = 5000.0
    assert run(xb, dy) != 0 and b"activation" in lib.dpc_last_error()
    assert torch.isfinite(dw).all()
    bad = dy.clone()
    bad[0, 0] = float("nan")
    assert run(x, bad) != 0
    lib.dpc_train_range_status(1, L.stream())


def test_a_halved_loss_scale_widens_the_f16x3_weight_gradient_window(dev):
    """with backward-data in f16x3 the dy operand of wgrad3 is split as (loss_scale * d loss) * 2^4 -- the window of the
    backward-data convolutions -- so an output gradient that trips the sentinel at 2^20 (the dynamic scaler's start) fits at 2^19.
    (r04 derived the operand scale as max(2^4, 2^24 / loss_scale): the split operand was d loss * 2^24 at every scale <= 2^20, the
    halving never helped and the run ended with FloatingPointError at scale 1.)  A wgrad3-eligible shape (W = 16, C % 32 == 0,
    N % 64 == 0), which the tiny fixture nets never reach."""
    import ctypes as C
    from diffphycon_amd import _lib as L
    lib = L.lib()
    g = load_golden("train_w")
    T = _net(g, dev, "f16x3", loss_scale=2.0 ** 20)
    B, Fr, H, W, Ci, N = 1, 3, 16, 16, 32, 64
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(B * Fr * H * W, Ci, generator=gen).to(dev)
    dy_true = torch.randn(B * Fr * H * W, N, generator=gen) * 1e-4
    dy_true[37, 5] = 6.0e-3                  # x 2^20 x 2^4 = 100 663 > 65504; x 2^19 x 2^4 = 50 331 fits
    ws = L.workspace(lib.dpc_conv_wgrad_workspace_bytes(Ci, N, 3, 3, 3, B * Fr * H), dev)
    dw = torch.empty(N, Ci, 3, 3, 3, device=dev)
    ref = None
    status = {}
    for e in (20, 19, 18):
        T.set_loss_scale(2.0 ** e)
        assert T.ctx.wgrad_dy_scale == 16.0
        dy = (dy_true * 2.0 ** e).to(dev)
        lib.dpc_train_range_status(1, L.stream())
        L.check(lib.dpc_conv_wgrad_cl(L.ptr(x), L.ptr(dy), L.ptr(dw), B, Fr, H, W, Ci, H, W, N, 3, 3, 3, 1, 1, 1, 1, 1, 0, Ci, 0, 1.0,
                                      T.ctx.wgrad_dy_scale, T.ctx.dgrad_limit, 0, C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
        status[e] = lib.dpc_train_range_status(1, L.stream())
        if status[e] == 0:
            got = dw / 2.0 ** e
            if ref is None:
                xd = x.double().cpu().reshape(B, Fr, H, W, Ci).permute(0, 4, 1, 2, 3).requires_grad_(False)
                w = torch.zeros(N, Ci, 3, 3, 3, dtype=torch.float64, requires_grad=True)
                y = torch.nn.functional.conv3d(xd, w, padding=1)
                y.backward(dy_true.double().reshape(B, Fr, H, W, N).permute(0, 4, 1, 2, 3))
                ref = w.grad
            err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
            assert err < 1e-5, (e, err)
    assert status[20] != 0 and status[19] == 0 and status[18] == 0, status
    # exact backward-data products keep the r04 rule: the loss scale is fixed and the split operand sits at O(1)
    Tx = _net(g, dev, "x6", loss_scale=1.0)
    assert Tx.ctx.wgrad_dy_scale == 2.0 ** 24


def test_fp32_weight_gradient_names_the_operand_that_left_the_window(dev):
    """the fp32 weight-gradient kernel watched both operands with one maximum and raised the GRADIENT bit for an
    ACTIVATION above the limit -- the dynamic scaler then halved the scale on every step until it died at scale 1 with the wrong
    message.  Now: activation operand -> bit 0 ("activation", left for dpc_train_range_status, untouched by the poison kernel),
    gradient operand -> bit 1; the ConvTranspose call form (accumulate bit 1: x is the gradient) swaps the roles."""
    import ctypes as C
    from diffphycon_amd import _lib as L
    lib = L.lib()
    B, Fr, H, W, Ci, N = 1, 2, 8, 8, 32, 32
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(B * Fr * H * W, Ci, generator=gen).to(dev)
    dy = torch.randn(B * Fr * H * W, N, generator=gen).to(dev)
    ws = L.workspace(lib.dpc_conv_wgrad_workspace_bytes(Ci, N, 1, 1, 1, B * Fr * H), dev)
    dw = torch.empty(N, Ci, 1, 1, 1, device=dev)
    gbuf = torch.zeros(64, device=dev)

    def run(xx, dd, acc):
        lib.dpc_train_range_status(1, L.stream())
        L.check(lib.dpc_conv_wgrad_cl(L.ptr(xx), L.ptr(dd), L.ptr(dw), B, Fr, H, W, Ci, H, W, N, 1, 1, 1, 1, 1, 0, 0, 0, 0, Ci, 0, 1.0,
                                      0.0, 4094.0, acc, C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
        gbuf.zero_()
        L.check(lib.dpc_train_range_poison(L.ptr(gbuf), L.stream()))          # consumes the gradient bit only
        poisoned = bool(torch.isinf(gbuf[0]).item())
        st = lib.dpc_train_range_status(1, L.stream())
        return poisoned, st, lib.dpc_last_error().decode() if st else ""
    assert run(x, dy, 0) == (False, 0, "")
    xb = x.clone()
    xb[9, 4] = 5000.0
    poisoned, st, msg = run(xb, dy, 0)
    assert not poisoned and st != 0 and "activation" in msg and "output gradient" not in msg
    db = dy.clone()
    db[9, 4] = 5000.0
    poisoned, st, msg = run(x, db, 0)
    assert poisoned and st == 0                                   # the poison kernel cleared the gradient bit
    # ConvTranspose form: x carries the gradient
    poisoned, st, msg = run(xb, dy, 2)
    assert poisoned and st == 0
    poisoned, st, msg = run(x, db, 2)
    assert not poisoned and st != 0 and "activation" in msg
    assert lib.dpc_conv_wgrad_cl(L.ptr(x), L.ptr(dy), L.ptr(dw), B, Fr, H, W, Ci, H, W, N, 1, 1, 1, 1, 1, 0, 0, 0, 0, Ci, 0, 1.0, 0.0, 0.0, 4,
                                 C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()) != 0


def test_column_reductions_and_small_linear_backward(dev):
    import ctypes as C
    from diffphycon_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(3)
    for rows, Cc in ((4099, 64), (700, 256), (16, 512), (333, 8)):
        dy, x = torch.randn(rows, Cc, generator=g), torch.randn(rows, Cc, generator=g) * 2 + 0.5
        mean, var = x.double().mean(1), x.double().var(1, unbiased=False)
        st = torch.stack((mean, (var + 1e-5).rsqrt()), 1).float()
        ws = L.workspace(lib.dpc_colsum_workspace_bytes(Cc), dev)
        dyd, xd, std = dy.to(dev), x.to(dev), st.to(dev)          # (named: a temporary would be freed before the kernel runs)
        for with_x in (False, True):
            out = torch.empty(Cc, device=dev)
            L.check(lib.dpc_colsum(L.ptr(dyd), L.ptr(xd) if with_x else None, L.ptr(std) if with_x else None,
                                   L.ptr(out), rows, Cc, 1.0, 0, C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
            ref = (dy.double() * ((x.double() - mean[:, None]) * st[:, 1:].double()) if with_x else dy.double()).sum(0)
            assert (out.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    for B, K, N, act in ((16, 64, 256, 0), (16, 256, 256, 2), (5, 256, 128, 1)):
        dy, x, Wt = torch.randn(B, N, generator=g), torch.randn(B, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
        xr = x.double().requires_grad_(True)
        Wr = Wt.double().requires_grad_(True)
        a = {0: lambda v: v, 1: torch.nn.functional.silu, 2: torch.nn.functional.gelu}[act](xr)
        (a @ Wr.t()).backward(dy.double())
        dx = torch.ones(B, K, device=dev)
        dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        dyd, xd, Wd = dy.to(dev), x.to(dev), Wt.to(dev)
        L.check(lib.dpc_small_linear_bwd(L.ptr(dyd), L.ptr(xd), L.ptr(Wd), L.ptr(dx), L.ptr(dW), L.ptr(db), B, K, N, act, 1, L.stream()))
        assert (dW.cpu().double() - Wr.grad).abs().max().item() < 1e-5
        assert (dx.cpu().double() - 1 - xr.grad).abs().max().item() < 1e-5             # accumulated onto the ones
        assert (db.cpu().double() - dy.double().sum(0)).abs().max().item() < 1e-5


@pytest.mark.parametrize("L_,heads", [(32, 4), (20, 4), (64, 2)])
def test_temporal_attention_backward(L_, heads, dev):
    """dpc_attention_bwd_seq (rotary + relative-position bias, strided sequences) against autograd through the reference
    formulation (...conv3d.py:311-351) in fp64: dqkv and the bias gradient."""
    import ctypes as C
    from diffphycon_amd import _lib as L
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import _rotary_tables
    lib = L.lib()
    g = torch.Generator().manual_seed(L_)
    Bn, HW = 2, 12                                   # sequences = (b, pixel), tokens = frames at stride HW rows
    rows = Bn * L_ * HW
    qkv = torch.randn(rows, 3 * heads * 32, generator=g)
    dout = torch.randn(rows, heads * 32, generator=g)
    bias = torch.randn(heads, L_, L_, generator=g)
    cos, sin = _rotary_tables(L_, 32)

    def rot(t):                                      # t [..., L, 32]
        t2 = t.reshape(*t.shape[:-1], 16, 2)
        rh = torch.stack((-t2[..., 1], t2[..., 0]), -1).reshape(t.shape)
        return t * cos.double() + rh * sin.double()
    q0 = qkv.double().requires_grad_(True)
    b0 = bias.double().requires_grad_(True)
    t = q0.reshape(Bn, L_, HW, 3, heads, 32).permute(3, 0, 2, 4, 1, 5)              # [3, B, HW, heads, L, 32]
    q, k, v = rot(t[0] * 32 ** -0.5), rot(t[1]), t[2]
    att = torch.softmax(q @ k.transpose(-1, -2) + b0, dim=-1) @ v                     # [B, HW, heads, L, 32]
    out = att.permute(0, 3, 1, 2, 4).reshape(rows, heads * 32)
    out.backward(dout.double())
    dqkv = torch.empty(rows, 3 * heads * 32, device=dev)
    dbias = torch.full((heads, L_, L_), 1.0, device=dev)
    ws = L.workspace(lib.dpc_attention_bwd_seq_workspace_bytes(heads, L_), dev)
    qd, dd, cd, sd_, bd = qkv.to(dev), dout.to(dev), cos.to(dev), sin.to(dev), bias.to(dev)
    L.check(lib.dpc_attention_bwd_seq(L.ptr(qd), L.ptr(dd), L.ptr(dqkv), L.ptr(dbias), heads, L_, Bn * HW, HW, L_ * HW, 1,
                                      HW, L.ptr(cd), L.ptr(sd_), L.ptr(bd), 1, C.c_void_p(ws.data_ptr()), ws.numel(), L.stream()))
    assert (dqkv.cpu().double() - q0.grad).abs().max().item() < 2e-5 * q0.grad.abs().max().item()
    assert (dbias.cpu().double() - 1 - b0.grad).abs().max().item() < 2e-5 * b0.grad.abs().max().item()


def test_train_script_two_ranks_keep_bit_identical_replicas(tmp_path):
    """train/train_2d_smoke.py (the reference's entry surface) under torch.distributed.run with two ranks on cuda:0 (gloo: RCCL
    refuses two ranks on one device): each rank trains on its own samples, ONE flat-gradient all-reduce per step, and after three
    optimizer steps the replicas' weights are bit-identical; a checkpoint in the reference's format is written."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DPC_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "train/train_2d_smoke.py", "--synthetic", "True", "--train_num_steps", "3", "--batch_size", "2",
           "--image_size", "16", "--save_and_sample_every", "3", "--results_path", str(tmp_path)]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "replicas identical: True" in p.stdout and "training complete" in p.stdout, p.stdout[-2000:]
    ck = torch.load(os.path.join(str(tmp_path), "joint", "model-1.pt"), map_location="cpu")
    assert ck["step"] == 3 and "model.init_conv.weight" in ck["model"] and int(ck["opt"]["state"][0]["step"]) == 3
    assert set(ck) == {"step", "model", "opt", "ema", "scaler"}                      # Trainer.save's keys (:942-954)
