"""Generate tests/golden/train_{joint,w,wide}.npz by importing the reference (build container only): one and two optimizer
steps of the smoke denoiser's training path, recorded from the reference's own code --

    GaussianDiffusion.p_losses / q_sample      /root/reference/diffusion/diffusion_2d_smoke.py:791-831
    Trainer.train's step sequence              :998-1054  (backward :1025, clip_grad_norm_(1.0) :1027, Adam :912 / :1035,
                                                MultiStepLR :914 / :1037, is_w_model slice :1018-1019)
    Unet3D_with_Conv3D                         /root/reference/model/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py

    python tools/gen_golden_train.py

Per net: initial weights (the reference's default init under a stated seed), two batches (state, t, noise), and for each
step (two; one for the dim-16 net) the loss, EVERY parameter gradient (before clipping), the total gradient norm, and the weights after the Adam update.
The EMA (ema-pytorch 0.7.3, environment.yaml:41) is not importable offline and is therefore not part of the fixture
(oracle/train_smoke.py restates its published update rule: PARITY UNPINNED for that one function).
Data only; no reference source text is stored.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
torch.set_num_threads(8)

CASES = (
    # tag, channels (2 = the w model: Trainer slices state[:, :, 3:5]), dim, mults, frames, hw, seed
    ("joint", 6, 8, (1, 2), 4, 16, 100),
    ("w", 2, 8, (1, 2), 4, 16, 101),
    ("wide", 6, 16, (1, 2), 8, 16, 102),      # (one step recorded: keeps the fixture small)
)


def main():
    from model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    from diffusion.diffusion_2d_smoke import GaussianDiffusion
    from torch.optim import Adam, lr_scheduler

    for tag, channels, dim, mults, frames, hw, seed in CASES:
        torch.manual_seed(seed)
        model = Unet3D_with_Conv3D(dim=dim, dim_mults=mults, channels=channels)
        diffusion = GaussianDiffusion(model, image_size=hw, frames=frames, timesteps=1000, sampling_timesteps=250,
                                      loss_type="l2", objective="pred_noise")
        named = [(k, p) for k, p in diffusion.named_parameters() if p.requires_grad]
        arrays = dict(channels=channels, dim=dim, dim_mults=np.array(mults), frames=frames, hw=hw, lr=1e-3,
                      betas=np.array([0.9, 0.99]), max_grad_norm=1.0)
        for k, v in model.state_dict().items():
            if not k.endswith("rotary_emb.freqs"):
                arrays["w0:" + k] = v.detach().clone()
        opt = Adam(diffusion.parameters(), lr=1e-3, betas=(0.9, 0.99))                 # train_2d_smoke.py:69, Trainer :912
        sched = lr_scheduler.MultiStepLR(opt, milestones=[50000, 150000, 300000], gamma=0.1)
        B = 2
        for step in range(1 if tag == "wide" else 2):
            state6 = torch.randn(B, frames, 6, hw, hw) * 0.5
            t = torch.randint(0, 1000, (B,)).long()
            state = state6[:, :, 3:5] if channels == 2 else state6                     # Trainer.train :1018-1019
            noise = torch.randn(state.shape)
            arrays[f"s{step}:state"] = state6.clone()
            arrays[f"s{step}:t"] = t.clone()
            arrays[f"s{step}:noise"] = noise.clone()
            loss = diffusion.p_losses(state.clone(), t, noise=noise.clone())
            arrays[f"s{step}:loss"] = loss.detach().clone()
            loss.backward()
            for k, p in named:
                arrays[f"s{step}:g:" + k[len("model."):]] = p.grad.detach().clone()
            total = torch.nn.utils.clip_grad_norm_(diffusion.parameters(), 1.0)        # accelerator.clip_grad_norm_ :1027
            arrays[f"s{step}:grad_norm"] = total.detach().clone()
            opt.step()
            opt.zero_grad()
            sched.step()
            for k, v in model.state_dict().items():
                if not k.endswith("rotary_emb.freqs"):
                    arrays[f"s{step}:w:" + k] = v.detach().clone()
        conv = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
        path = os.path.join(OUT, f"train_{tag}.npz")
        np.savez_compressed(path, **conv)
        print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB", "losses",
              float(arrays["s0:loss"]), "norms", float(arrays["s0:grad_norm"]))


if __name__ == "__main__":
    main()
