"""tests/golden/burgers_recurrence.npz: teacher-forced records of the reference's `recurrent_sample`
(/root/reference/diffusion/diffusion_1d_burgers.py:472-482: the re-noising step of --recurrence, "Universal Guidance" self
recurrence) for given (x_{t-1}, t, noise).  Build container only.   python tools/gen_golden_recurrence.py"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402


class _Dummy(nn.Module):
    channels, self_condition = 2, False


def main():
    from diffusion.diffusion_1d_burgers import GaussianDiffusion
    T = 20
    gd = GaussianDiffusion(_Dummy(), seq_length=(16, 32), timesteps=T, auto_normalize=False, use_conv2d=True, temporal=True,
                           recurrence=True, recurrence_k=2)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 2, 16, 32, generator=g)
    out = dict(T=T, x=x)
    real = torch.randn_like
    for t in (19, 7, 1, 0):
        z = torch.randn(x.shape, generator=g)
        torch.randn_like = lambda ref, _z=z: _z.clone()
        try:
            out[f"t{t}:x_t"] = gd.recurrent_sample(x.clone(), t)
        finally:
            torch.randn_like = real
        out[f"t{t}:z"] = z
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "burgers_recurrence.npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print("wrote", path)


if __name__ == "__main__":
    main()
