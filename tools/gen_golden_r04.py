"""Fixtures of r04 (committed under tests/golden/): outputs of the CPU ORACLE (oracle/unet3d.py, oracle/sampler_smoke.py -- themselves pinned
to the imported reference by tests/test_oracle_*.py) on seeded inputs whose evaluation takes minutes of host time, so that the GPU
suite does not re-run them on every box:

  full_extent_<case>.npz : oracle.unet3d_forward at the FULL extent of a BASELINE config (tests/test_gpu_unet3d.py: _FULL_EXTENT) -- for
                           every tap and the output: the tensor's shape, its max |value| and its values at `sample_index`'s positions
                           (the test draws the same): NSAMP positions from numpy.random.RandomState(seed of the case + index of
                           the tap) and, since r05, a DETERMINISTIC set on the boundaries the kernels tile by (`edge_index`: whole
                           lines along the frame, row and column axes through first / last points of the 4 x 8 x 8 output tiles,
                           so every tile of the tensor and every seam between two tiles -- along each axis -- is touched);
  drift_chain.npz        : the final state of the 100-step free-running guided DDPM chain of
                           test_full_width_100_step_chain_f16x3_drift_vs_exact_and_oracle on the oracle.

    python tools/gen_golden_r04.py [case ...]        (no argument: everything; ~10 minutes on 8 cores, ~12 GB of host memory)

tests/test_oracle_unet3d.py re-computes the s64_prior case and the first steps of the chain on the CPU and checks the files."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
NSAMP = 4096

FULL_EXTENT = {
    "s64_joint": (dict(dim=64, dim_mults=(1, 2, 4), channels=6), 41, (1, 32, 6, 64, 64), [611]),
    "s64_prior": (dict(dim=64, dim_mults=(1, 2, 4), channels=2), 42, (1, 32, 2, 64, 64), [7]),
    "s128": (dict(dim=64, dim_mults=(1, 2, 4), channels=6), 43, (1, 64, 6, 128, 128), [250]),
    "j128_state": (dict(dim=64, dim_mults=(1, 2, 4), channels=7, out_dim=4), 44, (1, 20, 7, 128, 128), [999]),
    "j128_theta": (dict(dim=64, dim_mults=(1, 2, 4), channels=7, out_dim=1), 45, (1, 20, 7, 128, 128), [3]),
    # r05: the 2-channel prior net of the prior-reweighted S128 config (inference_2d_smoke.py:48-52, 80-84) at ITS extent
    "s128_prior": (dict(dim=64, dim_mults=(1, 2, 4), channels=2), 46, (1, 64, 2, 128, 128), [512]),
}


def edge_index(shape, channels_axis):
    """Deterministic positions (flat, C-order) of a [1, C, F, H, W] tap (channels_axis 1) or the [1, F, C, H, W] output (2):
    whole LINES through the tensor along each of the frame / row / column axes, anchored at points that are first or last in a
    4 x 8 x 8 output tile of the convolution kernels (f in {0, 3, 4, F-1}, h, w in {0, 7, 8, H-1}; plus the centre seam), for
    three channels (first, middle, last: the 64- and 128-column blocks' ends).  A line crosses every tile along its axis and both
    sides of every tile seam, whatever the workgroup -> tile (XCD chunk) order is; a wrong tile narrower than the random sample's
    reach (~2 k elements at 4096 draws from 8.4 M) cannot hide from all three families."""
    if len(shape) != 5:
        return np.zeros(0, dtype=np.int64)
    if channels_axis == 1:
        _, C, F_, H, W = shape
    else:
        _, F_, C, H, W = shape
    cs = sorted({0, C // 2, C - 1})
    fa = sorted({0, min(3, F_ - 1), min(4, F_ - 1), F_ - 1})
    ha = sorted({0, min(7, H - 1), min(8, H - 1), H // 2 - 1 if H > 1 else 0, H - 1})
    wa = sorted({0, min(7, W - 1), min(8, W - 1), W // 2 if W > 1 else 0, W - 1})
    pts = []
    for c in cs:
        pts += [(c, f, h, w) for f in range(F_) for h, w in zip(ha, wa)]                  # lines along the frame axis
        pts += [(c, f, h, w) for h in range(H) for f in fa for w in wa]                   # lines along rows
        pts += [(c, f, h, w) for w in range(W) for f in fa for h in ha]                   # lines along columns
    a = np.array(sorted(set(pts)), dtype=np.int64)
    c, f, h, w = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    return (((c * F_ + f) * H + h) * W + w) if channels_axis == 1 else (((f * C + c) * H + h) * W + w)


def sample_index(seed, k, numel, shape=None, channels_axis=1):
    """Positions (flat, C-order) of the samples of the k-th recorded tensor of a case: NSAMP random draws followed by `edge_index`."""
    rnd = np.random.RandomState(1000 * seed + k).randint(0, numel, size=min(NSAMP, numel)).astype(np.int64)
    if shape is None:
        return rnd
    return np.concatenate([rnd, edge_index(tuple(int(v) for v in shape), channels_axis)])


def full_extent_inputs(case):
    cfg_kw, seed, shape, t = FULL_EXTENT[case]
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
    return cfg_kw, seed, x, torch.tensor(t)


class _SamplingTaps(dict):
    """The `taps` argument of oracle.unet3d_forward that keeps, per tap, only what the fixture records (shape, max |value|, the values at
    `sample_index`'s positions) and drops the tensor at once: 36 taps of 268 MB each need not be resident together (the S128 cases spent
    most of their time in page faults)."""

    def __init__(self, seed):
        super().__init__()
        self.seed, self.rec = seed, {}

    def __setitem__(self, name, r):
        k = len(self.rec)
        flat = r.contiguous().reshape(-1)
        idx = sample_index(self.seed, k, flat.numel(), tuple(r.shape), 2 if name == "y" else 1)
        self.rec[name] = (np.array(r.shape, dtype=np.int64), np.float32(flat.abs().max().item()),
                          flat[torch.from_numpy(idx)].numpy().astype(np.float32))


def full_extent_record(case):
    from oracle import unet3d as O
    cfg_kw, seed, x, tt = full_extent_inputs(case)
    cfg = O.Unet3DConfig(**cfg_kw)
    sd = O.synthetic_state_dict(cfg, seed=seed)
    taps = _SamplingTaps(seed)
    with torch.no_grad():
        y = O.unet3d_forward(sd, cfg, x, tt, taps=taps)
    taps["y"] = y
    out = {"names": np.array(list(taps.rec))}
    for name, (shape, absmax, values) in taps.rec.items():
        out[f"shape:{name}"] = shape
        out[f"absmax:{name}"] = absmax
        out[f"values:{name}"] = values
    return out


DRIFT = dict(T=100, F=8, HW=16, seeds=(51, 52, 53))


def drift_chain_inputs():
    from oracle import unet3d as O
    T, F_, HW = DRIFT["T"], DRIFT["F"], DRIFT["HW"]
    cj, cw = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6), O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=2)
    sdj, sdw = O.synthetic_state_dict(cj, seed=DRIFT["seeds"][0]), O.synthetic_state_dict(cw, seed=DRIFT["seeds"][1])
    gen = torch.Generator().manual_seed(DRIFT["seeds"][2])
    noises = torch.randn(T + 1, 1, F_, 6, HW, HW, generator=gen)
    init = torch.rand(1, HW, HW, generator=gen) * 2 - 1
    return cj, cw, sdj, sdw, noises, init


def drift_chain_oracle(steps=None):
    """Final state of the chain on the oracle; `steps` < T: the state after that many steps of the T-step schedule."""
    from oracle import unet3d as O
    from oracle import sampler_smoke as S
    cj, cw, sdj, sdw, noises, init = drift_chain_inputs()
    T, F_, HW = DRIFT["T"], DRIFT["F"], DRIFT["HW"]
    kw = dict(standard_fixed_ratio=0.01, w_prob_exp=0.97, w_energy=0.0, design_guidance="standard", coeff_ratio=0.0)
    if steps is not None:
        kw["steps"] = list(reversed(range(T)))[:steps]          # the first `steps` steps of the same schedule
    with torch.no_grad():
        return S.p_sample_loop(S.make_schedule(T, "sigmoid"), lambda x, t: O.unet3d_forward(sdj, cj, x, t),
                               lambda x, t: O.unet3d_forward(sdw, cw, x, t), (1, F_, 6, HW, HW), init, S.rescaler_tensor(),
                               list(noises), **kw)


if __name__ == "__main__":
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    want = sys.argv[1:] or list(FULL_EXTENT) + ["drift_chain"]
    for case in want:
        if case == "drift_chain":
            ref = drift_chain_oracle()
            np.savez_compressed(os.path.join(GOLDEN, "drift_chain.npz"), final=ref.numpy(), after3=drift_chain_oracle(3).numpy())
        else:
            np.savez_compressed(os.path.join(GOLDEN, f"full_extent_{case}.npz"), **full_extent_record(case))
        print("wrote", case, flush=True)
