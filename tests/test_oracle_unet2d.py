"""Pins oracle/unet2d.py against the reference's Unet2D (fixtures unet2d_{a,b}.npz from tools/gen_golden.py unet2d)."""
import numpy as np
import pytest
import torch

from oracle import unet2d as U
from conftest import load_golden


def _load(tag):
    g = load_golden("unet2d_" + tag)
    cfg = U.Unet2DConfig(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=2, out_dim=2,
                         resnet_block_groups=int(g["groups"]))
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w:")}
    return g, cfg, sd


@pytest.mark.parametrize("tag", ["a", "b"])
def test_forward_and_taps_match_reference(tag):
    g, cfg, sd = _load(tag)
    assert sorted(sd) == sorted(n for n, _, _ in U.param_shapes(cfg))
    for n, shape, _ in U.param_shapes(cfg):
        assert tuple(sd[n].shape) == tuple(shape), n
    taps = {}
    with torch.no_grad():
        y = U.unet2d_forward(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), taps)
    ref = torch.from_numpy(g["y"])
    assert (y - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-6          # per-block rel 1e-5 / abs 1e-6
    n_taps = 0
    for k in g.files:
        if k.startswith("tap:"):
            r = torch.from_numpy(g[k])
            assert (taps[k[4:]] - r).abs().max() <= 1e-5 * r.abs().max() + 1e-6, k
            n_taps += 1
    assert n_taps >= 14


def test_synthetic_state_dict_shapes():
    cfg = U.Unet2DConfig(dim=64, dim_mults=(1, 2, 4, 8, 16))
    sd = U.synthetic_state_dict(cfg)
    n = sum(v.numel() for v in sd.values())
    assert abs(n - 135.8e6) / 135.8e6 < 0.01                      # POPC joint model 135.8 M parameters
    cfg = U.Unet2DConfig(dim=64, dim_mults=(1, 2, 4))
    assert abs(sum(v.numel() for v in U.synthetic_state_dict(cfg).values()) - 9.9e6) / 9.9e6 < 0.02


def test_oracle_at_the_popc_width_matches_the_reference_fixture():
    """r06: tests/golden/unet2d_popc.npz = the REFERENCE's Unet2D at dim 64, mults (1, 2, 4, 8, 16), one GroupNorm group
    (tools/gen_golden_r06.py) on the synthetic weights of the stored seed."""
    g = load_golden("unet2d_popc")
    mults = tuple(int(v) for v in g["dim_mults"])
    cfg = U.Unet2DConfig(dim=64, dim_mults=mults, channels=2, out_dim=2, resnet_block_groups=1)
    sd = U.synthetic_state_dict(cfg, seed=int(g["seed"]))
    taps = {}
    with torch.no_grad():
        y = U.unet2d_forward(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), taps)
    ref = torch.from_numpy(g["y"])
    assert (y - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-6
    for k in ("init_conv", "mid_block2", "final_res_block"):
        r = torch.from_numpy(g["tap:" + k])
        assert (taps[k] - r).abs().max() <= 1e-5 * r.abs().max() + 1e-6, k
