"""Pins oracle/unet3d.py against the reference's own outputs (fixtures made by tools/gen_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import unet3d as O
from conftest import load_golden


def _sd(g, prefix="w:"):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def test_relpos_bucket_bit_exact():
    g = load_golden("relpos_bucket")
    for n in (4, 20, 32, 64):
        tab = O.relative_position_bucket(n).numpy()
        assert tab.dtype == np.int64
        assert np.array_equal(tab, g[f"n{n}"])


@pytest.mark.parametrize("tag", ["joint", "w", "wide"])
def test_unet3d_forward_matches_reference(tag):
    g = load_golden(f"unet3d_{tag}")
    cfg = O.Unet3DConfig(dim=int(g["dim"]), dim_mults=tuple(int(v) for v in g["dim_mults"]), channels=int(g["channels"]))
    sd = _sd(g)
    # the synthetic-weight recipe must enumerate exactly the reference's parameters
    shapes = dict(O.param_shapes(cfg))
    assert set(shapes) == set(sd)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    taps = {}
    y = O.unet3d_forward(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), taps=taps)
    ref = torch.from_numpy(g["y"])
    # tolerance: per SURVEY 8(d) full forward rel 1e-4; same torch CPU ops so we are far inside it
    assert torch.allclose(y, ref, rtol=1e-5, atol=2e-6), (y - ref).abs().max()
    for k in g.files:
        if k.startswith("tap:"):
            name = k[4:]
            assert name in taps, name
            r = torch.from_numpy(g[k])
            assert torch.allclose(taps[name], r, rtol=1e-5, atol=2e-6), (name, (taps[name] - r).abs().max())


def test_synthetic_state_dict_is_deterministic():
    cfg = O.Unet3DConfig(dim=8, dim_mults=(1, 2), channels=6)
    a = O.synthetic_state_dict(cfg, seed=5)
    b = O.synthetic_state_dict(cfg, seed=5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    x = torch.randn(1, 2, 6, 8, 8)
    y = O.unet3d_forward(a, cfg, x, torch.tensor([3]))
    assert y.shape == x.shape and torch.isfinite(y).all()


def test_product_relative_position_bucket_matches_reference_table():
    """The PRODUCT's host-side T5 bucket table (the one the HIP attention kernels consume) against the reference's
    RelativePositionBias._relative_position_bucket (...conv3d.py:86-104) for 4, 20, 32 and 64 frames: integer work, bit-exact
    (the value at distance 16 depends on fp32 log rounding)."""
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import _relative_position_bucket
    g = load_golden("relpos_bucket")
    for n in (4, 20, 32, 64):
        ref = g[f"n{n}"]
        got = _relative_position_bucket(n).numpy()
        assert got.dtype == ref.dtype and np.array_equal(got, ref), n


def test_r04_fixtures_are_what_the_oracle_computes():
    """tests/golden/full_extent_*.npz and drift_chain.npz (tools/gen_golden_r04.py) hold ORACLE outputs so that the GPU suite does not
    spend minutes of host time on them per box.  Re-derived here for the cheapest members: the s64_prior full-extent forward (every
    recorded sample of every tap, the output, shapes and maxima) and the first three steps of the 100-step drift chain.  Same torch
    CPU kernels on another host may differ in the last bits (oneDNN code paths): 2e-6 of the tensor's range."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_golden_r04 as G
    g = load_golden("full_extent_s64_prior")
    rec = G.full_extent_record("s64_prior")
    assert list(rec["names"]) == list(g["names"]) and len(rec["names"]) >= 30
    for name in g["names"]:
        assert np.array_equal(rec[f"shape:{name}"], g[f"shape:{name}"])
        scale = float(g[f"absmax:{name}"])
        assert abs(float(rec[f"absmax:{name}"]) - scale) <= 2e-6 * scale
        assert np.abs(rec[f"values:{name}"] - g[f"values:{name}"]).max() <= 2e-6 * scale, name
    d = load_golden("drift_chain")
    after3 = G.drift_chain_oracle(3).numpy()
    assert np.abs(after3 - d["after3"]).max() <= 2e-6 * np.abs(d["after3"]).max()
    assert d["final"].shape == d["after3"].shape and np.isfinite(d["final"]).all()


def test_s128_prior_fixture_is_what_the_oracle_computes():
    """r05: tests/golden/full_extent_s128_prior.npz (the 2-channel prior net at the S128 extent, 64 x 128 x 128: VERDICT r04 missing #3)
    re-derived from the oracle: shapes, maxima, the 4096 random samples AND the deterministic tile-edge lines of every tap
    (gen_golden_r04.edge_index).  ~1.5 minutes on 8 cores."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_golden_r04 as G
    g = load_golden("full_extent_s128_prior")
    rec = G.full_extent_record("s128_prior")
    assert list(rec["names"]) == list(g["names"]) and len(rec["names"]) == 37
    for k, name in enumerate(g["names"]):
        assert np.array_equal(rec[f"shape:{name}"], g[f"shape:{name}"])
        shape = tuple(int(v) for v in g[f"shape:{name}"])
        n_edge = len(G.edge_index(shape, 2 if name == "y" else 1))
        assert len(g[f"values:{name}"]) == min(G.NSAMP, int(np.prod(shape))) + n_edge and (n_edge > 4000 or len(shape) != 5)
        scale = float(g[f"absmax:{name}"])
        assert np.abs(rec[f"values:{name}"] - g[f"values:{name}"]).max() <= 2e-6 * scale, name


def test_edge_index_touches_every_tile_and_every_seam():
    """tools/gen_golden_r04.edge_index (the deterministic positions of the full-extent fixtures): along each of the frame / row / column
    axes the lines cross EVERY 4 x 8 x 8 output tile of the tensor and both sides of every seam between two tiles, for taps
    [1, C, F, H, W] and for the output layout [1, F, C, H, W]; positions are unique and inside the tensor."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_golden_r04 as G
    for shape, ax in (((1, 64, 32, 64, 64), 1), ((1, 128, 64, 32, 32), 1), ((1, 256, 20, 16, 16), 1), ((1, 32, 6, 64, 64), 2)):
        idx = G.edge_index(shape, ax)
        assert len(idx) == len(set(idx.tolist())) and idx.min() >= 0 and idx.max() < int(np.prod(shape))
        c, f, h, w = np.unravel_index(idx, shape)[1:] if ax == 1 else [np.unravel_index(idx, shape)[i] for i in (2, 1, 3, 4)]
        C_, F_, H_, W_ = (shape[1], shape[2], shape[3], shape[4]) if ax == 1 else (shape[2], shape[1], shape[3], shape[4])
        assert set(c.tolist()) == {0, C_ // 2, C_ - 1}
        tiles = {(ff // 4, hh // 8, ww // 8) for ff, hh, ww in zip(f.tolist(), h.tolist(), w.tolist())}
        # every tile index occurs along each axis (a line along an axis visits all tiles of that axis) ...
        assert {t[0] for t in tiles} == set(range((F_ + 3) // 4)) and {t[1] for t in tiles} == set(range(H_ // 8)) and {t[2] for t in tiles} == set(range(W_ // 8))
        # ... and so do both sides of every seam (last point of a tile, first point of the next)
        assert set(range(F_)) <= set(f.tolist()) and set(range(H_)) <= set(h.tolist()) and set(range(W_)) <= set(w.tolist())
    assert len(G.edge_index((1, 256), 1)) == 0
    assert len(G.sample_index(41, 3, 8388608, (1, 64, 32, 64, 64), 1)) == G.NSAMP + len(G.edge_index((1, 64, 32, 64, 64), 1))


def test_oracle_at_full_width_matches_the_reference_fixture():
    """r06: tests/golden/unet3d_dim64.npz is the REFERENCE's forward at dim 64, mults (1, 2, 4) (tools/gen_golden_r06.py) on the
    synthetic weights of the seed stored in the file -- the oracle is pinned to the reference at the width it is used as the
    checker for (8 channels per GroupNorm group, 64 / 128 / 256-wide attention), not only at dim 8 / 16."""
    g = load_golden("unet3d_dim64")
    cfg = O.Unet3DConfig(dim=64, dim_mults=(1, 2, 4), channels=6)
    sd = O.synthetic_state_dict(cfg, seed=int(g["seed"]))
    taps = {}
    with torch.no_grad():
        y = O.unet3d_forward(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), taps=taps)
    ref = torch.from_numpy(g["y"])
    assert (y - ref).abs().max() <= 1e-5 * ref.abs().max(), (y - ref).abs().max()
    n = 0
    for k in g.files:
        if k.startswith("tap:"):
            r = torch.from_numpy(g[k])
            assert (taps[k[4:]] - r).abs().max() <= 1e-5 * r.abs().max() + 1e-6, k
            n += 1
    assert n == 6
