"""`Burgers1D` (diffphycon_amd/dataset/data_1d.py) against records of the reference's class (dataset/data_1d.py:6-77) on the
same arrays (tools/gen_golden_burgers1d.py): rescaler, stacked / padded image, flat layout, zero-filled unobserved cells."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from diffphycon_amd.dataset.data_1d import Burgers1D, BurgersCache

KW = dict(dataset="burgers", input_steps=1, output_steps=10, time_interval=1, split="test", root_path="data/free_u_f_1e5",
          device="cpu", nt_total=11)


def _cache(g):
    return BurgersCache(g["u"], g["f"], g["x"])


def test_rescaler_and_length():
    g = load_golden("burgers1d")
    d = Burgers1D(**KW, dataset_cache=_cache(g))
    assert float(d.rescaler) == float(g["rescaler"])
    assert len(d) == int(g["n_samples"]) == len(_cache(g)) * d.time_stamps_effective
    # a 0-d tensor in the arrays' own precision, as the reference's torch.cat(...).abs().max() (data_1d.py:31-35)
    assert isinstance(d.rescaler, torch.Tensor) and d.rescaler.dim() == 0
    assert d.rescaler.dtype == torch.from_numpy(np.asarray(g["u"])).dtype


def test_one_definition_per_name():
    """a botched merge once left two `class Burgers1D` in the module, the first one dead code."""
    import ast
    import diffphycon_amd.dataset.data_1d as m
    names = [n.name for n in ast.parse(open(m.__file__).read()).body if isinstance(n, (ast.ClassDef, ast.FunctionDef))]
    assert len(names) == len(set(names)), names


@pytest.mark.parametrize("tag,kw", [("stack", dict(stack_u_and_f=True, pad_for_2d_conv=True)),
                                    ("stack_po", dict(stack_u_and_f=True, pad_for_2d_conv=True,
                                                      partially_observed_fill_zero_unobserved="front_rear_quarter")),
                                    ("flat", dict()),
                                    ("flat_po", dict(partially_observed_fill_zero_unobserved="front_rear_quarter"))])
def test_get_matches_the_reference(tag, kw):
    g = load_golden("burgers1d")
    d = Burgers1D(**KW, dataset_cache=_cache(g), **kw)
    for idx in (0, 3):
        assert np.array_equal(d.get(idx).numpy(), g[f"{tag}:{idx}:norm"])
        assert np.array_equal(d.get(idx, use_normalized=False).numpy(), g[f"{tag}:{idx}:raw"])
    if tag.startswith("stack"):
        assert d.get(0).shape == (2, 16, 128)


def test_get_target_layout_and_missing_h5py():
    g = load_golden("burgers1d")
    d = Burgers1D(**KW, rescaler=1, dataset_cache=_cache(g))
    assert np.array_equal(d.get(1).numpy(), g["target:1"])
    assert d.get(1)[:11].shape == (11, 128) and d.get(1)[11:].shape == (10, 128)        # utils.py:1378-1381
    with pytest.raises(ValueError):
        Burgers1D(**KW, dataset_cache=_cache(g), partially_observed_fill_zero_unobserved="other").get(0)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="h5py"):
            Burgers1D(**KW)                                  # the real file open says what is missing


def test_get_target_reads_through_burgers1d():
    """utils.get_target (:1353-1395) over the array-backed cache: unrescaled states / forces, list and int indices."""
    from diffphycon_amd.utils_burgers import get_target
    g = load_golden("burgers1d")
    u = get_target([1, 3], device=torch.device("cpu"), dataset_cache=_cache(g))
    assert u.shape == (2, 11, 128) and np.array_equal(u[0].numpy(), g["target:1"][:11])
    f = get_target(1, f=True, device=torch.device("cpu"), dataset_cache=_cache(g))
    assert f.shape == (1, 10, 128) and np.array_equal(f[0].numpy(), g["target:1"][11:])
    po = get_target(0, device=torch.device("cpu"), dataset_cache=_cache(g), partially_observed_fill_zero_unobserved="front_rear_quarter")
    assert np.array_equal(po[0].numpy(), g["flat_po:0:raw"][:11])
