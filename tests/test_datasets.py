"""Row f-3: the product's dataset readers against the REFERENCE's readers (dataset/data_2d.py) on the same tiny on-disk
datasets: tests/dataset_files.py writes them from a seed, tests/golden/datasets.npz holds what the reference returned."""
import numpy as np
import torch

from conftest import load_golden
from dataset_files import write_jellyfish_files, write_smoke_files


def _same(a, b):
    a = a.numpy() if torch.is_tensor(a) else np.asarray(a)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def test_jellyfish_test_split_reader(tmp_path):
    from diffphycon_amd.dataset.data_2d import Jellyfish
    g = load_golden("datasets")
    write_jellyfish_files(str(tmp_path), int(g["seed_jelly"]))
    for tag, ovp in (("full", False), ("pob", True)):
        ds = Jellyfish(dataset="jellyfish", dataset_path=str(tmp_path), time_steps=40, steps=20, time_interval=1,
                       is_train=False, is_testdata=True, only_vis_pressure=ovp)
        assert len(ds) == int(g[f"jelly:{tag}:len"])
        for i in (0, 2):
            state_0, thetas_0, bd_0, sim_id, thetas_gt = ds[i]
            assert _same(state_0, g[f"jelly:{tag}:{i}:state_0"])
            assert _same(thetas_0, g[f"jelly:{tag}:{i}:thetas_0"])
            assert _same(bd_0, g[f"jelly:{tag}:{i}:bd_0"])
            assert sim_id == int(g[f"jelly:{tag}:{i}:sim_id"])
            assert _same(thetas_gt, g[f"jelly:{tag}:{i}:thetas_gt"])


def test_smoke_test_split_reader(tmp_path):
    from diffphycon_amd.dataset.data_2d import Smoke
    g = load_golden("datasets")
    write_smoke_files(str(tmp_path), int(g["seed_smoke"]))
    ds = Smoke(dataset_path=str(tmp_path), is_train=False)
    assert len(ds) == int(g["smoke:len"])
    state, sim_id = ds[1]
    assert sim_id == int(g["smoke:1:sim_id"]) and tuple(state.shape) == (256, 6, 64, 64)
    assert _same(state[[0, 1, 8, 255]], g["smoke:1:state_frames"])
    assert np.array_equal(state.double().mean((2, 3)).numpy(), g["smoke:1:state_mean"])
    assert _same(ds.RESCALER, g["smoke:RESCALER"])


def test_jellyfish_pipeline_pads_62_to_64():
    """inference_2d_jellyfish.py:328-340: 62x62 fields are centred in a zero 64x64 frame."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "inference"))
    try:
        mod = importlib.import_module("inference_2d_jellyfish")
    finally:
        sys.path.remove(os.path.join(root, "inference"))
    s0, b0 = torch.randn(2, 3, 62, 62), torch.randn(2, 3, 62, 62)
    ps, pb = mod.pad_data(s0, b0, 64)
    assert ps.shape == pb.shape == (2, 3, 64, 64)
    assert torch.equal(ps[:, :, 1:-1, 1:-1], s0) and torch.equal(pb[:, :, 1:-1, 1:-1], b0)
    assert ps[:, :, 0].abs().sum() == 0 and pb[:, :, :, -1].abs().sum() == 0
    same = torch.randn(2, 3, 64, 64)
    assert mod.pad_data(same, same, 64)[0] is same
