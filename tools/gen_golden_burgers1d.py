"""Generate tests/golden/burgers1d.npz by importing the reference's `Burgers1D` (dataset/data_1d.py:6-77) over an ARRAY-backed
stand-in for its HDF5 cache (dataset/apps/burgers_h5py.py:206-273 -- h5py is not installed in this image; the stand-in returns
what HDF5Dataset.__getitem__ returns for the same arrays).  Build container only.

    python tools/gen_golden_burgers1d.py

Recorded: the seeded (u, f) arrays, calculate_rescaler's value and `get(idx)` for every option the inference / train scripts use
(stack_u_and_f + pad_for_2d_conv, partially_observed_fill_zero_unobserved, use_normalized) plus the flat layout get_target reads."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402


class _GeoDataset:                       # torch_geometric.data.Dataset stand-in: Burgers.__init__ calls it with (root, transform, pre_transform)
    def __init__(self, *a, **k):
        pass


sys.modules["torch_geometric.data"].Dataset = _GeoDataset
from dataset.apps import burgers_h5py as BH  # noqa: E402

RNG = np.random.default_rng(7)
N_SIMU = 5
U = (RNG.standard_normal((N_SIMU, 11, 128)) * 1.7).astype(np.float64)
Fo = (RNG.standard_normal((N_SIMU, 10, 128)) * 2.3).astype(np.float64)
X = np.linspace(0, 1, 128)


class ArrayH5:
    """HDF5Dataset (:206-273) over arrays: data['pde_11-128'] = U, data['pde_11-128_f'] = Fo, resolution ratios 1."""

    def __init__(self, path, mode, base_resolution=None, super_resolution=None, load_all=False, uniform_sample=-1):
        self.ratio_nt = self.ratio_nx = 1
        self.x = X

    def __len__(self):
        return U.shape[0]

    def __getitem__(self, idx):
        u_super = U[idx][::self.ratio_nt][:, :, None]
        u_base = u_super[:, ::self.ratio_nx, :]
        return u_base, u_super, Fo[idx], self.x


BH.HDF5Dataset = ArrayH5
from dataset.data_1d import Burgers1D  # noqa: E402

KW = dict(dataset="burgers", input_steps=1, output_steps=10, time_interval=1, is_y_diff=False, split="test", transform=None,
          pre_transform=None, verbose=False, root_path="data/free_u_f_1e5", device="cpu", nt_total=11)


def main():
    out = dict(u=U, f=Fo, x=X)
    d = Burgers1D(**KW)                                              # rescaler=None -> calculate_rescaler
    out["rescaler"] = np.asarray(d.rescaler)
    out["n_samples"] = np.asarray(d.n_simu * d.time_stamps_effective)
    for tag, kw in (("stack", dict(stack_u_and_f=True, pad_for_2d_conv=True)),
                    ("stack_po", dict(stack_u_and_f=True, pad_for_2d_conv=True, partially_observed_fill_zero_unobserved="front_rear_quarter")),
                    ("flat", dict()),
                    ("flat_po", dict(partially_observed_fill_zero_unobserved="front_rear_quarter"))):
        d = Burgers1D(**KW, **kw)
        for idx in (0, 3):
            out[f"{tag}:{idx}:norm"] = d.get(idx).numpy()
            out[f"{tag}:{idx}:raw"] = d.get(idx, use_normalized=False).numpy()
    d = Burgers1D(**KW, rescaler=1)                                  # get_target's construction (utils.py:1353-1370)
    out["target:1"] = d.get(1).numpy()
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "burgers1d.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB, rescaler", float(out["rescaler"]))


if __name__ == "__main__":
    main()
