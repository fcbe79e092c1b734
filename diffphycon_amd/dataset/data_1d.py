"""`Burgers1D` with the reference's contract (/root/reference/dataset/data_1d.py:6-77 over dataset/apps/burgers_h5py.py:18-73,
206-273): sample idx -> (u, f) of one simulation as the [2, 16, 128] image the 2-D denoiser trains on / the flat [21, 128]
layout `get_target` slices (utils.py:1353-1395), rescaled by max |u, f| of the split.

The on-disk format is the authors' HDF5 file `{root}/{dataset}_{split}.h5` with groups `{split}/pde_{nt}-{nx}` (states
[n, nt, nx]) and `{split}/pde_{nt}-{nx}_f` (forces [n, nt - 1, nx]).  h5py is not part of this image: `open_burgers_hdf5` imports
it on demand (and says so when it is missing); everything after the file open -- the part with arithmetic -- works on the
array-backed `BurgersCache`, which is what the tests drive against fixtures recorded from the reference."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset


class BurgersCache:
    """HDF5Dataset (burgers_h5py.py:206-273) after its file open: item idx -> (u_base, u_super, force, x)."""

    def __init__(self, u_super, force, x, ratio_nt=1, ratio_nx=1):
        self.u_super, self.force, self.x = np.asarray(u_super), np.asarray(force), np.asarray(x)
        self.ratio_nt, self.ratio_nx = int(ratio_nt), int(ratio_nx)

    def __len__(self):
        return self.u_super.shape[0]

    def __getitem__(self, idx):
        u_super = self.u_super[idx][::self.ratio_nt][:, :, None]          # (:268)
        u_base = u_super[:, ::self.ratio_nx, :]
        return u_base, u_super, self.force[idx], self.x


def open_burgers_hdf5(path, mode, base_resolution=(11, 128), super_resolution=(11, 128)):
    """The file open of HDF5Dataset.__init__ (:222-252), loading the split into memory."""
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError(f"reading {path} needs h5py, which this image does not ship; use the --synthetic flag, or build a "
                           "BurgersCache from arrays") from e
    with h5py.File(path, "r") as f:
        data = f[mode]
        base = f"pde_{base_resolution[0]}-{base_resolution[1]}"
        sup = f"pde_{super_resolution[0]}-{super_resolution[1]}"
        ratio_nt = int(data[sup].shape[1] / data[base].shape[1])
        ratio_nx = int(data[sup].shape[2] / data[base].shape[2])
        return BurgersCache(data[sup][:], data[sup + "_f"][:], data[base].attrs["x"], ratio_nt, ratio_nx)


class Burgers1D(Dataset):
    def __init__(self, dataset="burgers", input_steps=1, output_steps=10, time_interval=1, is_y_diff=False, split="train",
                 transform=None, pre_transform=None, verbose=False, root_path=None, *, device="cpu", rescaler=None,
                 stack_u_and_f=False, pad_for_2d_conv=False, partially_observed_fill_zero_unobserved=None, dataset_cache=None,
                 **kwargs):
        self.dataset, self.split = dataset, split
        self.root = "data/" if root_path is None else root_path
        self.nx = 128
        self.nt_total, self.nx_total = kwargs["nt_total"], 128
        self.input_steps, self.output_steps, self.time_interval = input_steps, output_steps, time_interval
        assert split in ["train", "test"]
        self.t_cushion_input = input_steps * time_interval if input_steps * time_interval > 1 else 1
        self.t_cushion_output = output_steps * time_interval if output_steps * time_interval > 1 else 1
        if dataset_cache is None:
            if (self.nt_total, self.nx_total) == (11, 128):                # (:61-64)
                path = os.path.join(self.root, "") + f"{dataset}_{split}.h5"
            else:
                path = os.path.join(self.root, "") + f"{dataset}_{split}_nt_{self.nt_total}_nx_{self.nx_total}.h5"
            dataset_cache = open_burgers_hdf5(path, split, (self.nt_total, self.nx), (self.nt_total, self.nx_total))
        self.dataset_cache = dataset_cache
        self.time_stamps = self.nt_total
        self.n_simu = len(self.dataset_cache)
        self.time_stamps_effective = (self.time_stamps - self.t_cushion_input - self.t_cushion_output + time_interval) // time_interval
        self.device = device
        if rescaler is None:
            self.calculate_rescaler()
        else:
            self.rescaler = rescaler
        self.stack_u_and_f = stack_u_and_f
        self.pad_for_2d_conv = pad_for_2d_conv
        self.fill_zero_unobserved = partially_observed_fill_zero_unobserved

    def calculate_rescaler(self):
        """The split's normalisation constant (data_1d.py:31-35): the largest |value| among all states and all forces, as a 0-d
        tensor in the arrays' own precision."""
        peak, kind = 0.0, np.float32
        for _, states, force, _ in self.dataset_cache:
            for arr in (np.asarray(states), np.asarray(force)):
                peak = max(peak, float(np.abs(arr).max()))
                kind = np.promote_types(kind, arr.dtype) if arr.dtype.kind == "f" else kind
        self.rescaler = torch.from_numpy(np.array(peak, dtype=kind))

    def __len__(self):
        return self.n_simu * self.time_stamps_effective

    def __getitem__(self, idx):
        return self.get(idx)

    def get(self, idx, use_normalized=True):
        """One sample (data_1d.py:38-77).  The simulation is `idx // time_stamps_effective` (every time window of a run maps to the
        whole run).  Layouts: stacked = a zero [2, 16, nx] image with the states in plane 0 (rows 0 .. nt) and the forces in plane 1
        (rows 0 .. nt - 1); flat = states followed by forces, [2 nt + 1, nx].  'front_rear_quarter' blanks the unobserved middle
        half of every state row.  Divided by the rescaler unless use_normalized is False."""
        record = self.dataset_cache[idx // self.time_stamps_effective]
        states = torch.as_tensor(np.asarray(record[1]), dtype=torch.float32).reshape(len(record[1]), -1).clone()
        forces = torch.as_tensor(np.asarray(record[2]), dtype=torch.float32).reshape(len(record[2]), -1)
        cells = states.shape[1]
        if self.fill_zero_unobserved == "front_rear_quarter":
            states[:, cells // 4:(3 * cells) // 4] = 0
        elif self.fill_zero_unobserved is not None:
            raise ValueError(f"partially observed mode {self.fill_zero_unobserved!r} is not one of: 'front_rear_quarter'")
        if self.stack_u_and_f != self.pad_for_2d_conv:
            raise AssertionError("stack_u_and_f and pad_for_2d_conv go together (the 2-D denoiser's image) or are both off (get_target)")
        if self.stack_u_and_f:
            sample = torch.zeros(2, 16, cells)
            sample[0, :states.shape[0]] = states
            sample[1, :forces.shape[0]] = forces
        else:
            sample = torch.cat((states, forces))
        return sample / self.rescaler if use_normalized else sample
