"""Dataset readers with the reference's on-disk layouts (dataset/data_2d.py:11-141 Jellyfish, :142-209 Smoke) plus synthetic
stand-ins.

`Jellyfish` reads `<root>/{train_data|test_data}/{states,bdry_merged_mask_offsets,bdry_head_thetas}/sim_%06d.npz` and
`normalization_max_min.pkl`, returning the same tuples as the reference for the diffusion train / test splits.

`Smoke` reads `<root>/{train|test/control}/sim_%06d/{Density,Velocity,Control,Smoke}.npy` exactly as the reference
does.  `SyntheticSmoke` fabricates the same tuple when no dataset is mounted (SURVEY.md 8d recipe: a 5x5 block of
density at rows 10..25 / cols 12..52, zero controls)."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

RESCALER = (2, 18, 20, 16, 20, 1)          # data_2d.py:167


class Jellyfish(Dataset):
    """dataset/data_2d.py:11-141.  Test split (`is_train=False`): item i -> (state_0 [3|1, s, s] normalised to [-1, 1],
    thetas_0, bd_mask_offset_0 [3, s, s], sim_id, thetas_gt [steps]); train split: windows of `steps` frames
    (state, bd_mask_offset, thetas, sim_id, time_id).  `for_pipeline=True` (surrogate-simulator pipeline) is out of scope."""

    def __init__(self, dataset, dataset_path, time_steps=40, steps=20, time_interval=1, is_train=True, is_testdata=False,
                 for_pipeline=False, only_vis_pressure=False):
        super().__init__()
        if not dataset.startswith("jellyfish"):
            raise ValueError("dataset must be 'jellyfish*'")
        if for_pipeline:
            raise NotImplementedError("for_pipeline=True feeds the surrogate-simulator evaluation, outside the sampling path")
        self.root, self.steps, self.time_steps, self.time_interval = dataset_path, steps, time_steps, time_interval
        self.is_train, self.only_vis_pressure = is_train, only_vis_pressure
        self.win_size = steps * time_interval
        self.dirname = "train_data" if is_train else "test_data"
        self.n_simu = (100 if is_train else 50) if is_testdata else (1000 if is_train else 100)
        self.time_steps_effective = (time_steps - self.win_size) // time_interval
        import pickle
        with open(os.path.join(self.root, self.dirname, "normalization_max_min.pkl"), "rb") as fh:
            nd = pickle.load(fh)
        for k in ("vx_max", "vx_min", "vy_max", "vy_min", "p_max", "p_min"):
            setattr(self, k, nd[k])

    def __len__(self):
        return self.n_simu * self.time_steps_effective if self.is_train else self.n_simu

    def _npz(self, sub, sim_id, key):
        return np.load(os.path.join(self.root, self.dirname, sub, "sim_{:06d}.npz".format(sim_id)))[key]

    def _unit(self, x, lo, hi):
        """[lo, hi] -> [-1, 1] with clamping (:71-75)."""
        return ((torch.clamp((x - lo) / (hi - lo), 0, 1) - 0.5) * 2).unsqueeze(1)

    def __getitem__(self, idx):
        sim_id, time_id = divmod(idx, self.time_steps_effective) if self.is_train else (idx, 0)
        raw = torch.FloatTensor(self._npz("states", sim_id, "a"))                        # [T, 3, s, s]: vx, vy, p
        p = self._unit(raw[:, 2], self.p_min, self.p_max)
        if self.only_vis_pressure:
            state_full = p
        else:
            state_full = torch.cat((self._unit(raw[:, 0], self.vx_min, self.vx_max),
                                    self._unit(raw[:, 1], self.vy_min, self.vy_max), p), 1)
        state_full[torch.isnan(state_full)] = 0
        bd_full = self._npz("bdry_merged_mask_offsets", sim_id, "a")                     # [T, s, s, 3]
        thetas_full = self._npz("bdry_head_thetas", sim_id, "thetas")                    # [T]
        win = slice(time_id, time_id + self.win_size)
        thetas = torch.FloatTensor(thetas_full[win])
        # The reference zeroes NaNs on a tensor that ALIASES the loaded array (torch.FloatTensor of a float32 view, :80-88), so
        # the frames of the window are cleaned in the array itself -- including frame 0 that the test split returns (:104-109)
        bd_full = np.array(bd_full, dtype=np.float32)
        bd_full[win][np.isnan(bd_full[win])] = 0
        if self.is_train:
            return state_full[win], torch.from_numpy(np.transpose(bd_full[win], (0, 3, 1, 2)).copy()), thetas, sim_id, time_id
        bd_0 = torch.from_numpy(np.transpose(bd_full[0], (2, 0, 1)).copy())
        return state_full[0], thetas[0], bd_0, sim_id, torch.FloatTensor(thetas_full[:self.win_size])


class Smoke(Dataset):
    def __init__(self, dataset_path, time_steps=256, steps=32, all_size=128, size=64, is_train=True):
        super().__init__()
        self.root = dataset_path
        self.steps, self.time_steps = steps, time_steps
        self.time_interval = int(time_steps / steps)
        self.all_size, self.size = all_size, size
        self.space_interval = int(all_size / size)
        self.is_train = is_train
        self.dirname = "train" if is_train else "test"
        self.sub_dirname = "control"
        self.n_simu = 20000 if is_train else 50
        self.RESCALER = torch.tensor(RESCALER).reshape(1, 6, 1, 1)

    def __len__(self):
        return self.n_simu

    def _load(self, sim_id, name):
        parts = [self.root, self.dirname] + ([] if self.is_train else [self.sub_dirname])
        return np.load(os.path.join(*parts, "sim_{:06d}/{}.npy".format(sim_id, name)))

    def __getitem__(self, sim_id):
        d = torch.tensor(self._load(sim_id, "Density"), dtype=torch.float).permute(2, 3, 0, 1)
        v = torch.tensor(self._load(sim_id, "Velocity"), dtype=torch.float).permute(2, 3, 0, 1)
        c = torch.tensor(self._load(sim_id, "Control"), dtype=torch.float).permute(2, 3, 0, 1)
        s = torch.tensor(self._load(sim_id, "Smoke"), dtype=torch.float)
        s = s[:, 1] / s.sum(-1)
        s = s.reshape(1, s.shape[0], 1, 1).expand(1, s.shape[0], self.size, self.size)
        if self.is_train:
            state = torch.cat((d, v, c, s), dim=0)[:, :32]
            return state.permute(1, 0, 2, 3) / self.RESCALER, sim_id
        state = torch.cat((d, v, c, s), dim=0)[:, :256]
        return state.permute(1, 0, 2, 3), sim_id


class SyntheticSmoke(Dataset):
    """Test-split stand-in: (state [256, 6, 64, 64] not rescaled, sim_id) with only the initial density populated."""

    def __init__(self, n_simu=50, size=64, seed=0, is_train=False):
        self.n_simu, self.size, self.is_train = n_simu, size, is_train
        self.RESCALER = torch.tensor(RESCALER).reshape(1, 6, 1, 1)
        g = torch.Generator().manual_seed(seed)
        self.pos = torch.stack((torch.randint(10, 26, (n_simu,), generator=g), torch.randint(12, 53, (n_simu,), generator=g)), 1)

    def __len__(self):
        return self.n_simu

    def __getitem__(self, sim_id):
        frames = 32 if self.is_train else 256
        state = torch.zeros(frames, 6, self.size, self.size)
        r, c = self.pos[sim_id].tolist()
        state[:, 0, r:r + 5, c:c + 5] = 1.0
        if self.is_train:
            return state / self.RESCALER, sim_id
        return state, sim_id
