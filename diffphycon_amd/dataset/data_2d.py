"""Dataset readers with the reference's on-disk layout (dataset/data_2d.py:142-209) plus a synthetic stand-in.

`Smoke` reads `<root>/{train|test/control}/sim_%06d/{Density,Velocity,Control,Smoke}.npy` exactly as the reference
does.  `SyntheticSmoke` fabricates the same tuple when no dataset is mounted (SURVEY.md 8d recipe: a 5x5 block of
density at rows 10..25 / cols 12..52, zero controls)."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

RESCALER = (2, 18, 20, 16, 20, 1)          # data_2d.py:167


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
