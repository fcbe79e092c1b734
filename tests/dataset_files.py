"""Writers of tiny on-disk datasets in the layouts the reference's readers expect (dataset/data_2d.py:11-141 Jellyfish,
:142-209 Smoke), from a NumPy PCG64 seed: used by tools/gen_golden_r02.py (which records what the REFERENCE's readers return
for them) and by tests/test_datasets.py (which rewrites the same files and checks the product's readers against that record)."""
import os

import numpy as np


def write_jellyfish_files(root, seed, n_sims=3, s=62):
    """Tiny `test_data` tree in the on-disk layout dataset/data_2d.py:11-141 reads."""
    rng = np.random.default_rng(seed)
    d = os.path.join(root, "test_data")
    for sub in ("states", "bdry_merged_mask_offsets", "bdry_head_thetas"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    import pickle
    norm = dict(vx_max=1.5, vx_min=-1.25, vy_max=2.0, vy_min=-1.0, p_max=3.0, p_min=-2.5)
    pickle.dump(norm, open(os.path.join(d, "normalization_max_min.pkl"), "wb"))
    for i in range(n_sims):
        st = (rng.standard_normal((40, 3, s, s)) * 1.2).astype(np.float32)
        st[3, 1, 5, 7] = np.nan                                        # the reader zeroes NaNs
        np.savez(os.path.join(d, "states", f"sim_{i:06d}.npz"), a=st)
        bd = rng.standard_normal((40, s, s, 3)).astype(np.float32)
        bd[0, 2, 3, 1] = np.nan
        np.savez(os.path.join(d, "bdry_merged_mask_offsets", f"sim_{i:06d}.npz"), a=bd)
        np.savez(os.path.join(d, "bdry_head_thetas", f"sim_{i:06d}.npz"), thetas=rng.uniform(0.2, 0.9, 40).astype(np.float32))


def write_smoke_files(root, seed, n_sims=2, nt=256, n=64):
    """Tiny `test/control` tree in the layout dataset/data_2d.py:142-209 reads (Density/Velocity/Control/Smoke .npy)."""
    rng = np.random.default_rng(seed)
    for i in range(n_sims):
        d = os.path.join(root, "test", "control", f"sim_{i:06d}")
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, "Density.npy"), rng.uniform(0, 1, (n, n, 1, nt + 1)).astype(np.float32))      # [H, W, C, T]
        np.save(os.path.join(d, "Velocity.npy"), rng.standard_normal((n, n, 2, nt + 1)).astype(np.float32))
        np.save(os.path.join(d, "Control.npy"), rng.standard_normal((n, n, 2, nt + 1)).astype(np.float32))
        np.save(os.path.join(d, "Smoke.npy"), rng.uniform(0.1, 1, (nt + 1, 8)).astype(np.float32))
