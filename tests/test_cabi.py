"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol include/dpc.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from diffphycon_amd import build, _lib
    build.build(verbose=False)                     # hipcc cross-compiles for gfx950 without a GPU
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/dpc.h but not exported"
    # and the ctypes binding table covers exactly the header
    assert sorted(_lib.exported_symbols()) == declared


def test_error_convention_without_gpu():
    from diffphycon_amd import _lib
    L = _lib.lib()
    assert L.dpc_version() >= 100
    rc = L.dpc_unet3d_create(None, None)
    assert rc < 0 and b"null" in L.dpc_last_error()


def test_product_path_refuses_cpu_tensors():
    import torch
    from diffphycon_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.ptr(torch.zeros(4))
    from diffphycon_amd.model.video_diffusion_pytorch.video_diffusion_pytorch_conv3d import Unet3D_with_Conv3D
    m = Unet3D_with_Conv3D(dim=8, dim_mults=(1, 2), channels=6)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 6, 8, 8), torch.zeros(1, dtype=torch.long))
