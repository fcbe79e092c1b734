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


def test_build_stamp_and_the_refusal_of_a_library_with_packed_fp32_ops(tmp_path):
    """correctness on a shared GPU rests on ONE compile flag (no v_pk_{mul,add,fma}_f32 in the device code, DESIGN.md
    6.2).  build.py measures the linked code objects and stamps the count (dpc_build_info); `_lib.lib()` refuses a library whose stamp is
    not clean unless DPC_ALLOW_PACKED_FP32=1 (deliberately NOT DPC_DEBUG, which the test suite sets).  Here: the product library is clean; the scanner FINDS the instruction in a kernel compiled without
    the flag (so a clean count means something); the loader check refuses such a stamp."""
    import subprocess
    import warnings
    from diffphycon_amd import build, _lib
    info = _lib.lib().dpc_build_info().decode()
    assert info.startswith("packed_fp32_insts=0;") and "-packed-fp32-ops" in info
    assert build.scan_packed_fp32(_lib.LIB_PATH)[0] == 0
    src = tmp_path / "pk.hip"
    src.write_text("#include <hip/hip_runtime.h>\n"
                   "typedef float f2 __attribute__((ext_vector_type(2)));\n"
                   "__global__ void k(const f2* a, const f2* b, f2* c) { int i = threadIdx.x; c[i] = a[i] * b[i] + c[i]; }\n")
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("needs hipcc")
    with_pk, without = str(tmp_path / "pk.so"), str(tmp_path / "nopk.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", str(src), "-o", with_pk])
    subprocess.check_call([hipcc, *build.FLAGS, "-shared", str(src), "-o", without], stderr=subprocess.DEVNULL)
    assert build.scan_packed_fp32(with_pk)[0] >= 1
    assert build.scan_packed_fp32(without)[0] == 0

    class Fake:
        def __init__(self, text):
            self.dpc_build_info = lambda: text.encode()
    _lib._check_build(Fake("packed_fp32_insts=0;code_objects=1;flags="))
    with pytest.raises(RuntimeError, match="packed fp32"):
        _lib._check_build(Fake("packed_fp32_insts=3;code_objects=1;flags="))
    os.environ["DPC_ALLOW_PACKED_FP32"] = "1"
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            _lib._check_build(Fake("packed_fp32_insts=3;code_objects=1;flags="))
            assert w and "DPC_ALLOW_PACKED_FP32" in str(w[0].message)
    finally:
        del os.environ["DPC_ALLOW_PACKED_FP32"]
