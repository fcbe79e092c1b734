"""Build-time pin of the fragment re-load order of the halo convolution kernels (third hazard): in the ISA hipcc emits
for gfx950, no LDS read may overwrite a register that one of the last FOUR issued MFMAs reads as its B operand (the matrix pipe reads
B while the instruction executes; r03 found one wrong fragment element now and then when a second process shared the GPU and the
re-load followed its reader directly).  tools/mfma_war_audit.py does the measurement; hipcc cross-compiles here, no GPU needed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("src,prefix", [("conv3w.hip", "conv3w_kernel"), ("conv3f3c.hip", "conv3f3c_kernel")])
def test_halo_conv_kernels_reload_b_fragments_four_mfmas_behind_their_reader(src, prefix):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc")
    import mfma_war_audit as A
    ks = A.kernels(A.compile_asm(src))
    seen = 0
    for name, lines in ks.items():
        if prefix not in name:
            continue
        best, n_mfma = A.audit(lines)
        if n_mfma == 0:
            continue
        seen += 1
        assert best["B"] is None or best["B"][0] >= 4, (name, best["B"])
    assert seen >= 2, "no instantiation of the kernel found in the ISA"
