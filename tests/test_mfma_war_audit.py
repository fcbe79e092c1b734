"""Build-time pin of the MFMA operand re-load order (third hazard): in the ISA hipcc emits for gfx950, no LDS read may
overwrite a register that one of the last FOUR issued MFMAs reads as its B operand (the matrix pipe reads B while the instruction
executes; r03 found one wrong fragment element now and then when a second process shared the GPU and a re-load followed its reader
directly).  r03 pinned the halo convolution kernels; since r04 EVERY MFMA kernel of the library is covered -- the implicit-GEMM family,
the stems, the fused attention kernels, the dense / linear attention cores and the training kernels (common.h: mfma_keep /
mfma_order_point / mfma_drain are the source-level tools).  tools/mfma_war_audit.py does the measurement on every control-flow path;
hipcc cross-compiles here, no GPU needed.  An MFMA whose result was consumed before the load has completed and is not counted."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# file -> (kernel-name fragments that must be found, minimal number of instantiations)
FILES = {
    "conv3w.hip": ("conv3w_kernel", 4),
    "conv3w4.hip": ("conv3w4_kernel", 4),
    "conv3f3c.hip": ("conv3f3c_kernel", 4),
    "igemm6.hip": ("igemm3_kernel", 4),
    "igemm_panel.hip": ("igemm3p_kernel", 24),
    "igemm_tile.hip": ("igemm3t_kernel", 2),
    "igemm_wide.hip": ("igemm3w_kernel", 4),
    "igemm_img.hip": ("igemm3i_kernel", 4),
    "stem7x6.hip": ("stem7", 5),
    "tattn3.hip": ("tattn3", 9),
    "lattn3.hip": ("lattn3_kernel", 2),
    "attn.hip": ("attention_kernel", 1),
    "unet2d.hip": (None, 0),
    "wgrad3.hip": ("wgrad3_kernel", 3),
    "train.hip": ("tattn_bwd_mfma_kernel", 1),
    "surr.hip": ("linattn_bwd", 2),
}


@pytest.mark.parametrize("src", sorted(FILES))
def test_no_lds_read_lands_in_a_b_operand_of_the_last_four_mfmas(src):
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs hipcc")
    import mfma_war_audit as A
    prefix, at_least = FILES[src]
    asm = A.compile_asm(src)
    # r04: the library is built WITHOUT packed fp32 VALU ops (diffphycon_amd/build.py: -packed-fp32-ops) -- with them a co-resident second
    # kernel changed results in a few per cent of launches (tools/det_ops2.py); the audit compiles with the product flags
    import re
    assert not re.search(r"\bv_pk_(mul|add|fma)_f32\b", asm), f"{src}: packed fp32 VALU instructions in the ISA (build flags lost?)"
    ks = A.kernels(asm)
    seen = 0
    for name, lines in ks.items():
        best, n_mfma = A.audit(lines)
        if n_mfma == 0:
            continue
        if prefix and prefix in name:
            seen += 1
        assert best["B"] is None or best["B"][0] >= 4, (name, best["B"])
    assert seen >= at_least, f"{src}: {seen} instantiations of {prefix} found in the ISA, expected >= {at_least}"


def test_the_audit_sees_the_hazard_it_is_looking_for():
    """A synthetic kernel text with the r03 pattern (a ds_read into the B registers right behind its reader; a second one four MFMAs
    later; a third behind a consumed accumulator) -- the tool must report 0, then 4, then nothing."""
    import mfma_war_audit as A
    mf = lambda d, a, b: f"v_mfma_f32_32x32x16_f16 v[{d}:{d + 15}], v[{a}:{a + 3}], v[{b}:{b + 3}], v[{d}:{d + 15}]"
    bad = [mf(0, 100, 104), "ds_read_b128 v[104:107], v1", "s_endpgm"]
    assert A.audit(bad)[0]["B"][0] == 0
    ok = [mf(0, 100, 104)] + [mf(16 * k, 100, 108) for k in range(1, 5)] + ["ds_read_b128 v[104:107], v1", "s_endpgm"]
    assert A.audit(ok)[0]["B"][0] == 4
    consumed = [mf(0, 100, 104), "v_mov_b32_e32 v200, v3", "ds_read_b128 v[104:107], v1", "s_endpgm"]
    assert A.audit(consumed)[0]["B"] is None
    loop = [".LBB0_1:", "ds_read_b128 v[104:107], v1", mf(0, 100, 104), "s_cbranch_scc1 .LBB0_1", "s_endpgm"]
    assert A.audit(loop)[0]["B"][0] == 0                     # (found through the back edge)
