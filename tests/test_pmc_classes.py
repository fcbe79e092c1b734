"""Host logic behind bench.py's `roofline.traffic`: tools/pmc_summary.py maps rocprofv3 kernel symbols to the profile classes of
libdpc's ProfScope (the classes bench.py's breakdown uses), and stamps every class with a hash of its kernel sources so that stale
counters are reported as `traffic: null` instead of silently surviving a kernel change.  Symbols as rocprofv3 prints them in
profiles/r03_k_kernel_stats.csv."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
P = importlib.util.module_from_spec(spec)
spec.loader.exec_module(P)

CASES = {
    "void dpc::conv3w_kernel<false, 64>(dpc::Conv3hParams)": "conv3x6_bn64",
    "void dpc::conv3w_kernel<true, 128>(dpc::Conv3hParams)": "conv3x6_bn128",
    "void dpc::conv3f3c_kernel<64, 1>(dpc::Conv3hParams)": "conv3x6_bn64",
    "void dpc::igemm3_kernel<64, true>(dpc::IgemmParams, unsigned char const*)": "igemm_bn64",
    "void dpc::igemm3_kernel<128, false>(dpc::IgemmParams, unsigned char const*)": "igemm_bn128",
    # row panels: N = 32 * NT * WN -- only <.., 1, 1, 2, 2, ..> is the 64-column form
    "void dpc::igemm3p_kernel<4, 1, 1, 2, 2, false, 64>(dpc::IgemmParams, unsigned char const*)": "igemm_bn64",
    "void dpc::igemm3p_kernel<8, 1, 1, 2, 2, true, 64>(dpc::IgemmParams, unsigned char const*)": "igemm_bn64",
    "void dpc::igemm3p_kernel<8, 2, 3, 1, 4, false, 64>(dpc::IgemmParams, unsigned char const*)": "igemm_bn128",
    "void dpc::igemm3p_kernel<16, 1, 1, 1, 4, true, 32>(dpc::IgemmParams, unsigned char const*)": "igemm_bn128",
    "void dpc::igemm3p_kernel<4, 1, 2, 2, 2, false, 64>(dpc::IgemmParams, unsigned char const*)": "igemm_bn128",
    "void dpc::igemm3t_kernel<2, 4, 2, 1, 2, 2, 128>(dpc::IgemmParams, unsigned char const*)": "igemm_bn64",
    "void dpc::igemm3t_kernel<4, 4, 1, 2, 2, 2, 64>(dpc::IgemmParams, unsigned char const*)": "igemm_bn128",
    "void dpc::igemm3i_kernel<false, 9>(dpc::IgemmParams, unsigned char const*)": "igemm_bn128",
    "void dpc::igemm3w_kernel<false, 64>(dpc::IgemmParams, unsigned char const*)": "igemm_bn64",
    "void dpc::igemm3w_kernel<true, 128>(dpc::IgemmParams, unsigned char const*)": "igemm_bn128",
    "void dpc::conv3w4_kernel<true, 64>(dpc::Conv3hParams)": "conv3x6_bn64",
    "void dpc::conv3w4_kernel<false, 128>(dpc::Conv3hParams)": "conv3x6_bn128",
    "void dpc::stem7x6_kernel<true, 8>(dpc::StemParams, unsigned char const*)": "stem_gather",
    "void dpc::stem7p_kernel<3>(dpc::StemParams, unsigned char const*)": "stem_gather",
    "void dpc::tattn3_kernel<64, true, 0>(dpc::TattnParams, unsigned char const*, unsigned char const*, float*)": "temporal_attention_fused",
    "void dpc::lattn3_kernel<64>(dpc::LattnParams, unsigned char const*, unsigned char const*)": "linear_attention_fused",
    "dpc::gn_apply_kernel(float const*, float*, float const*, float const*, float const*, float const*, float const*, long long, int, int, int, int*)": "groupnorm_silu",
    "dpc::attention_kernel(dpc::AttnParams, long long, int)": "attention_core",
}


def test_kernel_symbols_map_to_profile_classes():
    for sym, cls in CASES.items():
        assert P.classify(sym) == cls, (sym, P.classify(sym))


def test_every_stamped_class_names_existing_sources():
    for cls, files in P.SOURCES.items():
        for f in files:
            assert os.path.exists(os.path.join(ROOT, "diffphycon_amd", "csrc", f)), (cls, f)
        assert P.source_stamp(cls) is not None


def test_committed_traffic_matches_the_tree():
    """profiles/pmc_traffic.json must carry the stamp of the kernel sources in the tree for the class bench.py reports as the
    roofline kernel (conv3x6_bn64) -- otherwise the default bench line prints traffic: null."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert d["conv3x6_bn64"]["kernel_source_sha16"] == P.source_stamp("conv3x6_bn64")
