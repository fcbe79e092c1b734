// Row-panel form of the f16x3 implicit GEMM (igemm6.hip: igemm3_kernel) for the 1-tap operators with a short reduction: the
// 1x1x1 projections of the attention blocks (to_qkv behind a channel LayerNorm, to_out + residual) and the res_conv of a ResnetBlock
// (video_diffusion_pytorch_conv3d.py:159-163, 206-230, 232-257; burgers_1d/unet.py the same), K <= 256, N in {64, 128, 256, 384}; and
// (TAPS form) for few-tap operators whose im2col row is <= 256 wide: the 2 x 2-tap parity classes of ConvTranspose3d at 64 channels.
//
// Why: igemm3's 128 x 64 tiles walk K in 32-channel chunks -- each chunk is one dependent round trip to memory behind a barrier
// (r03 PMC: waves 56-69 % of their cycles in s_waitcnt, 5 us per chunk), every 128-byte piece of an activation row is fetched in a
// different iteration (DRAM sees 128-byte accesses at the row pitch), and the rows are re-read once per 64-column tile.  Here a
// 256-thread workgroup owns 64 rows and ALL N columns:
//   1. the whole [64][K] fp32 panel is requested at once (up to 16 dwordx4 loads per thread in flight, whole rows in address order:
//      64 KB per workgroup, two workgroups per CU), LayerNorm-normalised, split into two fp16 planes and parked in LDS;
//   2. the K loop then runs from LDS without a barrier: A fragments by ds_read_b128 (row pitch 4 K + 16 bytes: conflict-free),
//      weight fragments straight from L2 in the packed fragment order through a ring of four k-step register sets (prefetch
//      distance three k-steps; a set is re-loaded only after twelve or more younger MFMAs have been issued -- the matrix pipe reads
//      its B operand while it executes);
//   3. igemm3's vector epilogue (DPP 4 x 4 transpose, bias, residual, fused GroupNorm-apply residual, range sentinel).
// Activations are read from HBM exactly once and rows are written once; weights (<= 384 KB) stay in L2.
// Arithmetic, operand scales and the order of the partial products per output element are igemm3_kernel's.
#include <algorithm>

#include "common.h"
#include "f16x3.h"
#include "igemm_epilogue.h"

namespace dpc {

namespace gpn {
constexpr int WROW = 128;                   // packed weight bytes per output channel per 32-channel chunk
}  // namespace gpn

typedef _Float16 f16x8_p __attribute__((ext_vector_type(8)));

// KC: 32-channel chunks of the (zero-padded) reduction; wave grid WM x WN (4 waves), wave tile MT x NT blocks of 32 x 32
// BM: panel rows (64; 32 for the 16-chunk im2col rows of the 128-channel ConvTranspose classes: 66 KB of LDS either way)
template <int KC, int MT, int NT, int WM, int WN, bool TAPS, int BM>
__global__ __launch_bounds__(256, 2) void igemm3p_kernel(IgemmParams p, const unsigned char* __restrict__ wp6) {
    h3::hw_sat_enable();                               // (f16x3.h: operand conversions saturate in hardware)
    using namespace gpn;
    static_assert(WM * WN == 4 && WM * MT * 32 == BM, "four waves cover the panel");
    constexpr int KP = KC * 32;                                   // padded K
    constexpr int PITCH = KP * 4 + 16;                            // bytes per LDS row: plane 0 (KP fp16) | plane 1 | pad
    constexpr int GPR = KP / 8;                                   // 8-channel groups per row
    constexpr int GPT = BM * GPR / 256;                           // groups per thread (KC = 8: 8, 4: 4, 2: 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_p[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, hh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;
    const int K = p.C0 + p.C1;

    // ---- 1. the panel: group g = tid + 256 j -> row g / GPR, channels (g % GPR) * 8 .. + 7; all loads issued before any is used
    f32x4 v[GPT][2];
    f32x4 lg[GPT][2];                                             // LayerNorm gamma of the group's channels
    float lmean[GPT], linv[GPT];
    bool ok[GPT];
    if constexpr (!TAPS) {
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const int g = tid + 256 * j;
            const int row = g / GPR, c = (g % GPR) * 8;
            const long long m = m0 + row;
            ok[j] = m < p.M && c < K;
            const long long mm = ok[j] ? m : 0;
            const int cc = ok[j] ? c : 0;
            const float* src = cc < p.C0 ? p.a0 + mm * p.C0 + cc : p.a1 + mm * p.C1 + (cc - p.C0);
            v[j][0] = *reinterpret_cast<const f32x4*>(src);
            v[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
            if (p.ln_stats) {
                lmean[j] = p.ln_stats[2 * mm];
                linv[j] = p.ln_stats[2 * mm + 1];
                lg[j][0] = *reinterpret_cast<const f32x4*>(p.ln_gamma + cc);
                lg[j][1] = *reinterpret_cast<const f32x4*>(p.ln_gamma + cc + 4);
            }
        }
    } else {
        // several taps: the panel row of output point m is the concatenation over the taps of the shifted input pixels (the im2col
        // row, chunk index = tap * kchunks + kc as in the pack).  A thread's column group -- hence its tap and channels -- is the
        // same for all its rows (rows tid / GPR + (256 / GPR) j): one (frame, y, x) decomposition, then carries.
        constexpr int RSTEP = 256 / GPR;
        const int cg = tid % GPR, it = cg >> 2;
        const int tap = it / p.kchunks, kc = it - tap * p.kchunks;
        const int c = kc * 32 + (cg & 3) * 8;
        const int df = p.tdf[tap], dh = p.tdh[tap], dw = p.tdw[tap];
        const int HoWo = p.Ho * p.Wo;
        const long long mfirst = m0 + tid / GPR;
        long long bf = mfirst / HoWo;
        int hw = (int)(mfirst - bf * HoWo);
        int ho = hw / p.Wo, wo = hw - ho * p.Wo;
        int fr = (int)(bf % p.F);
        const bool cok = c < K && tap < p.ntaps;
        const float* base = c < p.C0 ? p.a0 + c : p.a1 + (c - p.C0);
        const int cs = c < p.C0 ? p.C0 : p.C1;
#pragma unroll
        for (int j = 0; j < GPT; ++j) {
            const long long m = mfirst + RSTEP * j;
            const int fi = fr + df, hi = ho * p.sh + dh, wi = wo * p.sw + dw;
            ok[j] = cok && m < p.M && (unsigned)fi < (unsigned)p.F && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
            const long long px = ok[j] ? ((bf + df) * p.Hi + hi) * p.Wi + wi : 0;
            const float* src = ok[j] ? base + px * cs : p.a0;
            v[j][0] = *reinterpret_cast<const f32x4*>(src);
            v[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
            wo += RSTEP;
            while (wo >= p.Wo) { wo -= p.Wo; ++ho; }
            while (ho >= p.Ho) { ho -= p.Ho; ++bf; ++fr; }
            while (fr >= p.F) fr -= p.F;
        }
    }
    // weight fragments: [chunk][Npad / 32][k-step][plane][half][n 32][16 B]; lane (n = l31, half hh): 1 KB contiguous per load
    const unsigned char* wlane = wp6 + (long long)(wn * NT) * 4096 + hh * 512 + l31 * 16;
    const long long wchunk = (long long)p.Npad * WROW;
    f16x8_p w[4][NT][2];                                          // ring of four k-step sets
    auto ldw = [&](int step, f16x8_p (&ws)[NT][2]) {
        const unsigned char* src = wlane + (step >> 1) * wchunk + (step & 1) * 2048;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) ws[nt][pl] = *reinterpret_cast<const f16x8_p*>(src + nt * 4096 + pl * 1024);
    };
    ldw(0, w[0]);
    ldw(1, w[1]);
    ldw(2, w[2]);
    // convert + park (LayerNorm as igemm3: ((x - mean) * inv) * gamma, then the operand scale)
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid + 256 * j;
        const int row = g / GPR, c = (g % GPR) * 8;
        f32x4 a = v[j][0], b = v[j][1];
        if (!TAPS && p.ln_stats) {
            a = (a - lmean[j]) * linv[j] * lg[j][0];
            b = (b - lmean[j]) * linv[j] * lg[j][1];
        }
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        a = ok[j] ? a * p.act_scale : z;
        b = ok[j] ? b * p.act_scale : z;
        h3::f16x8 pl[2];
        h3::split8(h3::sat16h(a.x), h3::sat16h(a.y), h3::sat16h(a.z), h3::sat16h(a.w), h3::sat16h(b.x), h3::sat16h(b.y), h3::sat16h(b.z),
                   h3::sat16h(b.w), pl);
        unsigned char* dst = smem_p + row * PITCH + c * 2;
        *reinterpret_cast<h3::f16x8*>(dst) = pl[0];
        *reinterpret_cast<h3::f16x8*>(dst + KP * 2) = pl[1];
    }
    __syncthreads();

    // ---- 2. K loop from LDS; k-step s = 16 channels
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const unsigned char* alane = smem_p + (wm * MT * 32 + l31) * PITCH + hh * 16;
    f16x8_p fa[2][MT][2];
    auto lda = [&](int step, f16x8_p (&a)[MT][2]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[mt][pl] = *reinterpret_cast<const f16x8_p*>(alane + mt * 32 * PITCH + pl * KP * 2 + step * 32);
    };
    auto mma = [&](const f16x8_p (&a)[MT][2], const f16x8_p (&ws)[NT][2]) {
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};          // small terms first (as igemm3)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][PA[term]], ws[nt][PB[term]], acc[mt][nt], 0, 0, 0);
    };
    constexpr int NS = 2 * KC;                                    // k-steps (4, 8, 16: multiples of 4)
    // A spent weight set (the MFMAs' B operand) keeps its registers until >= 4 younger MFMAs were issued (common.h: mfma_keep): a
    // k-step issues 3 MT NT MFMAs, so the set of step st is released at the end of step st + BACK; `spent` holds the SSA values
    // (no copies) past the ring's own re-load of the slot.
    constexpr int BACK = 3 * MT * NT >= 4 ? 1 : 2;
    f16x8_p spent[3][NT][2];
    lda(0, fa[0]);
#pragma unroll
    for (int s = 0; s < NS; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = s + u;
            if (st + 1 < NS) lda(st + 1, fa[(u + 1) & 1]);
            mma(fa[u & 1], w[u]);
            __builtin_amdgcn_sched_barrier(0);                    // (the re-load of set u+3 = u-1 stays behind this step's MFMAs)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) spent[st % 3][nt][pl] = w[u][nt][pl];
            if (st >= BACK) mfma_keep_set<MT, NT>(acc, spent[(st - BACK) % 3]);
            if (st + 3 < NS) ldw(st + 3, w[(u + 3) & 3]);
        }
    }
    // the last sets stay allocated to the end of the MFMA stream (nothing behind it reads LDS)
#pragma unroll
    for (int b = 0; b < BACK; ++b) mfma_keep_set<MT, NT>(acc, spent[(NS - 1 - b) % 3]);

    // ---- 3. epilogue (igemm_epilogue.h: igemm3_kernel's vector form, all residual loads in flight at once)
    const int q3 = l31 & 3;
    const long long bsmp = p.gn_raw ? m0 / p.gn_rows : 0;         // fused GroupNorm-apply residual: the panel lies inside one sample
    auto mrow = [&](int mt, int g) { return m0 + wm * MT * 32 + mt * 32 + 8 * g + 4 * hh + q3; };     // this lane's row after the transpose
    auto orow = [&](int mt, int g) -> long long {
        const long long m = mrow(mt, g);
        if (!TAPS || p.out_mode == 0) return m * p.N;
        const long long mm = m < p.M ? m : 0;                      // ConvTranspose parity scatter into [BF][2 Ho][2 Wo][N]
        const int HoWo = p.Ho * p.Wo;
        const long long bf = mm / HoWo;
        const int hw = (int)(mm - bf * HoWo), ho = hw / p.Wo, wo = hw - ho * p.Wo;
        return ((bf * (2 * p.Ho) + 2 * ho + p.par_a) * (long long)(2 * p.Wo) + 2 * wo + p.par_b) * p.N;
    };
    auto ncol = [&](int nt) { return wn * NT * 32 + nt * 32 + (l31 & ~3); };
    igemm_epilogue_vec<MT, NT>(p, acc, lane, bsmp, mrow, orow, ncol);
}

// shape-only rules (never the batch).  1: one tap at offset 0 (the lean path); 2: several taps / strides, tap-major im2col panel
static int panel_kind(const IgemmParams& p) {
    static const int on = debug_switch("DPC_IGEMM_PANEL", 1);
    const int K = p.C0 + p.C1;
    const int kc_all = p.ntaps * p.kchunks;
    if (!on || p.a0_stride || p.C0 % 8 || p.C1 % 8 || p.N != p.Npad || !(p.N == 64 || p.N == 128 || p.N == 256 || p.N == 384)) return 0;
    if (!(kc_all == 2 || kc_all == 4 || kc_all == 8 || (kc_all == 16 && p.N >= 128))) return 0;
    if (p.gn_raw && (p.gn_rows % 64 != 0 || p.out_mode != 0)) return 0;
    const bool plain = p.ntaps == 1 && p.tdf[0] == 0 && p.tdh[0] == 0 && p.tdw[0] == 0 && p.sh == 1 && p.sw == 1 && p.Hi == p.Ho && p.Wi == p.Wo;
    if (plain && p.out_mode == 0 && K <= 512) return 1;
    if (!p.ln_stats && (p.out_mode == 0 || p.out_mode == 2) && p.Wo >= 1 && p.Ho >= 1) return 2;
    return 0;
}
bool igemm3p_supported(const IgemmParams& p) { return panel_kind(p) != 0; }

template <int KC, int MT, int NT, int WM, int WN, bool TAPS, int BM>
static int launch_p(const IgemmParams& p, const void* wp6, hipStream_t s) {
    constexpr int LDS = BM * (KC * 128 + 16);
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3p_kernel<KC, MT, NT, WM, WN, TAPS, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        once = true;
    }
    const unsigned nwg = (unsigned)((p.M + BM - 1) / BM);
    hipLaunchKernelGGL((igemm3p_kernel<KC, MT, NT, WM, WN, TAPS, BM>), dim3(nwg), dim3(256), LDS, s, p, (const unsigned char*)wp6);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

template <int KC, bool TAPS>
static int launch_pk(const IgemmParams& p, const void* wp6, hipStream_t s) {
    if constexpr (KC == 16) {            // 32-row panels, waves 1 x 4
        switch (p.N) {
            case 384: return launch_p<KC, 1, 3, 1, 4, TAPS, 32>(p, wp6, s);
            case 256: return launch_p<KC, 1, 2, 1, 4, TAPS, 32>(p, wp6, s);
            default: return launch_p<KC, 1, 1, 1, 4, TAPS, 32>(p, wp6, s);
        }
    } else {
        switch (p.N) {
            case 384: return launch_p<KC, 2, 3, 1, 4, TAPS, 64>(p, wp6, s);
            case 256: return launch_p<KC, 2, 2, 1, 4, TAPS, 64>(p, wp6, s);
            case 128: return launch_p<KC, 1, 2, 2, 2, TAPS, 64>(p, wp6, s);
            default: return launch_p<KC, 1, 1, 2, 2, TAPS, 64>(p, wp6, s);
        }
    }
}

int launch_igemm3p(const IgemmParams& p, const void* wp6, hipStream_t s) {
    const int kc = p.ntaps * p.kchunks;
    if (panel_kind(p) == 1) {
        if (kc <= 2) return launch_pk<2, false>(p, wp6, s);
        if (kc <= 4) return launch_pk<4, false>(p, wp6, s);
        if (kc <= 8) return launch_pk<8, false>(p, wp6, s);
        return launch_pk<16, false>(p, wp6, s);
    }
    if (kc <= 2) return launch_pk<2, true>(p, wp6, s);
    if (kc <= 4) return launch_pk<4, true>(p, wp6, s);
    if (kc <= 8) return launch_pk<8, true>(p, wp6, s);
    return launch_pk<16, true>(p, wp6, s);
}

}  // namespace dpc
