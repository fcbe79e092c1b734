// Pixel-tile form of the f16x3 implicit GEMM for 3 x 3 stride-1 convolutions on SMALL images with a LONG reduction: the 4 x 32 / 2 x 16 /
// 1 x 8 levels of the Burgers U-Net (model/burgers_1d/unet.py:157-191 ResnetBlock / Block at :387-431; 256 -> 1024 channels, K = 9 C =
// 2304 .. 18432).  igemm_wide.hip runs them as an im2col GEMM: every (tap, 32-channel chunk) iteration loads the tap's shifted rows from
// global memory and converts them again -- each input pixel is fetched and split into its fp16 planes NINE times, and per chunk a wave
// issues 135 VALU instructions (72 of them the conversion) next to 24 MFMAs: 0.29 of the f16x3 roof, both waves of a SIMD waiting on
// instruction issue (r03 PMC).  Here, as in igemm_tile.hip (whose whole-K staging stops at K = 128), a workgroup stages the UNIQUE pixels
// of its 128 output rows -- the flattened pixel range [m0 - W - 1, m0 + 127 + W + 1], one LDS row of 2 x 32 fp16 per pixel, plus a zero
// row -- ONCE per 32-channel block of K, and the nine taps are per-lane LDS row offsets of the A fragment (a lane whose (y + dh, x + dw)
// leaves its image reads the zero row): 1/9 of the loads and conversions, 216 MFMAs per wave between two barriers.
//   * 256 threads = 4 waves as 2 x 2, wave tile 64 rows x 64 columns (MT = NT = 2); BM = 128, BN = 128.
//   * K loop: block kb -> [ (issue the global loads of block kb + 1 into registers) | 9 taps x 2 k-steps x 12 MFMAs from LDS, weight
//     fragments straight from L2 in the packed fragment order through a ring of three k-step sets | barrier | convert + store block
//     kb + 1 into the (single) LDS image | barrier ]: the HBM / L2 latency of a block hides behind the previous block's MFMAs, the
//     conversion (~100 VALU per thread and block) behind the CU's second workgroup.
//   * split-K by SHAPE only (igemm3w_slices: N >= 512: 2, N >= 1024: 4 slices of the channel blocks), raw partials + the fixed-order
//     reduce kernel of igemm6.hip; otherwise the shared vector epilogue (igemm_epilogue.h).
//   * spent weight sets (the MFMAs' B operand) keep their registers for >= 12 younger MFMAs (common.h: mfma_keep).
// Arithmetic, operand scales and the order of the three partial products per output element are igemm3_kernel's; the reduction runs
// channel-block-major (kb, tap) instead of (tap, kb), so results equal the wide kernel's to fp32 summation order only.  A result never
// depends on the batch: tile shape, slice count and kernel choice are functions of (N, taps, K, H, W).
#include <algorithm>

#include "common.h"
#include "f16x3.h"
#include "igemm_epilogue.h"

namespace dpc {

typedef _Float16 f16x8_i __attribute__((ext_vector_type(8)));

namespace gi {
constexpr int BM = 128, BN = 128, MT = 2, NT = 2;
constexpr int PITCH = 144;                  // bytes per LDS row: plane 0 (32 fp16) | plane 1 | 16 pad (conflict-free b128 fragment reads)
constexpr int WMAX = 32;                    // widest image: halo = 2 (W + 1) rows
constexpr int NRMAX = BM + 2 * (WMAX + 1);
constexpr int JMAX = (NRMAX * 4 + 255) / 256;          // 8-channel groups per thread and block (4 per pixel)
constexpr int LDS = (NRMAX + 1) * PITCH;
}  // namespace gi

// NTAPS: 9, or 3 for the H = 1 level (the launcher of igemm6.hip drops the six taps that only ever see padding)
template <bool SPLIT, int NTAPS>
__global__ __launch_bounds__(256, 2) void igemm3i_kernel(IgemmParams p, const unsigned char* __restrict__ wp6) {
    h3::hw_sat_enable();                               // (f16x3.h: operand conversions saturate in hardware)
    using namespace gi;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_i[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const long long m0 = (long long)(blockIdx.x / ntn) * BM;
    const int n0 = (blockIdx.x % ntn) * BN;
    const int HW = p.Hi * p.Wi, W = p.Wi;
    const int K = p.C0 + p.C1, kchunks = p.kchunks;
    int lo = 1 << 30, hi = -(1 << 30);                             // tap offsets in flattened pixels (wave-uniform)
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        const int o = p.tdh[t] * W + p.tdw[t];
        lo = min(lo, o); hi = max(hi, o);
    }
    const int NR = BM + hi - lo;                                   // staged pixel rows; row NR is the zero row
    const long long P0 = m0 + lo;                                  // flattened pixel of LDS row 0 (may be < 0)
    const long long PT = (long long)p.BF * HW;
    // channel blocks of this slice
    const int nsl = SPLIT ? p.ksplit : 1;
    const int kb0 = (int)((long long)kchunks * blockIdx.y / nsl), kb1 = (int)((long long)kchunks * (blockIdx.y + 1) / nsl);

    // ---- staging: group g = tid + 256 j -> LDS row g >> 2 (pixel P0 + row), channels (g & 3) * 8 .. + 7 of the block
    f32x4 v[JMAX][2];
    const int ngroups = NR * 4;
    auto issue = [&](int kb) {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int g = tid + 256 * j;
            const int row = g >> 2, c = kb * 32 + (g & 3) * 8;
            const long long px = P0 + row;
            const bool ok = g < ngroups && px >= 0 && px < PT && c < K;
            const long long pp = ok ? px : 0;
            const int cc = ok ? c : 0;
            const float* src = cc < p.C0 ? p.a0 + pp * p.C0 + cc : p.a1 + pp * p.C1 + (cc - p.C0);
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            v[j][0] = *reinterpret_cast<const f32x4*>(src);
            v[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
            if (!ok) { v[j][0] = z; v[j][1] = z; }
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
            const int g = tid + 256 * j;
            if (g < ngroups) {
                const f32x4 a = v[j][0] * p.act_scale, b = v[j][1] * p.act_scale;
                h3::f16x8 pl[2];
                h3::split8(h3::sat16h(a.x), h3::sat16h(a.y), h3::sat16h(a.z), h3::sat16h(a.w), h3::sat16h(b.x), h3::sat16h(b.y), h3::sat16h(b.z),
                           h3::sat16h(b.w), pl);
                unsigned char* dst = smem_i + (g >> 2) * PITCH + (g & 3) * 16;
                *reinterpret_cast<h3::f16x8*>(dst) = pl[0];
                *reinterpret_cast<h3::f16x8*>(dst + 64) = pl[1];
            }
        }
    };
    issue(kb0);
    for (int i = tid; i < PITCH / 16; i += 256) *reinterpret_cast<uint4*>(smem_i + NR * PITCH + i * 16) = uint4{0, 0, 0, 0};      // the zero row

    // ---- per lane: the LDS row of every tap's input pixel for its MT output rows (fixed for the launch)
    int arow[NTAPS][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = wm * MT * 32 + mt * 32 + l31;
        const long long m = m0 + r;
        const long long mm = m < p.M ? m : 0;
        const int hw = (int)(mm % HW);
        const int y = hw / W, x = hw - y * W;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int dh = p.tdh[t], dw = p.tdw[t];
            const bool ok = m < p.M && (unsigned)(y + dh) < (unsigned)p.Hi && (unsigned)(x + dw) < (unsigned)W;
            arow[t][mt] = (ok ? r + dh * W + dw - lo : NR) * PITCH + hh * 16;
        }
    }

    // weight fragments: [tap * kchunks + kc][Npad / 32][k-step][plane][half][n 32][16 B]; lane (n = l31, half hh)
    const unsigned char* wlane = wp6 + (long long)((n0 >> 5) + wn * NT) * 4096 + hh * 512 + l31 * 16;
    const long long wchunk = (long long)p.Npad * 128;
    // ring of THREE k-step sets with static slots: a block is 2 NTAPS = 18 (6) steps = a whole number of ring turns, so slot (step % 3)
    // holds across blocks without copies; the set of step s + 2 is requested when step s has been multiplied
    f16x8_i w[3][NT][2];
    constexpr int NS = 2 * NTAPS;                                 // steps per block: (tap s >> 1, k-step s & 1)
    static_assert(NS % 3 == 0, "the weight ring turns a whole number of times per block");
    auto ldw = [&](int kb, int s, f16x8_i (&ws)[NT][2]) {
        const unsigned char* src = wlane + ((long long)(s >> 1) * kchunks + kb) * wchunk + (s & 1) * 2048;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) ws[nt][pl] = *reinterpret_cast<const f16x8_i*>(src + nt * 4096 + pl * 1024);
    };
    ldw(kb0, 0, w[0]);
    ldw(kb0, 1, w[1]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w[2][nt][pl] = w[0][nt][pl];     // (a defined value for the first keep)

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    park();
    __syncthreads();

    f16x8_i fa[2][MT][2];
    auto lda = [&](int s, f16x8_i (&a)[MT][2]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[mt][pl] = *reinterpret_cast<const f16x8_i*>(smem_i + arow[s >> 1][mt] + pl * 64 + (s & 1) * 32);
    };
    auto mma = [&](const f16x8_i (&a)[MT][2], const f16x8_i (&ws)[NT][2]) {
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};          // small terms first (as igemm3)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][PA[term]], ws[nt][PB[term]], acc[mt][nt], 0, 0, 0);
    };
    for (int kb = kb0; kb < kb1; ++kb) {
        const bool more = kb + 1 < kb1;
        const int kbn = more ? kb + 1 : kb;                        // (past the end: the last block is requested again and dropped -- no branch
        issue(kbn);                                                //  around the loads: hipcc would wait vmcnt(0) behind it)
        lda(0, fa[0]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + 1 < NS) lda(s + 1, fa[(s + 1) & 1]);
            mma(fa[s & 1], w[s % 3]);
            __builtin_amdgcn_sched_barrier(0);
            // the set of step s - 1 (slot (s + 2) % 3) keeps its registers through this step's 12 MFMAs, then takes the set of step s + 2
            mfma_keep_set<MT, NT>(acc, w[(s + 2) % 3]);
            if (s + 2 < NS) ldw(kb, s + 2, w[(s + 2) % 3]);
            else ldw(kbn, s + 2 - NS, w[(s + 2) % 3]);
        }
        __syncthreads();                                           // every wave has read this block's fragments
        if (more) park();
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) mfma_keep_set<MT, NT>(acc, w[i]);

    // ---- epilogue
    if constexpr (SPLIT) {             // raw partial accumulators [slice][M][N]; finished by igemm3_reduce_kernel
        float* pb = p.part + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * MT * 32 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m >= p.M) continue;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + wn * NT * 32 + nt * 32 + l31;
                    if (n < p.N) pb[m * p.N + n] = acc[mt][nt][r];
                }
            }
    } else {
        const int q3 = l31 & 3;
        auto mrow = [&](int mt, int g) { return m0 + wm * MT * 32 + mt * 32 + 8 * g + 4 * hh + q3; };
        auto orow = [&](int mt, int g) { return mrow(mt, g) * p.N; };
        auto ncol = [&](int nt) { return n0 + wn * NT * 32 + nt * 32 + (l31 & ~3); };
        igemm_epilogue_vec<MT, NT>(p, acc, lane, 0, mrow, orow, ncol);
    }
}

// shape-only rule (never the batch): the nine taps of a 3 x 3 'same' convolution in (dh, dw) order, stride 1, images at most 32 wide that the
// halo-tile kernel does not take (H or W not a multiple of 8), a reduction of >= 24 (tap, chunk) iterations into >= 128 columns
bool igemm3i_supported(const IgemmParams& p) {
    static const int on = debug_switch("DPC_IGEMM_IMG", 1);
    if (!on || !(p.ntaps == 9 || p.ntaps == 3) || p.sh != 1 || p.sw != 1 || p.Hi != p.Ho || p.Wi != p.Wo || p.a0_stride || p.ln_stats || p.gn_raw)
        return false;
    if (p.out_mode != 0 || p.N % 4 || p.Npad % gi::BN || p.C0 % 8 || p.C1 % 8 || p.Wi > gi::WMAX || p.ntaps * p.kchunks < 24) return false;
    if (p.Hi % 8 == 0 && p.Wi % 8 == 0) return false;             // (the halo-tile kernel's shapes)
    int lo = 1 << 30, hi = -(1 << 30);
    for (int t = 0; t < p.ntaps; ++t) {
        if (p.tdf[t] != 0 || std::abs((int)p.tdh[t]) > 1 || std::abs((int)p.tdw[t]) > 1) return false;
        const int o = p.tdh[t] * p.Wi + p.tdw[t];
        lo = std::min(lo, o); hi = std::max(hi, o);
    }
    return hi - lo <= 2 * (gi::WMAX + 1);
}

template <int NTAPS>
static int launch_i(const IgemmParams& p, const void* wp6, int nsl, hipStream_t s) {
    using namespace gi;
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3i_kernel<true, NTAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3i_kernel<false, NTAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        once = true;
    }
    int lo = 1 << 30, hi = -(1 << 30);
    for (int t = 0; t < NTAPS; ++t) {
        const int o = p.tdh[t] * p.Wi + p.tdw[t];
        lo = std::min(lo, o); hi = std::max(hi, o);
    }
    const size_t lds = (size_t)(BM + hi - lo + 1) * PITCH;
    const unsigned nwg = (unsigned)((p.M + BM - 1) / BM) * (unsigned)(p.Npad / BN);
    if (nsl > 1) hipLaunchKernelGGL((igemm3i_kernel<true, NTAPS>), dim3(nwg, nsl), dim3(256), lds, s, p, (const unsigned char*)wp6);
    else hipLaunchKernelGGL((igemm3i_kernel<false, NTAPS>), dim3(nwg), dim3(256), lds, s, p, (const unsigned char*)wp6);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_igemm3i(const IgemmParams& p, const void* wp6, int nsl, hipStream_t s) {
    return p.ntaps == 9 ? launch_i<9>(p, wp6, nsl, s) : launch_i<3>(p, wp6, nsl, s);
}

}  // namespace dpc
