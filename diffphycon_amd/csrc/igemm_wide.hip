// LDS-tiled 256 x 128 form of the f16x3 implicit GEMM (igemm6.hip: igemm3_kernel) for the GEMM-shaped deep levels of the Burgers
// U-Net (model/burgers_1d/unet.py:387-431: 3x3 convolutions at 4x32 / 2x16 / 1x8 images, M = 2-32 k rows, K = 9 C = 2304-18432, N =
// 256-1024) and every other launch with a long reduction and N >= 128.  igemm3's 128 x 64 tiles stream each wave's weight
// fragments from L2 (22 FLOP per byte moved into the CU): those levels ran at ~150 TF/s.  Here a 512-thread workgroup (8 waves as
// 4 x 2, 64 x 64 accumulators each) stages BOTH operands of a 32-channel chunk through LDS -- activations converted and split on
// the way in, as igemm3 does; the pre-split weights of the 128-column tile copied in MFMA fragment order ([k-step][plane][32-column
// block][half][n][16 B]: conflict-free b128 reads AND writes) -- so a byte fetched into the CU feeds 44 FLOP.
//
// Pipeline (r03, second form): the loads of chunk i+2 are issued while chunk i is multiplied, held in registers for a whole
// iteration and written to the other LDS image after chunk i+1's barrier -- two register stages with static names (the loop is
// unrolled by two), one barrier per chunk.  The first form prefetched one chunk ahead through lambdas with `if (more)` around the
// loads: hipcc kept the weight stage in SCRATCH (load -> s_waitcnt -> scratch_store inside the loop), branched around every
// activation load and fetched the tap offsets with global_load_ubyte + vmcnt(0) -- every iteration paid a full memory latency
// (3 us per chunk, 26 % of the MFMA-bound rate).  Now: no branch in the loop body (out-of-image rows load row 0 and are zeroed
// at the split; iterations past the end re-load the last chunk), the per-row (frame, y, x) decomposition is done once, and the
// tap offsets come from an LDS table (lgkmcnt, not vmcnt).
// Same arithmetic, operand scales, partial-product order and epilogues as igemm3_kernel, and the reduction is walked in the same
// (tap, chunk) order.  NOT claimed bit-identical to the narrow kernel: the split-K slicing differs (by N here, by reduction length
// there), so launches that split sum their slices in a different grouping (tools/bench_igemm.py: 2.4e-6 vs 2.1e-6 of the output
// range against x6 on the 512 -> 256 concat shape).  What IS guaranteed and tested is that a result never depends on the batch:
// tile shape, slice count and kernel choice are functions of (N, taps, K) only.
#include <algorithm>

#include "common.h"
#include "f16x3.h"
#include "igemm_epilogue.h"

namespace dpc {

namespace gw {
constexpr int BM = 256, BK = 32;
constexpr int RS = 144;                     // LDS bytes per A row (2 planes x 64 B + 16 pad), as igemm3
constexpr int WROW = 128;                   // packed weight bytes per output channel per iteration
constexpr int ABYTES = BM * RS;
constexpr int lds_bytes(int bn) { return 2 * (ABYTES + bn * WROW) + 32 * 4; }
}  // namespace gw

typedef _Float16 f16x8_w __attribute__((ext_vector_type(8)));

struct WStage {             // one chunk in flight: 2 activation rows x 8 channels, 2 x 16 weight bytes, row validity bits
    f32x4 a[4];             // [row i][half]: a[2 i], a[2 i + 1]
    uint4 b[2];
    unsigned ok;
};

// LDS images (r03 PMC: half of the LDS-array cycles of the first form were bank conflicts):
//   A  [row 256][144 B]: plane 0 at +0, plane 1 at +64; a fragment read (lane = row, 16 B) walks 9 slots per row -> the 16-lane
//      groups of ds_read_b128 hit 16 different slots; a thread writes 8 channels = 16 B per plane, and the 8 lanes of a
//      ds_write_b128 group hold 8 CONSECUTIVE ROWS of one column piece (slots 9 r + j: distinct mod 8).
//   B  [32-column block 4][k-step 2][plane 2][half 2][n 32][16 B] = the packed weights' own order (pack_weights_g6_kernel): a
//      fragment read covers 1 KB contiguously and the copy-in is linear on both sides.
// BN = 128: 8 waves as 4 x 2, 64 x 64 each; BN = 64 (r03: the 64-column stride-2 down convolutions and other long reductions into
// 64 columns): 8 waves as 8 x 1, 32 x 64 each
template <bool SPLIT, int BN>
__global__ __launch_bounds__(512, 1) void igemm3w_kernel(IgemmParams p, const unsigned char* __restrict__ wp6) {
    h3::hw_sat_enable();                               // (f16x3.h: operand conversions saturate in hardware)
    using namespace gw;
    constexpr int BBYTES = BN * WROW, TAPOFF = 2 * (ABYTES + BBYTES);
    constexpr int MT = BN == 128 ? 2 : 1;                         // 32-row blocks per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = BN == 128 ? wave >> 1 : wave, wn = BN == 128 ? wave & 1 : 0, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    int bid = blockIdx.x;
    {
        const int nb = mtiles * ntn, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long m0 = (long long)__builtin_amdgcn_readfirstlane(bid / ntn) * BM;
    const int n0 = __builtin_amdgcn_readfirstlane((bid % ntn) * BN);
    int* taptab = reinterpret_cast<int*>(smem_w + TAPOFF);
    if (tid < 32) taptab[tid] = (p.tdf[tid] & 0xff) | ((p.tdh[tid] & 0xff) << 8) | ((p.tdw[tid] & 0xff) << 16);

    // ---- A: thread -> rows wave * 32 + i * 16 + (lane / 32) * 8 + lane % 8 (i = 0, 1), channels ((lane / 8) % 4) * 8 .. + 7 of the
    //      chunk; the row geometry is fixed for the whole launch
    const int acol = ((lane >> 3) & 3) * 8;
    const int arow0 = wave * 32 + (lane >> 5) * 8 + (lane & 7);
    const int HoWo = p.Ho * p.Wo;
    const int K = p.C0 + p.C1;
    int rf[2], rh[2], rw[2];
    long long rbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + arow0 + 16 * i;
        const bool ok = m < p.M;
        const long long mm = ok ? m : 0;
        const int bf = (int)(mm / HoWo);
        const int hw = (int)(mm - (long long)bf * HoWo);
        const int ho = hw / p.Wo;
        rf[i] = ok ? bf % p.F : -(1 << 20);                       // (a row past M fails every range test below)
        rh[i] = ho * p.sh;
        rw[i] = (hw - ho * p.Wo) * p.sw;
        rbase[i] = ((long long)bf * p.Hi + rh[i]) * p.Wi + rw[i];
    }
    const float* pa0[2] = {p.a0, p.a0};                          // this tap's pixel in source 0 / source 1 (row 0 when out of the image)
    const float* pa1[2] = {p.a1, p.a1};
    unsigned rok = 0;
    // weights: the tile's 128 columns of one iteration are 16 KB contiguous in the pack, already in fragment order: a straight copy,
    // thread t moves bytes [16 t, 16 t + 16) and [8192 + 16 t, ..) (1 KB per wave instruction on both sides, conflict-free writes)
    const unsigned char* wsrc = wp6 + (long long)(n0 >> 5) * 4096 + tid * 16;
    const int bdst = tid * 16;
    const long long wstep = (long long)p.Npad * WROW;
    const int nit_all = p.ntaps * p.kchunks;
    const int nsl = SPLIT ? p.ksplit : 1;
    const int it0 = (int)((long long)nit_all * blockIdx.y / nsl), niter = (int)((long long)nit_all * (blockIdx.y + 1) / nsl);
    __syncthreads();                                              // tap table visible

    int cur_tap = -1;
    auto issue = [&](int it_req, WStage& st) {
        const int it = it_req < niter ? it_req : niter - 1;       // past the end: re-load the last chunk (never multiplied)
        const int tap = it / p.kchunks, kc = it - tap * p.kchunks;
        if (tap != cur_tap) {                                     // wave-uniform, once per tap
            cur_tap = tap;
            const int t = taptab[tap];
            const int df = (signed char)(t & 0xff), dh = (signed char)((t >> 8) & 0xff), dw = (signed char)((t >> 16) & 0xff);
            const long long d = ((long long)df * p.Hi + dh) * p.Wi + dw;
            rok = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool v = (unsigned)(rf[i] + df) < (unsigned)p.F && (unsigned)(rh[i] + dh) < (unsigned)p.Hi &&
                               (unsigned)(rw[i] + dw) < (unsigned)p.Wi;
                const long long px = v ? rbase[i] + d : 0;
                pa0[i] = p.a0 + px * p.C0;
                pa1[i] = p.a1 + px * p.C1;
                rok |= v ? 1u << i : 0u;
            }
        }
        const int c = kc * BK + acol;
        const bool cok = c < K;                                   // a chunk column past K (K % 32 != 0) reads channel 0 and is zeroed
        const bool s0 = c < p.C0 || !cok;
        const int cc = cok ? (s0 ? c : c - p.C0) : 0;
        st.ok = cok ? rok : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* q = (s0 ? pa0[i] : pa1[i]) + cc;
            st.a[2 * i] = *reinterpret_cast<const f32x4*>(q);
            st.a[2 * i + 1] = *reinterpret_cast<const f32x4*>(q + 4);
        }
        const unsigned char* ws = wsrc + (long long)it * wstep;
        st.b[0] = *reinterpret_cast<const uint4*>(ws);
        if constexpr (BN == 128) st.b[1] = *reinterpret_cast<const uint4*>(ws + 8192);
    };
    auto stash = [&](const WStage& st, int buf) {
        unsigned char* A = smem_w + buf * ABYTES;
        unsigned char* B = smem_w + 2 * ABYTES + buf * BBYTES;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = st.ok >> i & 1;
            const f32x4 u = ok ? st.a[2 * i] * p.act_scale : z, v = ok ? st.a[2 * i + 1] * p.act_scale : z;
            h3::f16x8 pl[2];
            h3::split8(h3::sat16h(u.x), h3::sat16h(u.y), h3::sat16h(u.z), h3::sat16h(u.w), h3::sat16h(v.x), h3::sat16h(v.y), h3::sat16h(v.z),
                       h3::sat16h(v.w), pl);
            unsigned char* dst = A + (arow0 + 16 * i) * RS + acol * 2;
            *reinterpret_cast<h3::f16x8*>(dst) = pl[0];
            *reinterpret_cast<h3::f16x8*>(dst + 64) = pl[1];
        }
        *reinterpret_cast<uint4*>(B + bdst) = st.b[0];
        if constexpr (BN == 128) *reinterpret_cast<uint4*>(B + bdst + 8192) = st.b[1];
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int a_lane = (wm * MT * 32 + l31) * RS + hh * 16;
    const int b_lane = wn * 8192 + hh * 512 + l31 * 16;
    // The matrix work trails the fragment reads by half a chunk: right after a barrier a wave issues the k-step-0 reads of the new
    // image and multiplies the k-step-1 fragments it read BEFORE the barrier (held in registers), so the LDS latency that follows
    // every barrier -- both waves of a SIMD leave it together -- is covered by 12 MFMAs instead of idling the matrix pipe.
    f16x8_w fa0[MT][2], fb0[2][2], fa1[MT][2], fb1[2][2];      // [mt | nt][plane] of k-step 0 / 1
    auto read_frags = [&](int buf, int ks, f16x8_w (&fa)[MT][2], f16x8_w (&fb)[2][2]) {
        const unsigned char* A = smem_w + buf * ABYTES;
        const unsigned char* B = smem_w + 2 * ABYTES + buf * BBYTES;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb[nt][pl] = *reinterpret_cast<const f16x8_w*>(B + nt * 4096 + (ks * 2 + pl) * 1024 + b_lane);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fa[mt][pl] = *reinterpret_cast<const f16x8_w*>(A + a_lane + mt * 32 * RS + pl * 64 + ks * 32);
    };
    auto mfma12 = [&](const f16x8_w (&fa)[MT][2], const f16x8_w (&fb)[2][2]) {
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};         // small terms first (as igemm3)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[mt][PA[term]], fb[nt][PB[term]], acc[mt][nt], 0, 0, 0);
    };

    // One phase = one chunk (image `img`), one barrier:  read k-step 0 | multiply the previous chunk's k-step 1 (its fragments were
    // read before the barrier) and convert + store the NEXT chunk into the other image | multiply k-step 0 | read k-step 1 | issue
    // the loads of the chunk after next into the stage just stored.  A fragment register is re-loaded only after the MFMAs that
    // read it and twelve more have been issued (the matrix pipe reads its B operand while executing; an LDS return must never
    // overtake a queued MFMA -- DESIGN.md 6.2): the scheduling barriers pin exactly that order and leave the rest to hipcc.
    WStage s0, s1;
    auto phase = [&](int img, WStage& st, int it_next, bool trail) {
        // (k-step 0's B fragments were multiplied LAST in the previous phase: their registers stay allocated through the twelve trailing
        //  MFMAs below -- common.h: mfma_keep -- so this phase's k-step-0 reads land in another register set)
        f16x8_w fb0_spent[2][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) fb0_spent[nt][pl] = fb0[nt][pl];
        read_frags(img, 0, fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (trail) {
            mfma12(fa1, fb1);
            mfma_keep_set<MT, 2>(acc, fb0_spent);
        }
        stash(st, img ^ 1);
        mfma12(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(img, 1, fa1, fb1);
        issue(it_next, st);
        __syncthreads();
    };
    issue(it0, s0);
    issue(it0 + 1, s1);
    stash(s0, 0);
    issue(it0 + 2, s0);
    __syncthreads();
    phase(0, s1, it0 + 3, false);
    int it = it0 + 1;
    for (; it + 1 < niter; it += 2) {    // chunk `it` in image 1 (loads in s0 -> image 0 = chunk it+1), then chunk it+1 in image 0
        phase(1, s0, it + 3, true);
        phase(0, s1, it + 4, true);
    }
    if (it < niter) phase(1, s0, it + 3, true);
    mfma12(fa1, fb1);                    // k-step 1 of the last chunk
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
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = n0 + wn * 64 + nt * 32 + l31;
                    if (n < p.N) pb[m * p.N + n] = acc[mt][nt][r];
                }
            }
    } else {
        const int q3 = l31 & 3;
        auto mrow = [&](int mt, int g) { return m0 + wm * MT * 32 + mt * 32 + 8 * g + 4 * hh + q3; };
        auto orow = [&](int mt, int g) { return mrow(mt, g) * p.N; };            // (out_mode 0 only: see igemm3w_supported)
        auto ncol = [&](int nt) { return n0 + wn * 64 + nt * 32 + (l31 & ~3); };
        igemm_epilogue_vec<MT, 2>(p, acc, lane, 0, mrow, orow, ncol);
    }
}

// shape-only rules (never the batch): long reductions, plain [M][N] output; 128-column tiles when Npad allows, else 64
bool igemm3w_supported(const IgemmParams& p) {
    static const int on = debug_switch("DPC_IGEMM_LDSB", 1);
    const int nit = p.ntaps * p.kchunks;
    static const int nit_min = debug_switch("DPC_IGEMM_WMIN", 24);
    return on && nit >= nit_min && p.out_mode == 0 && !p.ln_stats && !p.gn_raw && !p.a0_stride && p.N % 4 == 0 && p.Npad % 64 == 0 && p.N >= 64;
}
// split-K slices of the wide kernel: the deep levels have few rows and many columns (N = 512: 2, N >= 1024: 4), >= 12 chunks each
int igemm3w_slices(const IgemmParams& p) {
    const int nit = p.ntaps * p.kchunks;
    int nsl = p.N >= 1024 ? 4 : p.N >= 512 ? 2 : 1;
    while (nsl > 1 && nit / nsl < 12) nsl >>= 1;
    return nsl;
}

template <int BN>
static int launch_w(const IgemmParams& p, const void* wp6, int nsl, hipStream_t s) {
    using namespace gw;
    constexpr int LDS = lds_bytes(BN);
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3w_kernel<true, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3w_kernel<false, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        once = true;
    }
    const int mtiles = (int)((p.M + BM - 1) / BM);
    const unsigned nwg = (unsigned)mtiles * (p.Npad / BN);
    if (nsl > 1) hipLaunchKernelGGL((igemm3w_kernel<true, BN>), dim3(nwg, nsl), dim3(512), LDS, s, p, (const unsigned char*)wp6);
    else hipLaunchKernelGGL((igemm3w_kernel<false, BN>), dim3(nwg), dim3(512), LDS, s, p, (const unsigned char*)wp6);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_igemm3w(const IgemmParams& p, const void* wp6, int nsl, hipStream_t s) {
    return p.Npad % 128 == 0 ? launch_w<128>(p, wp6, nsl, s) : launch_w<64>(p, wp6, nsl, s);
}

}  // namespace dpc
