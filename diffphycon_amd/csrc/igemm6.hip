// Implicit GEMM with fp32 semantics on the bf16 matrix cores ("bf16x6", see conv3x6.hip for the arithmetic): the same op
// family as igemm.hip (strided (1,4,4) down-conv, the four parity classes of ConvTranspose3d, 1x1 convs and attention
// projections with the fused channel-LayerNorm prologue, the 2-D U-Net's 3x3 / 2x2-stride-2 convs; virtual concat,
// residual epilogue, the three output modes), with every fp32 product evaluated as six exact bf16 partial products.
// Against igemm.hip the MFMA time per fp32 flop drops 2.67x (6 x 32 cycles vs 8 x 64 cycles per 16 channels).
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile 128 rows x BN = {64, 128} columns x 32 channels per iteration.
// A: global fp32 -> registers (prefetched one iteration ahead, LayerNorm applied) -> exact 3-way bf16 split -> LDS rows of
//    3 planes x 32 k bf16 = 192 B + 16 B pad (208 B = 52 banks = 13 x 4: any 16 rows distinct mod 16 hit 16 distinct
//    4-bank groups, so the ds_read_b128 fragment reads are conflict-free), double buffered.
// B: weights pre-split at load time into [iteration][n][3 planes][32 k] bf16 and read as fragments straight from L2/L1
//    into registers one iteration ahead (no LDS, no extra barrier).
//
// igemm3_kernel (default, DPC_IGEMM_MODE=f16x3) is the same kernel with the 2-way fp16 operand split of conv3f3.hip:
// 22-bit operands pre-scaled by 2^4 / 2^12, three partial products per product, A rows of 2 planes x 32 k fp16 = 128 B
// + 16 B pad (144 B = 9 x 16 B), weights [iteration][n][2 planes][32 k] fp16, epilogue rescale by 2^-16.
#include <algorithm>
#include <cstdio>

#include "common.h"
#include "igemm_epilogue.h"

namespace dpc {

namespace g6 {
constexpr int BM = 128, BK = 32;
constexpr int RS = 208;                     // LDS bytes per A row

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float lo_f32(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_f32(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }
__device__ __forceinline__ void split3(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    p1.x = cvt_pk_bf16(v.x, v.y);
    p1.y = cvt_pk_bf16(v.z, v.w);
    const float r0 = v.x - lo_f32(p1.x), r1 = v.y - hi_f32(p1.x), r2 = v.z - lo_f32(p1.y), r3 = v.w - hi_f32(p1.y);
    p2.x = cvt_pk_bf16(r0, r1);
    p2.y = cvt_pk_bf16(r2, r3);
    const float s0 = r0 - lo_f32(p2.x), s1 = r1 - hi_f32(p2.x), s2 = r2 - lo_f32(p2.y), s3 = r3 - hi_f32(p2.y);
    p3.x = cvt_pk_bf16(s0, s1);
    p3.y = cvt_pk_bf16(s2, s3);
}
}  // namespace g6

typedef __bf16 bf16x8_g __attribute__((ext_vector_type(8)));

template <int BN>
__global__ __launch_bounds__(256, 2) void igemm6_kernel(IgemmParams p, const unsigned char* __restrict__ wp6) {
    using namespace g6;
    constexpr int NT = BN / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    unsigned char* As0 = smem6;
    unsigned char* As1 = As0 + BM * RS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    int bid = blockIdx.x;
    {
        const int nb = mtiles * ntn, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long m0 = (long long)(bid / ntn) * BM;
    const int n0 = (bid % ntn) * BN;

    // ---- per-thread A rows: 4 rows (tid/8 + 32 i), one float4 column (tid%8)*4   (identical to igemm.hip)
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int HoWo = p.Ho * p.Wo;
    int r_bf[4], r_f[4], r_h[4], r_w[4];
    bool r_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = m0 + arow + 32 * i;
        r_ok[i] = m < p.M;
        const long long mm = r_ok[i] ? m : 0;
        const int bf = (int)(mm / HoWo);
        const int hw = (int)(mm - (long long)bf * HoWo);
        const int ho = hw / p.Wo;
        r_bf[i] = bf;
        r_f[i] = bf % p.F;
        r_h[i] = ho * p.sh;
        r_w[i] = (hw - ho * p.Wo) * p.sw;
    }
    const int K = p.C0 + p.C1;
    f32x4 ra[4];
    long long roff[4];
    bool rvalid[4];
    int cur_tap = -1;
    auto load_a = [&](int it) {
        const int tap = it / p.kchunks, kc = it - tap * p.kchunks;
        if (tap != cur_tap) {
            cur_tap = tap;
            const int df = p.tdf[tap], dh = p.tdh[tap], dw = p.tdw[tap];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int fi = r_f[i] + df, hi = r_h[i] + dh, wi = r_w[i] + dw;
                rvalid[i] = r_ok[i] && (unsigned)fi < (unsigned)p.F && (unsigned)hi < (unsigned)p.Hi &&
                            (unsigned)wi < (unsigned)p.Wi;
                roff[i] = ((long long)(r_bf[i] + df) * p.Hi + hi) * p.Wi + wi;
            }
        }
        const int c = kc * BK + acol;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = p.a0; cs = p.cs0; cc = c; }
        else { src = p.a1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && rvalid[i]) {
                v = *reinterpret_cast<const f32x4*>(src + roff[i] * cs + cc);
                if (p.ln_stats) {
                    const float mean = p.ln_stats[2 * roff[i]], inv = p.ln_stats[2 * roff[i] + 1];
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_gamma + c);
                    v = (v - mean) * inv * g;
                }
            }
            ra[i] = v;
        }
    };
    auto store_a = [&](unsigned char* As) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 p1, p2, p3;
            split3(ra[i], p1, p2, p3);
            unsigned char* dst = As + (arow + 32 * i) * RS + acol * 2;
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + 64) = p2;
            *reinterpret_cast<uint2*>(dst + 128) = p3;
        }
    };
    // weight fragments: [it][Npad][3][32] bf16 = 192 B per n; lane (n = l31, half hh) reads 8 k = 16 B per plane and k16 step
    const unsigned char* wlane = wp6 + ((long long)n0 + wn * (BN / 2) + l31) * 192 + hh * 16;
    bf16x8_g wc[2][NT][3], wx[2][NT][3];
    auto ldw = [&](int it, bf16x8_g (&w)[2][NT][3]) {
        const unsigned char* src = wlane + (long long)it * p.Npad * 192;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    w[ks][nt][pl] = *reinterpret_cast<const bf16x8_g*>(src + nt * 32 * 192 + pl * 64 + ks * 32);
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int a_lane = (wm * 64 + l31) * RS + hh * 16;
    const int niter = p.ntaps * p.kchunks;
    load_a(0);
    ldw(0, wc);
    store_a(As0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const bool more = it + 1 < niter;
        if (more) { load_a(it + 1); ldw(it + 1, wx); }
        const unsigned char* As = (it & 1) ? As1 : As0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_g a[2][3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[mt][pl] = *reinterpret_cast<const bf16x8_g*>(As + a_lane + mt * 32 * RS + pl * 64 + ks * 32);
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][PA[term]], wc[ks][nt][PB[term]], acc[mt][nt], 0, 0, 0);
        }
        if (more) {
            store_a((it & 1) ? As0 : As1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wc[ks][nt][pl] = wx[ks][nt][pl];
        }
        __syncthreads();
    }
    // ---- epilogue (same accumulator layout and output modes as igemm.hip)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 32 + l31;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m >= p.M) continue;
                float v = acc[mt][nt][r] + bv;
                if (p.resid) v += p.resid[m * p.N + n];
                long long o;
                if (p.out_mode == 0) {
                    o = m * p.N + n;
                } else if (p.out_mode == 1) {
                    const long long bf = m / HoWo, hw = m - bf * HoWo;
                    o = (bf * p.N + n) * (long long)HoWo + hw;
                } else {
                    const long long bf = m / HoWo;
                    const int hw = (int)(m - bf * HoWo), ho = hw / p.Wo, wo = hw - ho * p.Wo;
                    o = ((bf * (2 * (HoWo / p.Wo)) + 2 * ho + p.par_a) * (long long)(2 * p.Wo) + 2 * wo + p.par_b) * p.N + n;
                }
                p.out[o] = v;
            }
        }
    }
}

namespace g3 {
constexpr int BM = 128, BK = 32;
constexpr int RS = 144;                     // LDS bytes per A row (2 planes x 64 B + 16 pad)
constexpr int WROW = 128;                   // packed weight bytes per output channel per iteration
constexpr float SA = 16.0f, SW = 4096.0f;
typedef _Float16 f16x2_g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sat16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f); }
__device__ __forceinline__ void split2(const f32x4 v, uint2& p1, uint2& p2) {
    const float x0 = sat16x(v.x), x1 = sat16x(v.y), x2 = sat16x(v.z), x3 = sat16x(v.w);
    p1.x = cvt_pk_f16(x0, x1);
    p1.y = cvt_pk_f16(x2, x3);
    p2.x = f16_sub_pk(x0, x1, p1.x);
    p2.y = f16_sub_pk(x2, x3, p1.y);
}
}  // namespace g3

typedef _Float16 f16x8_g __attribute__((ext_vector_type(8)));

// Occupancy matters more than anything else for this family (streaming ops, MFMA 7 % busy): the 64-wide kernel is held to
// 128 VGPRs so that four workgroups share a CU (4 x 37 KB LDS).  A deeper A-prefetch ring was measured slower: it costs
// registers (fewer workgroups), and vmcnt retires in order, so the per-iteration wait for the next weight fragments also waits
// for every younger A prefetch.
// VEC: the vector epilogue (out_mode 0 / 2 with N % 4 == 0); !VEC: one dword per (row, channel) -- channels-first output, odd N.
// Two instantiations instead of a run-time branch: with both epilogues in one kernel the 64-wide form needed spilled registers,
// and a spilling build of this kernel has twice given batch-size dependent results at full size.
template <int BN, bool VEC>
__global__ __launch_bounds__(256, (BN == 64 ? 4 : 2)) void igemm3_kernel(IgemmParams p, const unsigned char* __restrict__ wp6) {
    fp16_ovfl_enable();                                 // (common.h: operand conversions saturate in hardware)
    using namespace g3;
    constexpr int NT = BN / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    unsigned char* As0 = smem3;
    unsigned char* As1 = As0 + BM * RS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    int bid = blockIdx.x;
    {
        const int nb = mtiles * ntn, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long m0 = (long long)__builtin_amdgcn_readfirstlane(bid / ntn) * BM;      // wave-uniform: keep it in SGPRs
    const int n0 = __builtin_amdgcn_readfirstlane((bid % ntn) * BN);

    // ---- per-thread A rows: 4 rows (tid/8 + 32 i), one float4 column (tid%8)*4   (identical to igemm.hip)
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int HoWo = p.Ho * p.Wo;
    // (the row -> (frame, y, x) decomposition is redone at every tap change instead of being kept in 16 registers: the
    //  64-wide kernel must fit 128 VGPRs WITHOUT spilling -- a spilling build gave micro-batch-dependent results)
    const int K = p.C0 + p.C1;
    f32x4 ra[4];
    long long roff[4];
    bool rvalid[4];
    int cur_tap = -1;
    auto load_a = [&](int it) {
        const int tap = it / p.kchunks, kc = it - tap * p.kchunks;
        if (tap != cur_tap) {
            cur_tap = tap;
            const int df = p.tdf[tap], dh = p.tdh[tap], dw = p.tdw[tap];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long m = m0 + arow + 32 * i;
                const bool ok = m < p.M;
                const long long mm = ok ? m : 0;
                const int bf = (int)(mm / HoWo);
                const int hw = (int)(mm - (long long)bf * HoWo);
                const int ho = hw / p.Wo;
                const int fi = bf % p.F + df, hi = ho * p.sh + dh, wi = (hw - ho * p.Wo) * p.sw + dw;
                rvalid[i] = ok && (unsigned)fi < (unsigned)p.F && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi;
                roff[i] = ((long long)(bf + df) * p.Hi + hi) * p.Wi + wi;
            }
        }
        const int c = kc * BK + acol;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = p.a0; cs = p.cs0; cc = c; }
        else { src = p.a1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && rvalid[i]) {
                v = *reinterpret_cast<const f32x4*>(src + roff[i] * cs + cc);
                if (p.ln_stats) {
                    const float mean = p.ln_stats[2 * roff[i]], inv = p.ln_stats[2 * roff[i] + 1];
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_gamma + c);
                    v = (v - mean) * inv * g;
                }
            }
            ra[i] = v;
        }
    };
    auto store_a = [&](unsigned char* As) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 p1, p2;
            split2(ra[i] * p.act_scale, p1, p2);
            unsigned char* dst = As + (arow + 32 * i) * RS + acol * 2;
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + 64) = p2;
        }
    };
    // weight fragments: [it][Npad / 32][k-step][plane][half][n 32][16 B] (pack_weights_g6_kernel): lane (n = l31, half hh)
    const unsigned char* wlane = wp6 + (long long)((n0 + wn * (BN / 2)) >> 5) * 4096 + hh * 512 + l31 * 16;
    f16x8_g wc[2][NT][2], wx[2][NT][2];
    auto ldw = [&](int it, f16x8_g (&w)[2][NT][2]) {
        const unsigned char* src = wlane + (long long)it * p.Npad * WROW;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    w[ks][nt][pl] = *reinterpret_cast<const f16x8_g*>(src + nt * 4096 + (ks * 2 + pl) * 1024);
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int a_lane = (wm * 64 + l31) * RS + hh * 16;
    // split-K: this workgroup owns iterations [it0, niter) of the (tap, channel-chunk) sequence
    const int nit_all = p.ntaps * p.kchunks;
    const int nsl = (!VEC && p.ksplit > 1) ? p.ksplit : 1;
    const int it0 = (int)((long long)nit_all * blockIdx.y / nsl), niter = (int)((long long)nit_all * (blockIdx.y + 1) / nsl);
    load_a(it0);
    ldw(it0, wc);
    store_a(As0);
    __syncthreads();
    for (int it = it0; it < niter; ++it) {
        const bool more = it + 1 < niter;
        if (more) { load_a(it + 1); ldw(it + 1, wx); }
        const unsigned char* As = ((it - it0) & 1) ? As1 : As0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8_g a[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    a[mt][pl] = *reinterpret_cast<const f16x8_g*>(As + a_lane + mt * 32 * RS + pl * 64 + ks * 32);
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};     // small terms first
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][PA[term]], wc[ks][nt][PB[term]], acc[mt][nt], 0, 0, 0);
        }
        if (more) {
            store_a(((it - it0) & 1) ? As0 : As1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wc[ks][nt][pl] = wx[ks][nt][pl];
        }
        __syncthreads();
    }
    // ---- epilogue (accumulator layout and output modes of igemm.hip).  Row-major loop: the output row offset -- two integer
    //      divisions in the channel-first and ConvTranspose-parity modes -- is computed once per accumulator row, not per element
    if (!VEC && p.ksplit > 1) {        // raw partial accumulators, [slice][M][N]; finished by igemm3_reduce_kernel (launched on the !VEC form)
        float* pb = p.part + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m >= p.M) continue;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + wn * (BN / 2) + nt * 32 + l31;
                    if (n < p.N) pb[m * p.N + n] = acc[mt][nt][r];
                }
            }
        return;
    }
    // Vector form (out_mode 0 / 2, N % 4 == 0): registers 4g .. 4g+3 of a lane are 4 consecutive rows of ONE channel; a 4 x 4
    // transpose inside each lane quad (two DPP exchange stages, no LDS) turns them into 4 consecutive CHANNELS of one row, so
    // output, residual and the fused GroupNorm input move as dwordx4 (16 contiguous bytes per lane; 4 x fewer memory
    // instructions than one dword per (row, channel): this family is HBM / issue bound, the matrix pipe is ~8 % busy).
    // The K loop keeps the activations as the MFMA *A* operand on purpose -- see DESIGN.md (B-operand hazard): the transposed
    // operand order (activations as B) gave batch-size dependent results under load.
    if constexpr (VEC) {
        const int q3 = l31 & 3;
        const long long bsmp = p.gn_raw ? m0 / p.gn_rows : 0;       // fused GroupNorm-apply residual: the tile lies inside one sample
        auto mrow = [&](int mt, int g) { return m0 + wm * 64 + mt * 32 + 8 * g + 4 * hh + q3; };      // this lane's row after the transpose
        auto orow = [&](int mt, int g) -> long long {
            const long long m = mrow(mt, g);
            if (p.out_mode == 0) return m * p.N;
            const long long mm = m < p.M ? m : 0;
            const long long bf = mm / HoWo;
            const int hw = (int)(mm - bf * HoWo), ho = hw / p.Wo, wo = hw - ho * p.Wo;
            return ((bf * (2 * (HoWo / p.Wo)) + 2 * ho + p.par_a) * (long long)(2 * p.Wo) + 2 * wo + p.par_b) * p.N;
        };
        auto ncol = [&](int nt) { return n0 + wn * (BN / 2) + nt * 32 + (l31 & ~3); };
        igemm_epilogue_vec<2, NT>(p, acc, lane, bsmp, mrow, orow, ncol);
    } else {
    float bv[NT];
    int ncol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ncol[nt] = n0 + wn * (BN / 2) + nt * 32 + l31;
        bv[nt] = (ncol[nt] < p.N && p.bias) ? p.bias[ncol[nt]] : 0.f;
    }
    const long long cstride = p.out_mode == 1 ? (long long)HoWo : 1ll;
    float gmu[NT], gga[NT], gbe[NT];          // fused GroupNorm-apply residual: the tile lies inside one sample
    if (p.gn_raw) {
        const long long bsmp = m0 / p.gn_rows;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = ncol[nt] < p.N ? ncol[nt] : 0;
            const float* cf = p.gn_coef + ((bsmp * (p.N >> 2) + (n >> 2)) * 5) * 4 + (n & 3);
            gmu[nt] = cf[0]; gga[nt] = cf[4]; gbe[nt] = cf[8];
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (m >= p.M) continue;
            long long orow;
            if (p.out_mode == 0) {
                orow = m * p.N;
            } else if (p.out_mode == 1) {
                const long long bf = m / HoWo, hw = m - bf * HoWo;
                orow = bf * p.N * (long long)HoWo + hw;
            } else {
                const long long bf = m / HoWo;
                const int hw = (int)(m - bf * HoWo), ho = hw / p.Wo, wo = hw - ho * p.Wo;
                orow = ((bf * (2 * (HoWo / p.Wo)) + 2 * ho + p.par_a) * (long long)(2 * p.Wo) + 2 * wo + p.par_b) * p.N;
            }
            const float* rrow = p.resid ? p.resid + m * p.N : nullptr;
            const float* grow = p.gn_raw ? p.gn_raw + m * p.N : nullptr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (ncol[nt] >= p.N) continue;
                float v = acc[mt][nt][r] * p.descale + bv[nt];
                if (rrow) v += rrow[ncol[nt]];
                if (grow) {
                    const float y = (grow[ncol[nt]] - gmu[nt]) * gga[nt] + gbe[nt];
                    v += y / (1.0f + expf(-y));
                }
                p.out[orow + ncol[nt] * cstride] = v;
            }
        }
    }
    }   // !VEC
}

// second half of a split-K launch: out = (sum of the slices in index order) * 2^-16 + bias (+ residual), [M][N] layout
__global__ __launch_bounds__(256) void igemm3_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                            const float* __restrict__ resid, float* __restrict__ out, long long MN,
                                                            int N, int nsl, float descale, int* oflag) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + i);
    for (int s = 1; s < nsl; ++s) v += *reinterpret_cast<const f32x4*>(part + (long long)s * MN + i);
    v *= descale;
    if (bias) v += *reinterpret_cast<const f32x4*>(bias + (int)(i % N));
    if (resid) v += *reinterpret_cast<const f32x4*>(resid + i);
    overflow_note4(oflag, v);
    *reinterpret_cast<f32x4*>(out + i) = v;
}

// sized for the larger (bf16x6) layout; the f16x3 layout uses 128 of the 192 bytes per (iteration, n)
size_t igemm6_packed_bytes(int Npad, int K, int ntaps) { return (size_t)ntaps * igemm_kchunks(K) * Npad * 192; }

int launch_igemm6(const IgemmParams& p_in, const void* wp6, hipStream_t s) {
    using namespace g6;
    IgemmParams p = p_in;
    if (p.act_scale == 0.f) p.act_scale = g3::SA;                    // f16x3 kernels: activation scale and its inverse x 2^-12
    p.descale = 1.0f / (p.act_scale * g3::SW);
    p.cs0 = p.a0_stride ? p.a0_stride : p.C0;
    p.oflag = igemm_mode_default() == 2 ? overflow_flag_current() : nullptr;      // the range only exists in the f16x3 mode
    DPC_REQUIRE(p.C0 % 4 == 0 && p.C1 % 4 == 0, "igemm6: channel counts must be multiples of 4");
    DPC_REQUIRE(p.ntaps >= 1 && p.ntaps <= 32, "igemm6: 1..32 taps");
    DPC_REQUIRE(!(p.ln_stats && (p.ntaps != 1 || p.C1 != 0)), "igemm6: LayerNorm prologue needs a 1-tap single-source op");
    DPC_REQUIRE(p.kchunks == igemm_kchunks(p.C0 + p.C1), "igemm6: kchunks mismatch");
    DPC_REQUIRE(wp6 != nullptr, "igemm6: split weights missing");
    if (p.M == 0) return DPC_OK;
    // Dead taps: a tap whose input row (or column) lies outside the image for EVERY output row multiplies zeros only -- on the
    // 1 x 8 images of the Burgers U-Net's deepest level that is 6 of the 9 taps of every 3x3 convolution.  When the live taps are
    // one contiguous range of the pack the launch simply covers that range (bit-identical sums: the skipped terms are exact zeros).
    {
        int lo = p.ntaps, hi = -1;
        bool contiguous = true;
        for (int t = 0; t < p.ntaps; ++t) {
            bool hok = false, wok = false;
            for (int o = 0; o < p.Ho && !hok; ++o) hok = (unsigned)(o * p.sh + p.tdh[t]) < (unsigned)p.Hi;
            for (int o = 0; o < p.Wo && !wok; ++o) wok = (unsigned)(o * p.sw + p.tdw[t]) < (unsigned)p.Wi;
            if (hok && wok) {
                if (hi >= 0 && t != hi + 1) contiguous = false;
                lo = std::min(lo, t);
                hi = t;
            }
        }
        if (contiguous && hi >= lo && (lo > 0 || hi < p.ntaps - 1)) {
            const int n = hi - lo + 1;
            for (int t = 0; t < n; ++t) { p.tdf[t] = p.tdf[lo + t]; p.tdh[t] = p.tdh[lo + t]; p.tdw[t] = p.tdw[lo + t]; }
            wp6 = static_cast<const unsigned char*>(wp6) + (size_t)lo * p.kchunks * p.Npad * (igemm_mode_default() == 2 ? g3::WROW : 192);   // bytes per (iteration, n)
            p.ntaps = n;
        }
    }
    const int mtiles = (int)((p.M + BM - 1) / BM);
    static const int log_shapes = debug_switch("DPC_IGEMM_LOG", 0);
    if (log_shapes)
        fprintf(stderr, "igemm M=%lld C0=%d C1=%d N=%d taps=%d(%d) img=%dx%d->%dx%d F=%d out_mode=%d resid=%d ln=%d gn=%d a0s=%d\n", p.M, p.C0, p.C1,
                p.N, p.ntaps, p_in.ntaps, p.Hi, p.Wi, p.Ho, p.Wo, p.F, p.out_mode, p.resid != nullptr, p.ln_stats != nullptr, p.gn_raw != nullptr,
                p.a0_stride);
    const double flops = 2.0 * (double)p.M * p.N * (double)p_in.ntaps * (p.C0 + p.C1);      // algorithmic: all taps of the operator
    const double bytes = 4.0 * ((double)p.M * (p.N + (p.resid ? p.N : 0)) + (double)p.BF * p.Hi * p.Wi * (p.C0 + p.C1) +
                                (double)p.ntaps * (p.C0 + p.C1) * p.N);
    // few row tiles (deep U-Net levels: 2048 rows x 1024 channels in the Burgers POPC net): prefer 64-wide column tiles
    // so that the launch still covers the 256 CUs at least twice
    // 64-wide tiles at four workgroups per CU beat the 128-wide tile (two per CU) on every shape of both workloads
    // (smoke 31.4 -> 24.9 ms/step, Burgers 24.2 -> 22.0): DPC_IGEMM_WIDE=1 re-enables the wide tile for A/B runs
    static const int wide_ok = debug_switch("DPC_IGEMM_WIDE", 0);
    const bool wide = wide_ok && !p.gn_raw && p.Npad % 128 == 0 && p.N > 64 && (long long)mtiles * (p.Npad / 128) >= 512;
    ProfScope prof((p.Npad % 128 == 0 && p.N > 64) ? PROF_IGEMM128 : PROF_IGEMM64, flops, bytes, s);
    if (igemm_mode_default() == 2) {
        const size_t lds3 = 2 * (size_t)g3::BM * g3::RS;
        if (igemm3t_supported(p)) return launch_igemm3t(p, wp6, s);      // stride-1 2 x 2-tap classes: unique pixels staged once (igemm_tile.hip)
        if (igemm3p_supported(p)) return launch_igemm3p(p, wp6, s);      // 1-tap, K <= 256: 64-row panels over all N (igemm_panel.hip)
        // split-K for the GEMM-shaped deep levels of the 2-D U-Net (K x taps >= 4096: few row tiles, hundreds of iterations;
        // 256 workgroups leave three quarters of the 4-per-CU slots empty): four slices of the iteration range run as
        // separate workgroups, a second kernel adds them in fixed order.  The rule looks at the reduction length only, never
        // at the batch, so a trajectory's result does not depend on how the batch is sharded or micro-batched.  The scratch
        // buffer is grown on first use (warm-up), never inside a steady-state step.
        static const int split_ok = debug_switch("DPC_IGEMM_SPLITK", 1);
        const long long nwg = (long long)mtiles * (p.Npad / 64);
        const int nit = p.ntaps * p.kchunks;
        const bool img = igemm3i_supported(p);                       // 3 x 3 on small images: unique pixels staged once per channel block (igemm_img.hip)
        const bool lds_b = !img && igemm3w_supported(p);             // 256 x 128 tiles, both operands through LDS (igemm_wide.hip)
        int nsl = 1;
        if (img || lds_b) nsl = split_ok ? igemm3w_slices(p) : 1;
        else if (split_ok && !wide && p.out_mode == 0 && !p.ln_stats && !p.gn_raw && p.N % 4 == 0 && nit >= 128) nsl = 4;
        DPC_REQUIRE(!p.gn_raw || (p.out_mode == 0 && !wide && p.N % 4 == 0 && p.gn_rows % 128 == 0),
                    "igemm3: fused GroupNorm residual needs out_mode 0, N % 4 == 0, rows per sample % 128 == 0");
        if (nsl > 1) {
            // partial sums: scratch per (device, stream), grown on first use (warm-up), never inside a steady-state step
            float* scratch = nullptr;
            if (int rc = stream_scratch(SCRATCH_SPLITK, s, (size_t)nsl * p.M * p.N * sizeof(float), &scratch)) return rc;
            IgemmParams q = p;
            q.ksplit = nsl;
            q.part = scratch;
            if (img) {
                if (int rc = launch_igemm3i(q, wp6, nsl, s)) return rc;
            } else if (lds_b) {
                if (int rc = launch_igemm3w(q, wp6, nsl, s)) return rc;
            } else {
                hipLaunchKernelGGL((igemm3_kernel<64, false>), dim3((unsigned)nwg, nsl), dim3(256), lds3, s, q, (const unsigned char*)wp6);
                DPC_LAUNCH_CHECK();
            }
            const long long MN = p.M * p.N;
            hipLaunchKernelGGL(igemm3_reduce_kernel, dim3((unsigned)((MN / 4 + 255) / 256)), dim3(256), 0, s, scratch, p.bias, p.resid,
                               p.out, MN, p.N, nsl, p.descale, p.oflag);
            DPC_LAUNCH_CHECK();
            return DPC_OK;
        }
        if (img) return launch_igemm3i(p, wp6, 1, s);
        if (lds_b) return launch_igemm3w(p, wp6, 1, s);
        const bool vec = (p.N & 3) == 0 && p.out_mode != 1;
        if (wide) {
            if (vec) hipLaunchKernelGGL((igemm3_kernel<128, true>), dim3(mtiles * (p.Npad / 128)), dim3(256), lds3, s, p, (const unsigned char*)wp6);
            else hipLaunchKernelGGL((igemm3_kernel<128, false>), dim3(mtiles * (p.Npad / 128)), dim3(256), lds3, s, p, (const unsigned char*)wp6);
        } else {
            DPC_REQUIRE(p.Npad % 64 == 0, "igemm3: Npad must be a multiple of 64");
            if (vec) hipLaunchKernelGGL((igemm3_kernel<64, true>), dim3(mtiles * (p.Npad / 64)), dim3(256), lds3, s, p, (const unsigned char*)wp6);
            else hipLaunchKernelGGL((igemm3_kernel<64, false>), dim3(mtiles * (p.Npad / 64)), dim3(256), lds3, s, p, (const unsigned char*)wp6);
        }
        DPC_LAUNCH_CHECK();
        return DPC_OK;
    }
    const size_t lds = 2 * (size_t)BM * RS;
    if (wide) {
        hipLaunchKernelGGL(igemm6_kernel<128>, dim3(mtiles * (p.Npad / 128)), dim3(256), lds, s, p, (const unsigned char*)wp6);
    } else {
        DPC_REQUIRE(p.Npad % 64 == 0, "igemm6: Npad must be a multiple of 64");
        hipLaunchKernelGGL(igemm6_kernel<64>, dim3(mtiles * (p.Npad / 64)), dim3(256), lds, s, p, (const unsigned char*)wp6);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ---- weight pre-split: wp6[it = tap*kchunks + kc][n][plane][kk] (bf16) from w[n*stride_n + c*stride_c + tap_off[tap]]
struct PackTaps6 { int off[32]; };
__global__ void pack_weights_g6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int N, int Npad, int K,
                                       int kchunks, int ntaps, long long stride_n, long long stride_c, PackTaps6 t, int f16x3,
                                       int* __restrict__ ovf) {
    const long long total = (long long)ntaps * kchunks * Npad * 32;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % 32);
        long long r = i / 32;
        const int n = (int)(r % Npad);
        r /= Npad;
        const int kc = (int)(r % kchunks);
        const int tap = (int)(r / kchunks);
        const int c = kc * 32 + kk;
        float v = 0.f;
        if (n < N && c < K) v = w[n * stride_n + c * stride_c + t.off[tap]];
        if (f16x3) {
            v = v * g3::SW;
            if (!(fabsf(v) <= 65504.f)) atomicOr(ovf, 1);
            v = g3::sat16(v);
            const unsigned h1 = g3::cvt_pk_f16(v, 0.f) & 0xffffu;
            const unsigned h2 = g3::cvt_pk_f16(v - (float)__builtin_bit_cast(g3::f16x2_g, h1).x, 0.f) & 0xffffu;
            // fragment order (r03): per iteration [32-column block][k-step 2][plane 2][half 2][n 32][8 fp16] -- the 16 bytes lane
            // (n, half) feeds to one MFMA are contiguous ACROSS the wave (1 KB per fragment load = 8 cache lines instead of 32 lines
            // of the former [n][plane][32 k] rows), and a 32-column block's four fragments are one 4 KB run (igemm_wide's LDS image)
            const int ks = kk >> 4, hf = (kk >> 3) & 1, j = kk & 7;
            unsigned short* d3 = wp + ((long long)tap * kchunks + kc) * Npad * 64 + ((n >> 5) * 4 + ks * 2) * 512 + hf * 256 + (n & 31) * 8 + j;
            d3[0] = (unsigned short)h1;
            d3[512] = (unsigned short)h2;
            continue;
        }
        const unsigned p1 = g6::cvt_pk_bf16(v, 0.f) & 0xffffu;
        const float r1 = v - __uint_as_float(p1 << 16);
        const unsigned p2 = g6::cvt_pk_bf16(r1, 0.f) & 0xffffu;
        const float r2 = r1 - __uint_as_float(p2 << 16);
        const unsigned p3 = g6::cvt_pk_bf16(r2, 0.f) & 0xffffu;
        unsigned short* dst = wp + (((long long)tap * kchunks + kc) * Npad + n) * 96 + kk;
        dst[0] = (unsigned short)p1;
        dst[32] = (unsigned short)p2;
        dst[64] = (unsigned short)p3;
    }
}

int launch_pack_weights_g6(const float* w, void* wp6, int N, int Npad, int K, int ntaps, long long stride_n,
                           long long stride_c, const int* tap_off_host, hipStream_t s) {
    DPC_REQUIRE(ntaps <= 32, "pack6: at most 32 taps");
    PackTaps6 t;
    for (int i = 0; i < 32; ++i) t.off[i] = i < ntaps ? tap_off_host[i] : 0;
    const int kchunks = igemm_kchunks(K);
    const long long total = (long long)ntaps * kchunks * Npad * 32;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_g6_kernel, dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp6), N, Npad,
                       K, kchunks, ntaps, stride_n, stride_c, t, igemm_mode_default() == 2 ? 1 : 0, f16x3_weight_overflow_flag());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
