// LDS-tiled 256 x 128 form of the f16x3 implicit GEMM (igemm6.hip: igemm3_kernel) for the GEMM-shaped deep levels of the Burgers
// U-Net (model/burgers_1d/unet.py:387-431: 3x3 convolutions at 4x32 / 2x16 / 1x8 images, M = 2-32 k rows, K = 9 C = 2304-18432, N =
// 256-1024) and every other launch with a long reduction and N >= 128.  igemm3's 128 x 64 / 128 x 128 tiles stream each wave's
// weight fragments from L2 (22 FLOP per byte moved into the CU): those levels ran L2-bandwidth bound at ~150 TF/s.  Here a
// 512-thread workgroup (8 waves as 4 x 2, 64 x 64 accumulators each) stages BOTH operands of a 32-channel chunk through LDS --
// activations converted and split on the way in, as igemm3 does; the pre-split weights of the 128-column tile copied in MFMA
// fragment order ([k-step][plane][n][half][16 B]: conflict-free b128 reads) -- so a byte fetched into the CU feeds 44 FLOP.
// Both LDS images are double-buffered (one barrier per chunk); the B fragments of the two k-steps of a chunk live in SEPARATE
// registers (no ds_read ever targets a register an in-flight MFMA still reads: DESIGN.md 6.2).  Same arithmetic, operand scales,
// partial-product order, split-K protocol and epilogues as igemm3_kernel; the reduction is walked in the same (tap, chunk) order.
#include <algorithm>

#include "common.h"
#include "f16x3.h"

namespace dpc {

namespace gw {
constexpr int BM = 256, BN = 128, BK = 32;
constexpr int RS = 144;                     // LDS bytes per A row (2 planes x 64 B + 16 pad), as igemm3
constexpr int WROW = 128;                   // packed weight bytes per output channel per iteration
constexpr int ABYTES = BM * RS, BBYTES = BN * WROW;
constexpr int LDS = 2 * (ABYTES + BBYTES);
}  // namespace gw

typedef _Float16 f16x8_w __attribute__((ext_vector_type(8)));

template <bool VEC>
__global__ __launch_bounds__(512, 1) void igemm3w_kernel(IgemmParams p, const unsigned char* __restrict__ wp6) {
    using namespace gw;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    auto Abuf = [&](int i) { return smem_w + i * ABYTES; };
    auto Bbuf = [&](int i) { return smem_w + 2 * ABYTES + i * BBYTES; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    int bid = blockIdx.x;
    {
        const int nb = mtiles * ntn, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long m0 = (long long)__builtin_amdgcn_readfirstlane(bid / ntn) * BM;
    const int n0 = __builtin_amdgcn_readfirstlane((bid % ntn) * BN);
    // ---- A: thread -> 4 rows (tid / 8 + 64 i), one float4 column (tid % 8) * 4
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int HoWo = p.Ho * p.Wo;
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
                const long long m = m0 + arow + 64 * i;
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
            if (cok && rvalid[i]) v = *reinterpret_cast<const f32x4*>(src + roff[i] * cs + cc);
            ra[i] = v;
        }
    };
    auto store_a = [&](unsigned char* A) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = ra[i] * p.act_scale;
            uint2 p1, p2;
            const float x0 = h3::sat16(v.x), x1 = h3::sat16(v.y), x2 = h3::sat16(v.z), x3 = h3::sat16(v.w);
            p1.x = h3::cvt_pk(x0, x1); p1.y = h3::cvt_pk(x2, x3);
            p2.x = f16_sub_pk(x0, x1, p1.x); p2.y = f16_sub_pk(x2, x3, p1.y);
            unsigned char* dst = A + (arow + 64 * i) * RS + acol * 2;
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + 64) = p2;
        }
    };
    // ---- B: thread -> column n = tid / 4, 32-byte part (tid % 4) = plane * 2 + k-step of its 128-byte row
    const int bn = tid >> 2, bpart = tid & 3;
    uint4 rb[2];
    auto load_b = [&](int it) {
        const unsigned char* src = wp6 + ((long long)it * p.Npad + n0 + bn) * WROW + bpart * 32;
        rb[0] = *reinterpret_cast<const uint4*>(src);
        rb[1] = *reinterpret_cast<const uint4*>(src + 16);
    };
    auto store_b = [&](unsigned char* B) {
        const int pl = bpart >> 1, ks = bpart & 1;
        unsigned char* dst = B + ((ks * 2 + pl) * BN + bn) * 32;
        *reinterpret_cast<uint4*>(dst) = rb[0];
        *reinterpret_cast<uint4*>(dst + 16) = rb[1];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const int a_lane = (wm * 64 + l31) * RS + hh * 16;
    const int b_lane = (wn * 64 + l31) * 32 + hh * 16;
    const int nit_all = p.ntaps * p.kchunks;
    const int nsl = (!VEC && p.ksplit > 1) ? p.ksplit : 1;
    const int it0 = (int)((long long)nit_all * blockIdx.y / nsl), niter = (int)((long long)nit_all * (blockIdx.y + 1) / nsl);
    load_a(it0);
    load_b(it0);
    store_a(Abuf(0));
    store_b(Bbuf(0));
    __syncthreads();
    for (int it = it0; it < niter; ++it) {
        const bool more = it + 1 < niter;
        if (more) { load_a(it + 1); load_b(it + 1); }
        const unsigned char* A = Abuf((it - it0) & 1);
        const unsigned char* B = Bbuf((it - it0) & 1);
        f16x8_w b[2][2][2];                  // [k-step][nt][plane]: both k-steps' fragments in separate registers
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    b[ks][nt][pl] = *reinterpret_cast<const f16x8_w*>(B + ((ks * 2 + pl) * BN + nt * 32) * 32 + b_lane);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8_w a[2][2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    a[mt][pl] = *reinterpret_cast<const f16x8_w*>(A + a_lane + mt * 32 * RS + pl * 64 + ks * 32);
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};     // small terms first (as igemm3)
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][PA[term]], b[ks][nt][PB[term]], acc[mt][nt], 0, 0, 0);
        }
        if (more) {
            store_a(Abuf(((it - it0) & 1) ^ 1));
            store_b(Bbuf(((it - it0) & 1) ^ 1));
        }
        __syncthreads();
    }
    // ---- epilogue
    if (!VEC && p.ksplit > 1) {        // raw partial accumulators [slice][M][N]; finished by igemm3_reduce_kernel
        float* pb = p.part + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m >= p.M) continue;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = n0 + wn * 64 + nt * 32 + l31;
                    if (n < p.N) pb[m * p.N + n] = acc[mt][nt][r];
                }
            }
        return;
    }
    if constexpr (VEC) {
        const int q3 = l31 & 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long long m = m0 + wm * 64 + mt * 32 + 8 * g + 4 * hh + q3;
                const long long orow = m * p.N;                   // (out_mode 0 only: see igemm3w_supported)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = acc[mt][nt][4 * g + e];
                    {
                        const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
#define DPC_QUAD_XCHG(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
                        const float r0 = DPC_QUAD_XCHG(b0 ? x[0] : x[1], 0xB1), r1 = DPC_QUAD_XCHG(b0 ? x[2] : x[3], 0xB1);
                        const float y0 = b0 ? r0 : x[0], y1 = b0 ? x[1] : r0, y2 = b0 ? r1 : x[2], y3 = b0 ? x[3] : r1;
                        const float s0 = DPC_QUAD_XCHG(b1 ? y0 : y2, 0x4E), s1 = DPC_QUAD_XCHG(b1 ? y1 : y3, 0x4E);
#undef DPC_QUAD_XCHG
                        x[0] = b1 ? s0 : y0; x[2] = b1 ? y2 : s0; x[1] = b1 ? s1 : y1; x[3] = b1 ? y3 : s1;
                    }
                    const int n = n0 + wn * 64 + nt * 32 + (l31 & ~3);
                    if (m >= p.M || n >= p.N) continue;
                    f32x4 v = f32x4{x[0], x[1], x[2], x[3]} * p.descale;
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (p.resid) v += *reinterpret_cast<const f32x4*>(p.resid + m * p.N + n);
                    overflow_note4(p.oflag, v);
                    *reinterpret_cast<f32x4*>(p.out + orow + n) = v;
                }
            }
    }
}

// shape-only rule (never the batch): long reductions into >= 128 columns, plain [M][N] output
bool igemm3w_supported(const IgemmParams& p) {
    static const int on = debug_switch("DPC_IGEMM_LDSB", 1);
    const int nit = p.ntaps * p.kchunks;
    return on && p.out_mode == 0 && !p.ln_stats && !p.gn_raw && !p.a0_stride && p.N % 4 == 0 && p.Npad % 128 == 0 && p.N >= 128 && nit >= 64;
}

int launch_igemm3w(const IgemmParams& p, const void* wp6, int nsl, hipStream_t s) {
    using namespace gw;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    const unsigned nwg = (unsigned)mtiles * (p.Npad / BN);
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        once = true;
    }
    if (nsl > 1) hipLaunchKernelGGL(igemm3w_kernel<false>, dim3(nwg, nsl), dim3(512), LDS, s, p, (const unsigned char*)wp6);
    else hipLaunchKernelGGL(igemm3w_kernel<true>, dim3(nwg), dim3(512), LDS, s, p, (const unsigned char*)wp6);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
