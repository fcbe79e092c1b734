// Conv3d 3x3x3 / stride 1 / pad 1, channels-last, fp32 semantics on the bf16 matrix cores ("bf16x6").
//
// gfx950 has no TF32-like mode and its fp32 MFMA runs at 1/16 of the bf16 rate.  Here every fp32 operand is split
// exactly into three bf16 terms  a = a1 + a2 + a3  (8 + 8 + 8 mantissa bits, |a3| <= 2^-16 |a|), and a product is
// evaluated as the six partial products of relative weight >= 2^-16
//        a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are
// <= 2^-24 relative, i.e. below fp32 rounding; measured against fp64 the result is as accurate as the native fp32
// MFMA chain (tests/test_gpu_ops.py::test_conv3d_cl, same tolerance; tools/bf16x6_error.py).  6 bf16 MFMAs of K=16
// cost 6*32 cycles against 8*64 cycles for the fp32 MFMA over the same 16 channels: 2.67x the fp32-MFMA rate.
//
// Data flow = conv3h.hip: a workgroup owns a 4x4x8 output tile; per 16-channel chunk the 6x6x10 halo is staged once
// in LDS -- already split into the three bf16 planes (112 B per point: 3 x 32 B + 16 B pad => conflict-free
// ds_read_b128 with the pitch-12 / lane_hw layout) -- and the 27 taps are LDS offsets.  Weights are pre-split at load
// time ([tap][chunk][n][3 planes][16] bf16) and double-buffered through LDS one tap ahead.
// Reference op: nn.Conv3d(dim, dim_out, (3,3,3), padding=(1,1,1)) in Block (video_diffusion_pytorch_conv3d.py:192).
#include <type_traits>

#include "common.h"

namespace dpc {

namespace x6 {
constexpr int TF = 4, TH = 4, TW = 8;
constexpr int HF = TF + 2, HH = TH + 2, HWL = TW + 2, HWD = 12;
constexpr int NLOG = HF * HH * HWL;        // 360 halo points
constexpr int NSLOT = HF * HH * HWD;       // 432 LDS slots
constexpr int KC = 16;
constexpr int PST = 112;                   // bytes per halo point / weight row in LDS (3 planes x 32 B + 16 pad)
constexpr int HLOADS = (NLOG * 4 + 255) / 256;

__device__ __forceinline__ void lane_hw(int i, int& h, int& w) {      // see conv3h.hip
    if (i < 4) { h = 0; w = i; }
    else if (i < 12) { h = 1; w = i - 4; }
    else if (i < 16) { h = 0; w = i - 8; }
    else if (i < 20) { h = 3; w = i - 16; }
    else if (i < 28) { h = 2; w = i - 20; }
    else { h = 3; w = i - 24; }
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf16_lo_to_f32(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf16_hi_to_f32(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// exact 3-way split of 4 floats -> three (2 x u32) packs of 4 bf16 each
__device__ __forceinline__ void split3(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    p1.x = cvt_pk_bf16(v.x, v.y);
    p1.y = cvt_pk_bf16(v.z, v.w);
    const float r0 = v.x - bf16_lo_to_f32(p1.x), r1 = v.y - bf16_hi_to_f32(p1.x);
    const float r2 = v.z - bf16_lo_to_f32(p1.y), r3 = v.w - bf16_hi_to_f32(p1.y);
    p2.x = cvt_pk_bf16(r0, r1);
    p2.y = cvt_pk_bf16(r2, r3);
    const float s0 = r0 - bf16_lo_to_f32(p2.x), s1 = r1 - bf16_hi_to_f32(p2.x);
    const float s2 = r2 - bf16_lo_to_f32(p2.y), s3 = r3 - bf16_hi_to_f32(p2.y);
    p3.x = cvt_pk_bf16(s0, s1);
    p3.y = cvt_pk_bf16(s2, s3);
}
}  // namespace x6

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));


// ---- explicit asynchronous loads: the compiler's scheduler sinks ordinary prefetch loads down to their first use
// (measured: `s_waitcnt vmcnt(0)` a dozen MFMAs after the issue), so the streaming loads of the pipelined kernels are
// issued with inline asm and retired with hand-counted `s_waitcnt vmcnt(N)`; the waited-for registers are in/out
// operands of the wait so that no consumer can be scheduled above it.
template <int OFF>
__device__ __forceinline__ void gload16(bf16x8& dst, const unsigned char* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gload16f(f32x4& dst, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm3(bf16x8& a, bf16x8& b, bf16x8& c) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm6f(f32x4& a, f32x4& b, f32x4& c, f32x4& d, f32x4& e, f32x4& f) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}

template <int BN, int BDIRECT>
__global__ __launch_bounds__(256, 2) void conv3x6_kernel(Conv3hParams p) {     // 3 waves/SIMD would spill (168 VGPR cap)     // LDS (63/77 KB) admits 2 workgroups/CU
    using namespace x6;
    constexpr int NT = BN / 64;
    constexpr int BQ = BN * 12 / 256;      // 8-byte weight pieces per thread per (tap, chunk): 3 (BN=64) or 6
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;                         // [NSLOT][PST]
    unsigned char* Bs0 = halo + NSLOT * PST;            // [BN][PST]
    unsigned char* Bs1 = Bs0 + BN * PST;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int ntf = (p.F + TF - 1) / TF, nth = (p.H + TH - 1) / TH, ntw = (p.W + TW - 1) / TW;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = (bid % ntn) * BN;
    int t = bid / ntn;
    const int w0 = (t % ntw) * TW; t /= ntw;
    const int h0 = (t % nth) * TH; t /= nth;
    const int f0 = (t % ntf) * TF;
    const int b = t / ntf;
    const int K = p.C0 + p.C1;

    long long hoff[HLOADS];
    bool hok[HLOADS];
    int hdst[HLOADS];
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
        const int q = tid + 256 * i;
        const int pt = q >> 2;
        const int pf = pt / (HH * HWL), ph = (pt / HWL) % HH, pw = pt % HWL;
        const int f = f0 - 1 + pf, h = h0 - 1 + ph, w = w0 - 1 + pw;
        hok[i] = pt < NLOG && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        hoff[i] = (((long long)b * p.F + f) * p.H + h) * p.W + w;
        if (p.dbg & 8) hoff[i] = (((long long)0 * p.F + (pf + 1)) * p.H + (ph + 1)) * p.W + (pw + 1);   // perf attribution: L2-resident halo
        hdst[i] = ((pt / HWL) * HWD + pt % HWL) * PST + (q & 3) * 8;      // + plane*32
    }
    const int hslot = (tid & 3) * 4;

    f32x4 hreg[HLOADS];
    auto load_halo = [&](int kc) {
        const int c = kc * KC + hslot;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = p.a0; cs = p.C0; cc = c; }
        else { src = p.a1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && hok[i] && !(p.dbg & 2)) v = *reinterpret_cast<const f32x4*>(src + hoff[i] * cs + cc);
            hreg[i] = v;
        }
    };
    auto store_halo = [&](int kc) {
        if (p.in_coef) {
            // fused GroupNorm -> (scale + 1, shift) -> SiLU of the producer (Block.forward, ...conv3d.py:196-204), applied
            // when the prefetched chunk is split into LDS; the zero padding of the convolution applies to the ACTIVATED
            // tensor, so out-of-range points stay 0
            const int c = kc * KC + hslot;
            if (c < K) {
                const f32x4* cf = reinterpret_cast<const f32x4*>(p.in_coef) + ((long long)b * (K >> 2) + (c >> 2)) * 5;
                const f32x4 mu = cf[0], ga = cf[1], be = cf[2], sc = cf[3], sh = cf[4];
#pragma unroll
                for (int i = 0; i < HLOADS; ++i) {
                    if (hok[i]) {
                        f32x4 y = (hreg[i] - mu) * ga + be;
                        y = y * sc + sh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.0f + expf(-y[e]));
                        hreg[i] = y;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            if (tid + 256 * i < NLOG * 4) {
                uint2 p1, p2, p3;
                split3(hreg[i], p1, p2, p3);
                *reinterpret_cast<uint2*>(halo + hdst[i]) = p1;
                *reinterpret_cast<uint2*>(halo + hdst[i] + 32) = p2;
                *reinterpret_cast<uint2*>(halo + hdst[i] + 64) = p3;
            }
        }
    };
    // weights: packed [tap][kc][n][3][16] bf16 = 96 B per n; 8-byte piece e = tid + 256 i -> row e/12, piece e%12
    uint2 breg[BQ];
    auto load_b = [&](int tap, int kc) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wp) +
                                   (((long long)tap * p.kchunks + kc) * p.Npad + n0) * 96;
#pragma unroll
        for (int i = 0; i < BQ; ++i) breg[i] = *reinterpret_cast<const uint2*>(src + (long long)(tid + 256 * i) * 8);
    };
    auto store_b = [&](unsigned char* Bs) {
#pragma unroll
        for (int i = 0; i < BQ; ++i) {
            const int e = tid + 256 * i;
            *reinterpret_cast<uint2*>(Bs + (e / 12) * PST + (e % 12) * 8) = breg[i];
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    int lh, lw;
    lane_hw(l31, lh, lw);
    const int a_lane = (((wm * 2) * HH + lh) * HWD + lw) * PST + hh * 16;
    const int b_lane = (wn * (BN / 2) + l31) * PST + hh * 16;

    if constexpr (BDIRECT == 4) {
        // v4 = v2 with hand-scheduled memory pipeline (see gload16 / wait_vm3): weights two taps ahead in a 3-deep ring,
        // next-chunk halo issued at tap 0 behind the tap-2 weights, A fragments one tap ahead, vmcnt counted by hand.
        static_assert(HLOADS == 6, "wait_vm6f assumes 6 halo loads per thread");
        constexpr int LW = 3 * NT;                     // global loads per ldw
        const unsigned char* wlane = reinterpret_cast<const unsigned char*>(p.wp) +
                                     ((long long)n0 + wn * (BN / 2) + l31) * 96 + hh * 16;
        const long long wstep = (long long)p.Npad * 96;                  // bytes per (tap, chunk)
        bf16x8 w[3][NT][3];
        bf16x8 a[2][2][3];
        auto ldw = [&](int tap, int kc, bf16x8 (&dst)[NT][3]) {
            const unsigned char* src = wlane + ((long long)tap * p.kchunks + kc) * wstep;
            gload16<0>(dst[0][0], src); gload16<32>(dst[0][1], src); gload16<64>(dst[0][2], src);
            if constexpr (NT == 2) { gload16<3072>(dst[1][0], src); gload16<3104>(dst[1][1], src); gload16<3136>(dst[1][2], src); }
        };
        auto waitw = [&](auto n, bf16x8 (&r)[NT][3]) {
            constexpr int N = decltype(n)::value;
            if constexpr (NT == 2) wait_vm3<N + 3>(r[0][0], r[0][1], r[0][2]);      // first half may retire with 3 more in flight
            wait_vm3<N>(r[NT - 1][0], r[NT - 1][1], r[NT - 1][2]);
        };
        // halo loads: out-of-range / out-of-channel lanes read element 0 and are zeroed after the wait
        bool hz[HLOADS];
        auto load_halo_async = [&](int kc) {
            const int c = kc * KC + hslot;
            const float* src;
            int cs, cc;
            if (c < p.C0) { src = p.a0; cs = p.C0; cc = c; }
            else { src = p.a1; cs = p.C1; cc = c - p.C0; }
            const bool cok = c < K;
#pragma unroll
            for (int i = 0; i < HLOADS; ++i) {
                hz[i] = !(cok && hok[i]);
                gload16f(hreg[i], hz[i] ? p.a0 : src + hoff[i] * cs + cc);
            }
        };
        auto halo_fix = [&]() {
#pragma unroll
            for (int i = 0; i < HLOADS; ++i)
                if (hz[i]) hreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto lda = [&](int tap, bf16x8 (&dst)[2][3]) {
            const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * PST;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    dst[mt][pl] = *reinterpret_cast<const bf16x8*>(halo + aoff + mt * (HH * HWD * PST) + pl * 32);
        };
        load_halo_async(0);
        ldw(0, 0, w[0]);
        ldw(1, 0, w[1]);
        wait_vm6f<2 * LW>(hreg[0], hreg[1], hreg[2], hreg[3], hreg[4], hreg[5]);
        halo_fix();
        store_halo(0);
        __syncthreads();
        lda(0, a[0]);
        for (int kc = 0; kc < p.kchunks; ++kc) {
            const bool more_kc = kc + 1 < p.kchunks;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int t2 = tap + 2;
                if (t2 < 27) ldw(t2, kc, w[t2 % 3]);
                else if (more_kc) ldw(t2 - 27, kc + 1, w[t2 % 3]);
                if (tap == 0 && more_kc) load_halo_async(kc + 1);
                if (tap < 26) lda(tap + 1, a[(tap + 1) & 1]);
                // retire w[tap % 3]: count the vector loads issued after it
                if (more_kc) {
                    if (tap <= 2) waitw(std::integral_constant<int, 2 * LW + HLOADS>{}, w[tap % 3]);
                    else if (tap == 3) {          // the halo loads are older than the tap-3 weights: they retire here
                        waitw(std::integral_constant<int, 2 * LW>{}, w[tap % 3]);
                        wait_vm6f<2 * LW>(hreg[0], hreg[1], hreg[2], hreg[3], hreg[4], hreg[5]);
                    } else waitw(std::integral_constant<int, 2 * LW>{}, w[tap % 3]);
                } else {
                    if (tap <= 24) waitw(std::integral_constant<int, 2 * LW>{}, w[tap % 3]);
                    else if (tap == 25) waitw(std::integral_constant<int, LW>{}, w[tap % 3]);
                    else waitw(std::integral_constant<int, 0>{}, w[tap % 3]);
                }
                constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tap & 1][mt][PA[term]], w[tap % 3][nt][PB[term]],
                                                                                  acc[mt][nt], 0, 0, 0);
            }
            if (more_kc) {
                halo_fix();
                __syncthreads();
                store_halo(kc + 1);
                __syncthreads();
                lda(0, a[0]);
            }
        }
    } else if constexpr (BDIRECT == 2) {
        // v2: weight fragments straight from L2/L1 into a 3-deep register ring (TWO taps ahead; 27 % 3 == 0 keeps the
        // ring index static across channel chunks), A fragments double-buffered one tap ahead, the 27 taps fully
        // unrolled (tap offsets are ds_read immediates), and the 6 split-product MFMAs of the 2*NT accumulators issued
        // round-robin so that no MFMA depends on its predecessor.  The halo prefetch of the next chunk is issued right
        // after the weight load of tap 2, so the in-order vmcnt only couples it to loads that are needed >= 2 taps later.
        const unsigned char* wlane = reinterpret_cast<const unsigned char*>(p.wp) +
                                     ((long long)n0 + wn * (BN / 2) + l31) * 96 + hh * 16;
        bf16x8 w[3][NT][3];
        bf16x8 a[2][2][3];
        auto ldw = [&](int tap, int kc, bf16x8 (&dst)[NT][3]) {
            const unsigned char* src = wlane + ((p.dbg & 4) ? 0ll : ((long long)tap * p.kchunks + kc) * p.Npad * 96);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) dst[nt][pl] = *reinterpret_cast<const bf16x8*>(src + nt * 32 * 96 + pl * 32);
        };
        auto lda = [&](int tap, bf16x8 (&dst)[2][3]) {
            const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * PST;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    dst[mt][pl] = *reinterpret_cast<const bf16x8*>(halo + aoff + mt * (HH * HWD * PST) + pl * 32);
        };
        load_halo(0);
        ldw(0, 0, w[0]);
        if (p.kchunks > 0) ldw(1, 0, w[1]);
        store_halo(0);
        __syncthreads();
        lda(0, a[0]);
        for (int kc = 0; kc < p.kchunks; ++kc) {
            const bool more_kc = kc + 1 < p.kchunks;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int t2 = tap + 2;
                if (t2 < 27) ldw(t2, kc, w[t2 % 3]);
                else if (more_kc) ldw(t2 - 27, kc + 1, w[t2 % 3]);
                if (tap == 0 && more_kc) load_halo(kc + 1);
                if (tap < 26) lda(tap + 1, a[(tap + 1) & 1]);
                asm volatile("" ::: "memory");      // pin the prefetches HERE (the scheduler otherwise sinks them to their use)
                constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tap & 1][mt][PA[term]], w[tap % 3][nt][PB[term]],
                                                                                  acc[mt][nt], 0, 0, 0);
                asm volatile("" ::: "memory");
            }
            if (more_kc) {
                __syncthreads();
                store_halo(kc + 1);
                __syncthreads();
                lda(0, a[0]);
            }
        }
    } else if constexpr (BDIRECT == 1) {
        // weight fragments straight from L2/L1 into registers, one tap ahead; no barrier inside the tap loop
        const unsigned char* wlane = reinterpret_cast<const unsigned char*>(p.wp) +
                                     ((long long)n0 + wn * (BN / 2) + l31) * 96 + hh * 16;
        bf16x8 wc[NT][3], wnx[NT][3];
        auto ldw = [&](int tap, int kc, bf16x8 (&w)[NT][3]) {
            const unsigned char* src = wlane + ((p.dbg & 4) ? 0ll : ((long long)tap * p.kchunks + kc) * p.Npad * 96);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) w[nt][pl] = *reinterpret_cast<const bf16x8*>(src + nt * 32 * 96 + pl * 32);
        };
        load_halo(0);
        ldw(0, 0, wc);
        store_halo(0);
        __syncthreads();
        for (int kc = 0; kc < p.kchunks; ++kc) {
            const bool more_kc = kc + 1 < p.kchunks;
            if (more_kc) load_halo(kc + 1);
            for (int tap = 0; tap < 27; ++tap) {
                const bool last_tap = tap == 26;
                if (!last_tap || more_kc) ldw(last_tap ? 0 : tap + 1, last_tap ? kc + 1 : kc, wnx);
                const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
                const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * PST;
                bf16x8 a[2][3];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        a[mt][pl] = *reinterpret_cast<const bf16x8*>(halo + aoff + mt * (HH * HWD * PST) + pl * 32);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        f32x16 c = acc[mt][nt];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], wc[nt][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][1], wc[nt][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][2], wc[nt][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], wc[nt][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][1], wc[nt][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], wc[nt][0], c, 0, 0, 0);
                        acc[mt][nt] = c;
                    }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wc[nt][pl] = wnx[nt][pl];
            }
            if (more_kc) {
                __syncthreads();
                store_halo(kc + 1);
                __syncthreads();
            }
        }
    } else {
    load_halo(0);
    load_b(0, 0);
    store_halo(0);
    store_b(Bs0);
    __syncthreads();
    int it = 0;
    for (int kc = 0; kc < p.kchunks; ++kc) {
        const bool more_kc = kc + 1 < p.kchunks;
        if (more_kc) load_halo(kc + 1);
        for (int tap = 0; tap < 27; ++tap, ++it) {
            const bool last_tap = tap == 26;
            const bool has_next = !last_tap || more_kc;
            if (has_next) load_b(last_tap ? 0 : tap + 1, last_tap ? kc + 1 : kc);
            const unsigned char* Bs = (it & 1) ? Bs1 : Bs0;
            const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * PST;
            bf16x8 a[2][3], bfr[NT][3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[mt][pl] = *reinterpret_cast<const bf16x8*>(halo + aoff + mt * (HH * HWD * PST) + pl * 32);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bfr[nt][pl] = *reinterpret_cast<const bf16x8*>(Bs + b_lane + nt * 32 * PST + pl * 32);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    f32x16 c = acc[mt][nt];
                    // smallest terms first
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], bfr[nt][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][1], bfr[nt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][2], bfr[nt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], bfr[nt][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][1], bfr[nt][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][0], bfr[nt][0], c, 0, 0, 0);
                    acc[mt][nt] = c;
                }
            if (has_next) store_b((it & 1) ? Bs0 : Bs1);
            __syncthreads();
            if (last_tap && more_kc) {
                store_halo(kc + 1);
                __syncthreads();
            }
        }
    }

    }   // !BDIRECT

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 32 + l31;
        const bool nok = n < p.N;
        const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int f = f0 + wm * 2 + mt;
            if (f >= p.F) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
                int ih, iw;
                lane_hw(i, ih, iw);
                const int h = h0 + ih, w = w0 + iw;
                if (nok && h < p.H && w < p.W && (!(p.dbg & 1) || acc[mt][nt][r] == 1.2345f)) {
                    const float v = acc[mt][nt][r] + bv;
                    p.out[((((long long)b * p.F + f) * p.H + h) * p.W + w) * p.N + n] = v;
                    ssum += v;
                    ssq += v * v;
                }
            }
        }
        if (p.gn_part) {
            // GroupNorm statistics of the OUTPUT: this wave's 2 frames x 32 points of column n (fixed summation order)
            ssum += __shfl_xor(ssum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (hh == 0 && nok) {
                const long long tile = ((long long)(f0 / TF) * nth + h0 / TH) * ntw + w0 / TW;
                float* dst = p.gn_part + ((((long long)b * ((long long)ntf * nth * ntw) + tile) * 2 + wm) * p.N + n) * 2;
                dst[0] = ssum;
                dst[1] = ssq;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent variant: a workgroup walks over work items (tile, n-block) with stride gridDim.x and treats the sequence of
// (item, channel-chunk) pairs as ONE software pipeline: the halo of the next chunk -- including the first chunk of the
// NEXT tile -- is prefetched while the current chunk's 27 taps run, the weight ring keeps streaming across item
// boundaries, and the epilogue stores of a finished tile are issued in front of the next tile's first MFMAs.  This removes
// the per-workgroup prologue / epilogue bubbles (first halo latency, output burst) that cost ~20 us per launch round with
// one tile per workgroup (r01 attribution: 0.36 ms of the 1.19 ms 64->64 @ 64x64 launch did not scale with K).
template <int BN>
__global__ __launch_bounds__(256, 2) void conv3x6p_kernel(Conv3hParams p, long long nitems) {
    using namespace x6;
    constexpr int NT = BN / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int ntf = (p.F + TF - 1) / TF, nth = (p.H + TH - 1) / TH, ntw = (p.W + TW - 1) / TW;
    const int K = p.C0 + p.C1;
    // XCD-contiguous item ranges: workgroup g runs on XCD g % 8; give every XCD one contiguous eighth of the items
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = (nwg + 7 - xcd) / 8;
    const long long per_xcd = (nitems + 7) / 8;
    const long long xbeg = xcd * per_xcd, xend = (xbeg + per_xcd < nitems) ? xbeg + per_xcd : nitems;

    struct Tile { int n0, b, f0, h0, w0; };
    auto decode = [&](long long item) {
        Tile t;
        t.n0 = (int)(item % ntn) * BN;
        long long r = item / ntn;
        t.w0 = (int)(r % ntw) * TW; r /= ntw;
        t.h0 = (int)(r % nth) * TH; r /= nth;
        t.f0 = (int)(r % ntf) * TF;
        t.b = (int)(r / ntf);
        return t;
    };
    // per-thread halo slots (tile independent part)
    int hdst[HLOADS], hpf[HLOADS], hph[HLOADS], hpw[HLOADS];
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
        const int q = tid + 256 * i, pt = q >> 2;
        hpf[i] = pt / (HH * HWL); hph[i] = (pt / HWL) % HH; hpw[i] = pt % HWL;
        hdst[i] = (pt < NLOG) ? ((pt / HWL) * HWD + pt % HWL) * PST + (q & 3) * 8 : -1;
    }
    const int hslot = (tid & 3) * 4;
    f32x4 hreg[HLOADS];
    auto load_halo = [&](const Tile& t, int kc) {
        const int c = kc * KC + hslot;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = p.a0; cs = p.C0; cc = c; }
        else { src = p.a1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            const int f = t.f0 - 1 + hpf[i], h = t.h0 - 1 + hph[i], w = t.w0 - 1 + hpw[i];
            const bool ok = hdst[i] >= 0 && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && ok) v = *reinterpret_cast<const f32x4*>(src + ((((long long)t.b * p.F + f) * p.H + h) * p.W + w) * cs + cc);
            hreg[i] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            if (hdst[i] >= 0) {
                uint2 p1, p2, p3;
                split3(hreg[i], p1, p2, p3);
                *reinterpret_cast<uint2*>(halo + hdst[i]) = p1;
                *reinterpret_cast<uint2*>(halo + hdst[i] + 32) = p2;
                *reinterpret_cast<uint2*>(halo + hdst[i] + 64) = p3;
            }
        }
    };
    int lh, lw;
    lane_hw(l31, lh, lw);
    const int a_lane = (((wm * 2) * HH + lh) * HWD + lw) * PST + hh * 16;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.wp) + ((long long)wn * (BN / 2) + l31) * 96 + hh * 16;
    bf16x8 w[3][NT][3];
    bf16x8 a[2][2][3];
    auto ldw = [&](int n0, int tap, int kc, bf16x8 (&dst)[NT][3]) {
        const unsigned char* src = wbase + (((long long)tap * p.kchunks + kc) * p.Npad + n0) * 96;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[nt][pl] = *reinterpret_cast<const bf16x8*>(src + nt * 32 * 96 + pl * 32);
    };
    auto lda = [&](int tap, bf16x8 (&dst)[2][3]) {
        const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
        const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * PST;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                dst[mt][pl] = *reinterpret_cast<const bf16x8*>(halo + aoff + mt * (HH * HWD * PST) + pl * 32);
    };
    f32x16 acc[2][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };
    auto epilogue = [&](const Tile& t) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = t.n0 + wn * (BN / 2) + nt * 32 + l31;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int f = t.f0 + wm * 2 + mt;
                if (f >= p.F) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    int ih, iw;
                    lane_hw(i, ih, iw);
                    const int h = t.h0 + ih, ww = t.w0 + iw;
                    if (h < p.H && ww < p.W)
                        p.out[((((long long)t.b * p.F + f) * p.H + h) * p.W + ww) * p.N + n] = acc[mt][nt][r] + bv;
                }
            }
        }
    };

    long long item = xbeg + wg_in_xcd;
    if (item >= xend) return;
    Tile cur = decode(item);
    zero_acc();
    load_halo(cur, 0);
    ldw(cur.n0, 0, 0, w[0]);
    ldw(cur.n0, 1, 0, w[1]);
    store_halo();
    __syncthreads();
    lda(0, a[0]);
    while (true) {
        const long long nitem = item + wgs_per_xcd;
        const bool more_items = nitem < xend;
        Tile nxt = cur;
        if (more_items) nxt = decode(nitem);
        for (int kc = 0; kc < p.kchunks; ++kc) {
            const bool last_kc = kc + 1 == p.kchunks;
            const bool more = !last_kc || more_items;
            const int nkc = last_kc ? 0 : kc + 1;
            const int nn0 = last_kc ? nxt.n0 : cur.n0;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int t2 = tap + 2;
                if (t2 < 27) ldw(cur.n0, t2, kc, w[t2 % 3]);
                else if (more) ldw(nn0, t2 - 27, nkc, w[t2 % 3]);
                if (tap == 0 && more) load_halo(last_kc ? nxt : cur, nkc);
                if (tap < 26) lda(tap + 1, a[(tap + 1) & 1]);
                asm volatile("" ::: "memory");      // pin the prefetches HERE (the scheduler otherwise sinks them to their use)
                constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tap & 1][mt][PA[term]], w[tap % 3][nt][PB[term]],
                                                                                  acc[mt][nt], 0, 0, 0);
                asm volatile("" ::: "memory");      // keep the prefetch distance as written: no load hoisting across taps
            }
            if (more) {
                __syncthreads();
                store_halo();
                __syncthreads();
                lda(0, a[0]);
            }
            if (last_kc) {
                epilogue(cur);
                zero_acc();
            }
        }
        if (!more_items) break;
        item = nitem;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// N = 64 layers: 8x4x8 output tile, the four waves stacked along the frame axis, every wave 2 frames x 32 points x all
// 64 output channels (4 accumulators per wave instead of 2: r01 PMC showed the 2x2-wave BN=64 kernel at 56 % MFMA busy
// against 76 % for the 4-accumulator BN=128 kernel).  10x6x10 halo = 600 points x 112 B = 66 KB LDS, 2 workgroups / CU.
namespace x6t {
using namespace x6;
constexpr int TF8 = 8;
constexpr int HF8 = TF8 + 2;
constexpr int NLOG8 = HF8 * HH * HWL;        // 600 halo points
constexpr int NSLOT8 = HF8 * HH * HWD;       // 720 LDS slots
constexpr int HLOADS8 = (NLOG8 * 4 + 255) / 256;
}  // namespace x6t

__global__ __launch_bounds__(256, 2) void conv3x6t_kernel(Conv3hParams p) {
    using namespace x6t;
    constexpr int NT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;                         // [NSLOT8][PST]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / 64;
    const int ntf = (p.F + TF8 - 1) / TF8, nth = (p.H + TH - 1) / TH, ntw = (p.W + TW - 1) / TW;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = (bid % ntn) * 64;
    int t = bid / ntn;
    const int w0 = (t % ntw) * TW; t /= ntw;
    const int h0 = (t % nth) * TH; t /= nth;
    const int f0 = (t % ntf) * TF8;
    const int b = t / ntf;
    const int K = p.C0 + p.C1;

    long long hoff[HLOADS8];
    bool hok[HLOADS8];
    int hdst[HLOADS8];
#pragma unroll
    for (int i = 0; i < HLOADS8; ++i) {
        const int q = tid + 256 * i;
        const int pt = q >> 2;
        const int pf = pt / (HH * HWL), ph = (pt / HWL) % HH, pw = pt % HWL;
        const int f = f0 - 1 + pf, h = h0 - 1 + ph, w = w0 - 1 + pw;
        hok[i] = pt < NLOG8 && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        hoff[i] = (((long long)b * p.F + f) * p.H + h) * p.W + w;
        hdst[i] = ((pt / HWL) * HWD + pt % HWL) * PST + (q & 3) * 8;
    }
    const int hslot = (tid & 3) * 4;
    f32x4 hreg[HLOADS8];
    auto load_halo = [&](int kc) {
        const int c = kc * KC + hslot;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = p.a0; cs = p.C0; cc = c; }
        else { src = p.a1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < HLOADS8; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && hok[i]) v = *reinterpret_cast<const f32x4*>(src + hoff[i] * cs + cc);
            hreg[i] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < HLOADS8; ++i) {
            if (tid + 256 * i < NLOG8 * 4) {
                uint2 p1, p2, p3;
                split3(hreg[i], p1, p2, p3);
                *reinterpret_cast<uint2*>(halo + hdst[i]) = p1;
                *reinterpret_cast<uint2*>(halo + hdst[i] + 32) = p2;
                *reinterpret_cast<uint2*>(halo + hdst[i] + 64) = p3;
            }
        }
    };
    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    int lh, lw;
    lane_hw(l31, lh, lw);
    const int a_lane = (((wm * 2) * HH + lh) * HWD + lw) * PST + hh * 16;
    const unsigned char* wlane = reinterpret_cast<const unsigned char*>(p.wp) + ((long long)n0 + l31) * 96 + hh * 16;
    bf16x8 wc[NT][3], wnx[NT][3];
    auto ldw = [&](int tap, int kc, bf16x8 (&w)[NT][3]) {
        const unsigned char* src = wlane + ((long long)tap * p.kchunks + kc) * p.Npad * 96;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w[nt][pl] = *reinterpret_cast<const bf16x8*>(src + nt * 32 * 96 + pl * 32);
    };
    load_halo(0);
    ldw(0, 0, wc);
    store_halo();
    __syncthreads();
    for (int kc = 0; kc < p.kchunks; ++kc) {
        const bool more_kc = kc + 1 < p.kchunks;
        if (more_kc) load_halo(kc + 1);
        for (int tap = 0; tap < 27; ++tap) {
            const bool last_tap = tap == 26;
            if (!last_tap || more_kc) ldw(last_tap ? 0 : tap + 1, last_tap ? kc + 1 : kc, wnx);
            const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * PST;
            bf16x8 a[2][3];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[mt][pl] = *reinterpret_cast<const bf16x8*>(halo + aoff + mt * (HH * HWD * PST) + pl * 32);
            constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][PA[term]], wc[nt][PB[term]], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wc[nt][pl] = wnx[nt][pl];
        }
        if (more_kc) {
            __syncthreads();
            store_halo();
            __syncthreads();
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 32 + l31;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int f = f0 + wm * 2 + mt;
            if (f >= p.F) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;
                int ih, iw;
                lane_hw(i, ih, iw);
                const int h = h0 + ih, w = w0 + iw;
                if (h < p.H && w < p.W)
                    p.out[((((long long)b * p.F + f) * p.H + h) * p.W + w) * p.N + n] = acc[mt][nt][r] + bv;
            }
        }
    }
}

int launch_conv3x6_impl(const Conv3hParams& p, bool wide, long long tiles, double flops, double bytes, hipStream_t s);

long long conv3x6_tiles_per_sample(int F, int H, int W) {
    using namespace x6;
    return (long long)((F + TF - 1) / TF) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
}

int launch_conv3x6(const Conv3hParams& p, hipStream_t s) {
    using namespace x6;
    DPC_REQUIRE(p.C0 % 4 == 0 && p.C1 % 4 == 0, "conv3x6: channel counts must be multiples of 4");
    DPC_REQUIRE(p.kchunks == (p.C0 + p.C1 + KC - 1) / KC, "conv3x6: kchunks mismatch");
    if (p.B == 0) return DPC_OK;
    const long long tiles = (long long)p.B * ((p.F + TF - 1) / TF) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const double M = (double)p.B * p.F * p.H * p.W;
    const double flops = 2.0 * M * p.N * 27.0 * (p.C0 + p.C1);
    const double bytes = 4.0 * (M * p.N + M * (p.C0 + p.C1) + 27.0 * (p.C0 + p.C1) * p.N);
    const bool wide = p.Npad % 128 == 0 && p.N > 64;
#ifdef DPC_ENABLE_CONV_DBG            // attribution builds only (tools/build_variant.py): these bits make the kernels skip work
    static const int dbg = debug_switch("DPC_CONV_DBG", 0);
#else
    constexpr int dbg = 0;
#endif
    Conv3hParams pd = p;
    pd.dbg = dbg;
    return launch_conv3x6_impl(pd, wide, tiles, flops, bytes, s);
}

int launch_conv3x6_impl(const Conv3hParams& p, bool wide, long long tiles, double flops, double bytes, hipStream_t s) {
    using namespace x6;
    ProfScope prof(wide ? PROF_CONV3X6_128 : PROF_CONV3X6_64, flops, bytes, s);
    static const int bdirect = debug_switch("DPC_CONV3X6_BDIRECT", 3);   // 0 LDS weights, 1 direct, 2 pipelined direct (BN=64 only), 3 pipelined direct for both widths (default), 4 hand-counted vmcnt
    static const int tall = debug_switch("DPC_CONV3X6_TALL", 0);   // measured = v2, kept opt-in
    const bool fused_gn = p.gn_part || p.in_coef;          // only the default kernels implement the GroupNorm fusion
    DPC_REQUIRE(!(p.in_coef && p.C1 != 0), "conv3x6: fused input normalisation needs a single source");
    if (tall && !wide && p.F >= 8 && !fused_gn) {
        using namespace x6t;
        DPC_REQUIRE(p.Npad % 64 == 0, "conv3x6: Npad must be a multiple of 64");
        const long long tiles8 = (long long)p.B * ((p.F + TF8 - 1) / TF8) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
        const long long grid = tiles8 * (p.Npad / 64);
        DPC_REQUIRE(grid < (1ll << 31), "conv3x6: grid too large");
        const size_t lds = (size_t)NSLOT8 * PST;
        static DeviceOnce once;
        if (!once) { DPC_HIP(hipFuncSetAttribute((const void*)conv3x6t_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
        hipLaunchKernelGGL(conv3x6t_kernel, dim3((unsigned)grid), dim3(256), lds, s, p);
        DPC_LAUNCH_CHECK();
        return DPC_OK;
    }
    static const int persistent = debug_switch("DPC_CONV3X6_PERSISTENT", 0);
    if (persistent && !wide && p.kchunks >= 1 && !fused_gn) {
        DPC_REQUIRE(p.Npad % 64 == 0, "conv3x6: Npad must be a multiple of 64");
        const long long nitems = tiles * (p.Npad / 64);
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            hipDeviceProp_t prop;
            DPC_HIP(hipGetDevice(&dev));
            DPC_HIP(hipGetDeviceProperties(&prop, dev));
            ncu = prop.multiProcessorCount;
        }
        const long long grid = std::min<long long>(nitems, (long long)ncu * 2);
        hipLaunchKernelGGL((conv3x6p_kernel<64>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p, nitems);
        DPC_LAUNCH_CHECK();
        return DPC_OK;
    }
    if (wide) {
        const long long grid = tiles * (p.Npad / 128);
        DPC_REQUIRE(grid < (1ll << 31), "conv3x6: grid too large");
        if (bdirect == 4) {
            hipLaunchKernelGGL((conv3x6_kernel<128, 4>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else if (bdirect == 3) {      // v2 at BN = 128 needs > 256 VGPRs (spills): opt-in only
            hipLaunchKernelGGL((conv3x6_kernel<128, 2>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else if (bdirect) {
            hipLaunchKernelGGL((conv3x6_kernel<128, 1>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else {
            const size_t lds = (size_t)NSLOT * PST + 2 * 128 * PST;
            static DeviceOnce once;
            if (!once) { DPC_HIP(hipFuncSetAttribute((const void*)conv3x6_kernel<128, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
            hipLaunchKernelGGL((conv3x6_kernel<128, 0>), dim3((unsigned)grid), dim3(256), lds, s, p);
        }
    } else {
        DPC_REQUIRE(p.Npad % 64 == 0, "conv3x6: Npad must be a multiple of 64");
        const long long grid = tiles * (p.Npad / 64);
        DPC_REQUIRE(grid < (1ll << 31), "conv3x6: grid too large");
        if (bdirect == 4) {
            hipLaunchKernelGGL((conv3x6_kernel<64, 4>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else if (bdirect >= 2) {
            hipLaunchKernelGGL((conv3x6_kernel<64, 2>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else if (bdirect) {
            hipLaunchKernelGGL((conv3x6_kernel<64, 1>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else {
            const size_t lds = (size_t)NSLOT * PST + 2 * 64 * PST;
            hipLaunchKernelGGL((conv3x6_kernel<64, 0>), dim3((unsigned)grid), dim3(256), lds, s, p);
        }
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ---- weight pre-split: reference [N][K][3][3][3] fp32 -> [27][kchunks][Npad][3 planes][16] bf16
__global__ void pack_weights_x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int N, int Npad,
                                       int K, int kchunks) {
    const long long total = 27ll * kchunks * Npad * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % 16);
        long long r = i / 16;
        const int n = (int)(r % Npad);
        r /= Npad;
        const int kc = (int)(r % kchunks);
        const int tap = (int)(r / kchunks);
        const int c = kc * 16 + kk;
        float v = 0.f;
        if (n < N && c < K) v = w[((long long)n * K + c) * 27 + tap];
        const unsigned p1 = x6::cvt_pk_bf16(v, 0.f) & 0xffffu;
        const float r1 = v - __uint_as_float(p1 << 16);
        const unsigned p2 = x6::cvt_pk_bf16(r1, 0.f) & 0xffffu;
        const float r2 = r1 - __uint_as_float(p2 << 16);
        const unsigned p3 = x6::cvt_pk_bf16(r2, 0.f) & 0xffffu;
        unsigned short* dst = wp + (((long long)tap * kchunks + kc) * Npad + n) * 48 + kk;
        dst[0] = (unsigned short)p1;
        dst[16] = (unsigned short)p2;
        dst[32] = (unsigned short)p3;
    }
}

int launch_pack_weights_x6(const float* w, void* wp, int N, int Npad, int K, hipStream_t s) {
    const int kchunks = (K + 15) / 16;
    const long long total = 27ll * kchunks * Npad * 16;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_x6_kernel, dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), N,
                       Npad, K, kchunks);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
