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

template <int BN, bool BDIRECT>
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
            if (cok && hok[i]) v = *reinterpret_cast<const f32x4*>(src + hoff[i] * cs + cc);
            hreg[i] = v;
        }
    };
    auto store_halo = [&]() {
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

    if constexpr (BDIRECT) {
        // weight fragments straight from L2/L1 into registers, one tap ahead; no barrier inside the tap loop
        const unsigned char* wlane = reinterpret_cast<const unsigned char*>(p.wp) +
                                     ((long long)n0 + wn * (BN / 2) + l31) * 96 + hh * 16;
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
                store_halo();
                __syncthreads();
            }
        }
    } else {
    load_halo(0);
    load_b(0, 0);
    store_halo();
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
                store_halo();
                __syncthreads();
            }
        }
    }

    }   // !BDIRECT

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 32 + l31;
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

int conv_mode_default() {
    static const int mode = [] {
        const char* e = getenv("DPC_CONV_MODE");
        if (e && (e[0] == 'f' || e[0] == 'F')) return 0;
        return 1;
    }();
    return mode;
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
    ProfScope prof(wide ? PROF_CONV3X6_128 : PROF_CONV3X6_64, flops, bytes, s);
    static const int bdirect = [] { const char* e = getenv("DPC_CONV3X6_BDIRECT"); return e ? atoi(e) : 1; }();   // default: direct (measured faster: no per-tap barrier)
    if (wide) {
        const long long grid = tiles * (p.Npad / 128);
        DPC_REQUIRE(grid < (1ll << 31), "conv3x6: grid too large");
        if (bdirect) {
            hipLaunchKernelGGL((conv3x6_kernel<128, true>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else {
            const size_t lds = (size_t)NSLOT * PST + 2 * 128 * PST;
            static bool once = false;
            if (!once) { DPC_HIP(hipFuncSetAttribute((const void*)conv3x6_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
            hipLaunchKernelGGL((conv3x6_kernel<128, false>), dim3((unsigned)grid), dim3(256), lds, s, p);
        }
    } else {
        DPC_REQUIRE(p.Npad % 64 == 0, "conv3x6: Npad must be a multiple of 64");
        const long long grid = tiles * (p.Npad / 64);
        DPC_REQUIRE(grid < (1ll << 31), "conv3x6: grid too large");
        if (bdirect) {
            hipLaunchKernelGGL((conv3x6_kernel<64, true>), dim3((unsigned)grid), dim3(256), (size_t)NSLOT * PST, s, p);
        } else {
            const size_t lds = (size_t)NSLOT * PST + 2 * 64 * PST;
            hipLaunchKernelGGL((conv3x6_kernel<64, false>), dim3((unsigned)grid), dim3(256), lds, s, p);
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
