// Conv3d 3x3x3 / stride 1 / pad 1, channels-last, on the fp16 matrix cores with split operands ("f16x3").
//
// Every fp32 operand x is pre-scaled by a power of two (activations 2^4, weights 2^12: keeps both split terms inside
// fp16's narrow exponent range) and written as  x = h1 + h2 + e  with h1 = fp16(x), h2 = fp16(x - h1) (round to nearest):
// 22 significant bits, |e| <= 2^-22 |x|.  A product is the three partial products of weight >= 2^-11
//        a1 b1 + (a1 b2 + a2 b1)
// each exact in fp32 (11 x 11-bit significands) and accumulated in fp32 by v_mfma_f32_32x32x16_f16; the dropped a2 b2 term
// and the split remainders are <= 3 * 2^-22 relative per product.  On a K = 27 x 64 convolution this is BELOW the fp32
// accumulation error of the sum itself: against fp64 the result is as accurate as the 6-product bf16 scheme of
// conv3x6.hip (tools/f16x3_error.py: rms 1.2e-6 vs 1.4e-6 of the sum's magnitude, plain fp32 0.75e-6) at half the
// matrix-core work, and far inside the 1e-4 forward tolerance of SURVEY 8(d).  (The reference itself runs these
// convolutions in TF32 -- 10 significant bits -- on the GPUs it was written for: torch.backends.cudnn.allow_tf32.)
// DPC_CONV_MODE=x6 selects the bf16x6 kernels, DPC_CONV_MODE=f32 the native fp32 MFMA.
//
// Data flow = conv3x6.hip's pipelined direct-weight variant: a workgroup owns a 4x4x8 output tile; per 16-channel chunk
// the 6x6x10 halo is staged once in LDS already split into the two fp16 planes (80 B per point: 2 x 32 B + 16 B pad =>
// conflict-free ds_read_b128 with the pitch-12 / lane_hw layout), the 27 taps are LDS offsets, and the pre-split weights
// ([tap][chunk][n][2 planes][16] fp16) stream from L2 into a 3-deep register ring two taps ahead.  The epilogue rescales
// by 2^-16, adds the bias and emits the GroupNorm partial sums; the halo staging optionally applies the producer's
// GroupNorm + (scale, shift) + SiLU (see Conv3hParams).
// Reference op: nn.Conv3d(dim, dim_out, (3,3,3), padding=(1,1,1)) in Block (video_diffusion_pytorch_conv3d.py:192).
#include "common.h"

namespace dpc {

namespace f3 {
constexpr int TH = 4, TW = 8;
constexpr int HH = TH + 2, HWL = TW + 2, HWD = 12;
constexpr int KC = 16;
constexpr int PST = 80;                    // bytes per halo point in LDS (2 planes x 32 B + 16 pad; 5 x 16 B: odd)
constexpr int WROW = 64;                   // bytes per output channel per (tap, chunk) in the packed weights
constexpr float SA = 16.0f, SW = 4096.0f;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lane_hw(int i, int& h, int& w) {      // see conv3h.hip
    if (i < 4) { h = 0; w = i; }
    else if (i < 12) { h = 1; w = i - 4; }
    else if (i < 16) { h = 0; w = i - 8; }
    else if (i < 20) { h = 3; w = i - 16; }
    else if (i < 28) { h = 2; w = i - 20; }
    else { h = 3; w = i - 24; }
}

__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sat16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f); }

// 4 floats (already scaled) -> two (2 x u32) packs of 4 fp16 each: h1 = fp16(x), h2 = fp16(x - h1)
__device__ __forceinline__ void split2(const f32x4 v, uint2& p1, uint2& p2) {
    const float x0 = sat16x(v.x), x1 = sat16x(v.y), x2 = sat16x(v.z), x3 = sat16x(v.w);
    p1.x = cvt_pk_f16(x0, x1);
    p1.y = cvt_pk_f16(x2, x3);
    p2.x = f16_sub_pk(x0, x1, p1.x);
    p2.y = f16_sub_pk(x2, x3, p1.y);
}
}  // namespace f3

// WM x WN waves (WM * WN = 4): wave (wm, wn) owns frames 2 wm, 2 wm + 1 of the tile (2 x 32 points) and BN / WN channels.
// <64, 2> / <128, 2>: 4x4x8 tile, 2 x 2 waves.  <64, 4>: 8x4x8 tile, 4 x 1 waves -- every A fragment feeds both 32-channel
// column tiles (half the LDS reads per MFMA of <64, 2>) and the prologue / epilogue are amortised over twice the work.
// WR: weight ring depth (divides 27), AD: A-fragment ring depth (2 or 3).
template <int BN, int WM, int WR, int AD>
__global__ __launch_bounds__(256, 2) void conv3f3_kernel(Conv3hParams p) {
    fp16_ovfl_enable();                                 // (common.h: operand conversions saturate in hardware)
    static_assert(27 % WR == 0 && (AD == 2 || AD == 3), "static ring indices across channel chunks");
    using namespace f3;
    constexpr int WN = 4 / WM;
    constexpr int NT = BN / (32 * WN);
    constexpr int TF = 2 * WM, HF = TF + 2;
    constexpr int NLOG = HF * HH * HWL;        // 360 / 600 halo points
    constexpr int HLOADS = (NLOG * 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f3[];
    unsigned char* halo = smem_f3;                      // [NSLOT][PST]

    const float sa = p.act_scale != 0.f ? p.act_scale : SA;            // (uniform: scalar registers)
    const float descale = 1.0f / (sa * SW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int ntf = (p.F + TF - 1) / TF, nth = (p.H + TH - 1) / TH, ntw = (p.W + TW - 1) / TW;
    int bid = blockIdx.x;
    {   // XCD-aware order: consecutive tiles (shared halo planes, same weights) land on the same XCD's L2
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
            if (cok && hok[i] && !(p.dbg & 2)) v = *reinterpret_cast<const f32x4*>(src + hoff[i] * cs + cc);
            hreg[i] = v;
        }
    };
    auto store_halo = [&](int kc) {
        if (p.in_coef) {
            // fused GroupNorm -> (scale + 1, shift) -> SiLU of the producer (Block.forward, ...conv3d.py:196-204); the zero
            // padding of the convolution applies to the ACTIVATED tensor, so out-of-range points stay 0
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
                uint2 p1, p2;
                split2(hreg[i] * sa, p1, p2);
                *reinterpret_cast<uint2*>(halo + hdst[i]) = p1;
                *reinterpret_cast<uint2*>(halo + hdst[i] + 32) = p2;
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

    // weight fragments straight from L2/L1 into a WR-deep register ring (WR-1 taps ahead: at 3 MFMAs per product a tap
    // lasts only 192 (BN = 64) / 384 cycles, so the 64-wide kernel runs 8 taps ahead to cover the L2 latency; 27 % WR == 0
    // keeps the ring index static across channel chunks), A fragments AD-1 taps ahead, the 27 taps fully unrolled (tap
    // offsets are ds_read immediates), the 3 split-product MFMAs of the 2*NT accumulators issued round-robin.
    const unsigned char* wlane = reinterpret_cast<const unsigned char*>(p.wp) +
                                 ((long long)n0 + wn * (BN / WN) + l31) * WROW + hh * 16;
    f16x8 w[WR][NT][2];
    f16x8 a[AD][2][2];
    auto ldw = [&](int tap, int kc, f16x8 (&dst)[NT][2]) {
        if ((p.dbg & 16) && (tap | kc)) return;                  // perf attribution: no weight traffic after the first tap
        const unsigned char* src = wlane + ((p.dbg & 4) ? 0ll : ((long long)tap * p.kchunks + kc) * p.Npad * WROW);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) dst[nt][pl] = *reinterpret_cast<const f16x8*>(src + nt * 32 * WROW + pl * 32);
    };
    auto lda = [&](int tap, f16x8 (&dst)[2][2]) {
        if ((p.dbg & 32) && tap) return;                         // perf attribution: no A-fragment LDS reads after tap 0
        const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
        const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * PST;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                dst[mt][pl] = *reinterpret_cast<const f16x8*>(halo + aoff + mt * (HH * HWD * PST) + pl * 32);
    };
    load_halo(0);
#pragma unroll
    for (int i = 0; i < WR - 1; ++i) ldw(i, 0, w[i]);
    store_halo(0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AD - 1; ++i) lda(i, a[i]);
    for (int kc = 0; kc < p.kchunks; ++kc) {
        const bool more_kc = kc + 1 < p.kchunks;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int tw = tap + WR - 1, ta = tap + AD - 1;
            if (tw < 27) ldw(tw, kc, w[tw % WR]);
            else if (more_kc) ldw(tw - 27, kc + 1, w[tw % WR]);
            if (tap == 0 && more_kc) load_halo(kc + 1);
            if (ta < 27) lda(ta, a[ta % AD]);
            asm volatile("" ::: "memory");      // pin the prefetches HERE (the scheduler otherwise sinks them to their use)
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};     // small terms first
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tap % AD][mt][PA[term]], w[tap % WR][nt][PB[term]],
                                                                             acc[mt][nt], 0, 0, 0);
            asm volatile("" ::: "memory");
        }
        if (more_kc) {
            __syncthreads();
            store_halo(kc + 1);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < AD - 1; ++i) lda(i, a[i]);
        }
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / WN) + nt * 32 + l31;
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
                const int h = h0 + ih, w_ = w0 + iw;
                if (nok && h < p.H && w_ < p.W && (!(p.dbg & 1) || acc[mt][nt][r] == 1.2345f)) {
                    const float v = acc[mt][nt][r] * descale + bv;
                    p.out[((((long long)b * p.F + f) * p.H + h) * p.W + w_) * p.N + n] = v;
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
                float* dst = p.gn_part + ((((long long)b * ((long long)ntf * nth * ntw) + tile) * WM + wm) * p.N + n) * 2;
                dst[0] = ssum;
                dst[1] = ssq;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Big-tile variant (default where the shape allows): one 256-thread workgroup per CU, 512 registers per lane.
// r01 attribution of the kernel above (DPC_CONV_DBG bits 16 / 32): streaming the weight fragments through the 64 B/clk
// vector L1 costs 28 % of the launch at every shape -- at 3 MFMAs per product a wave retires a (2 x NT)-tile tap in
// 192 NT cycles, so L1 runs at 2/3 of its bandwidth -- while the LDS A-fragment reads cost 4 %.  Here every wave owns a
// 4 x 2 grid of 32x32 accumulators (4 point slabs x 64 channels, 128 accumulator registers): a weight fragment feeds
// 4 slabs and an A fragment 2 column tiles, so per MFMA the L1 and the LDS traffic are both HALF of the (2 x 2) kernel's.
//   BN = 64 : 4 x 1 waves over an 8 x 8 x 8 output tile  (halo 10 x 10 x 10, 96 KB)
//   BN = 128: 2 x 2 waves over a  4 x 8 x 8 output tile  (halo  6 x 10 x 10, 57.6 KB)
// A wave's slab mt is frame 2 wm + (mt >> 1), rows 4 (mt & 1) .. +3 of the tile.  With one workgroup per CU nothing else
// hides the channel-chunk hand-over, so the GroupNorm/SiLU/split arithmetic of the NEXT chunk's halo runs in the middle
// of the current chunk's taps (VALU co-issues with the MFMAs) and the hand-over is barrier, 16-byte LDS stores, barrier.
namespace f3b {
using namespace f3;
constexpr int TH8 = 8, HH8 = TH8 + 2;
}

// KD = 3: the 3x3x3 convolution.  KD = 1: a (1,3,3) convolution -- the 2-D U-Net's 3x3 convs with the batch on the frame
// axis (no coupling between frames: 9 taps, no frame halo).  Partial frame tiles are allowed (F need not divide), so the
// kernel choice never depends on the batch size.
template <int BN, int KD, bool PERSIST = false>
__global__ __launch_bounds__(256, 1) void conv3f3b_kernel(Conv3hParams p) {
    fp16_ovfl_enable();                                 // (common.h: operand conversions saturate in hardware)
    using namespace f3b;
    constexpr int WM = BN == 64 ? 4 : 2, WN = 4 / WM, MT = 4, NT = 2;
    constexpr int TF = 2 * WM, HF = TF + KD - 1, NTAPS = 9 * KD, FPAD = KD / 2;
    constexpr int NLOG = HF * HH8 * HWL;               // 1000 / 600 halo points
    constexpr int HLOADS = (NLOG * 4 + 255) / 256;     // 16 / 10
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f3b[];
    unsigned char* halo = smem_f3b;                    // [HF*HH8*HWD slots][PST]

    const float sa = p.act_scale != 0.f ? p.act_scale : SA;            // (uniform: scalar registers)
    const float descale = 1.0f / (sa * SW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int ntf = (p.F + TF - 1) / TF, nth = (p.H + TH8 - 1) / TH8, ntw = (p.W + TW - 1) / TW;
    // Persistent form (p.total_wg > gridDim.x): the grid is one workgroup per CU (a multiple of 8) and every workgroup walks
    // its XCD's share of the tiles, so a CU never waits for a workgroup to drain and the next one to be dispatched.
    const int nb = PERSIST ? p.total_wg : (int)gridDim.x;
    int wg = blockIdx.x;
    do {
    if (PERSIST && wg != (int)blockIdx.x) __syncthreads();          // the previous tile's last taps still read the halo
    int bid = wg;
    {   // XCD-aware order: consecutive tiles (shared halo planes, same weights) land on the same XCD's L2
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = (bid % ntn) * BN;
    int t = bid / ntn;
    const int w0 = (t % ntw) * TW; t /= ntw;
    const int h0 = (t % nth) * TH8; t /= nth;
    const int f0 = (t % ntf) * TF;
    const int b = t / ntf;
    const int K = p.C0 + p.C1;
#ifdef DPC_CONV_STAMPS
    unsigned long long tstamp[16];
    int nstamp = 0;
    auto stamp = [&]() { if (nstamp < 16) tstamp[nstamp++] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [&]() {};
#endif
    stamp();

    // halo bookkeeping: a validity bit mask, the point index relative to the sample (32-bit) and the LDS slot per quad.
    // (Recomputing the two index arrays at every chunk cost 1-2 k cycles of MFMA-idle address arithmetic per chunk.)
    unsigned hokm = 0;
    int hpt[HLOADS], hdst[HLOADS];
    const float* xb0 = p.a0 + (long long)b * p.F * p.H * p.W * p.C0;
    const float* xb1 = p.a1 ? p.a1 + (long long)b * p.F * p.H * p.W * p.C1 : nullptr;
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
        const int pt = (tid + 256 * i) >> 2;
        const int pf = pt / (HH8 * HWL), ph = (pt / HWL) % HH8, pw = pt % HWL;
        const int f = f0 - FPAD + pf, h = h0 - 1 + ph, w = w0 - 1 + pw;
        if (pt < NLOG && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) hokm |= 1u << i;
        hpt[i] = (f * p.H + h) * p.W + w;
        hdst[i] = ((pt / HWL) * HWD + pt % HWL) * PST + ((tid + 256 * i) & 3) * 8;          // + plane*32
    }
    const int hslot = (tid & 3) * 4;

    f32x4 hreg[HLOADS];
    uint4 hpk[HLOADS];
    auto load_halo = [&](int kc) {
        const int c = kc * KC + hslot;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = xb0; cs = p.C0; cc = c; }
        else { src = xb1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && ((hokm >> i) & 1)) v = *reinterpret_cast<const f32x4*>(src + (long long)hpt[i] * cs + cc);
            hreg[i] = v;
        }
    };
    // producer's GroupNorm -> (scale + 1, shift) -> SiLU (Block.forward, ...conv3d.py:196-204) when fused, then pre-scale and
    // split; the zero padding of the convolution applies to the ACTIVATED tensor, so out-of-range points stay 0
    auto prepare_halo = [&](int kc) {
        if (p.in_coef) {
            const int c = kc * KC + hslot;
            if (c < K) {
                const f32x4* cf = reinterpret_cast<const f32x4*>(p.in_coef) + ((long long)b * (K >> 2) + (c >> 2)) * 5;
                const f32x4 mu = cf[0], ga = cf[1], be = cf[2], sc = cf[3], sh = cf[4];
#pragma unroll
                for (int i = 0; i < HLOADS; ++i) {
                    if ((hokm >> i) & 1) {
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
            uint2 p1, p2;
            split2(hreg[i] * sa, p1, p2);
            hpk[i] = uint4{p1.x, p1.y, p2.x, p2.y};
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            if (tid + 256 * i < NLOG * 4) {
                const int d = hdst[i];
                *reinterpret_cast<uint2*>(halo + d) = uint2{hpk[i].x, hpk[i].y};
                *reinterpret_cast<uint2*>(halo + d + 32) = uint2{hpk[i].z, hpk[i].w};
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    int lh, lw;
    lane_hw(l31, lh, lw);
    const int a_lane = (((wm * 2) * HH8 + lh) * HWD + lw) * PST + hh * 16;
    const unsigned char* wlane = reinterpret_cast<const unsigned char*>(p.wp) +
                                 ((long long)n0 + wn * (BN / WN) + l31) * WROW + hh * 16;
    f16x8 w[3][NT][2];
    f16x8 a[2][MT][2];
    // the weight stream is walked with ONE running pointer (order: taps 0..26 of chunk 0, taps 0..26 of chunk 1, ...): with the
    // taps unrolled, per-tap base addresses would otherwise be hoisted into ~54 registers
    const long long wstride = (long long)p.Npad * WROW, wtap = wstride * p.kchunks;
    const unsigned char* wnext = wlane;
    int wtap_i = 0, wkc_i = 0;
    auto ldw = [&](int, int, f16x8 (&dst)[NT][2]) {
        const unsigned char* src = wnext;
        if (++wtap_i == NTAPS) { wtap_i = 0; ++wkc_i; wnext = wlane + wkc_i * wstride; }
        else wnext += wtap;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) dst[nt][pl] = *reinterpret_cast<const f16x8*>(src + nt * 32 * WROW + pl * 32);
    };
    auto lda = [&](int tap, f16x8 (&dst)[MT][2]) {
        const int df = KD == 3 ? tap / 9 : 0, dh = (tap / 3) % 3, dw = tap % 3;
        const int aoff = a_lane + ((df * HH8 + dh) * HWD + dw) * PST;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                dst[mt][pl] = *reinterpret_cast<const f16x8*>(halo + aoff + (((mt >> 1) * HH8 + 4 * (mt & 1)) * HWD) * PST + pl * 32);
    };
    load_halo(0);
    ldw(0, 0, w[0]);
    ldw(1, 0, w[1]);
    prepare_halo(0);
    store_halo();
    __syncthreads();
    lda(0, a[0]);
    stamp();                                   // 1: prologue
    for (int kc = 0; kc < p.kchunks; ++kc) {
        const bool more_kc = kc + 1 < p.kchunks;
        auto tap_body = [&](int tap) {
            const int tw = tap + 2;
            if (tw < NTAPS) ldw(tw, kc, w[tw % 3]);
            else if (more_kc) ldw(tw - NTAPS, kc + 1, w[tw % 3]);
            if (tap < NTAPS - 1) lda(tap + 1, a[(tap + 1) & 1]);
            // pin the prefetches HERE: without the scheduling barrier the compiler hoists this tap's MFMAs above them (they are
            // not memory operations), which leaves every load right in front of its first use -- no prefetch distance at all
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};     // small terms first
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tap & 1][mt][PA[term]], w[tap % 3][nt][PB[term]],
                                                                             acc[mt][nt], 0, 0, 0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // (the halo code stays OUTSIDE the unrolled tap loops: inside, the unroller gives up and the rings go to scratch)
        // the next chunk's halo loads go BEHIND the tap-0 weight prefetch: vmcnt retires in order, so every weight load issued
        // after them also waits for them -- this way the first such load is needed three taps (2300 cycles) later
        tap_body(0);
        if (kc < 2) stamp();                   // tap 0
        if (more_kc) load_halo(kc + 1);
        asm volatile("" ::: "memory");
        if (kc < 2) stamp();                   // halo issue
#pragma unroll
        for (int tap = 1; tap < NTAPS / 2; ++tap) tap_body(tap);
        if (kc < 2) stamp();                   // taps 1..11
        if (more_kc) prepare_halo(kc + 1);
        if (kc < 2) stamp();                   // prepare
#pragma unroll
        for (int tap = NTAPS / 2; tap < NTAPS; ++tap) tap_body(tap);
        if (kc < 2) stamp();                   // taps 12..26
        if (more_kc) {
            __syncthreads();
            store_halo();
            __syncthreads();
            lda(0, a[0]);
            if (kc < 2) stamp();               // hand-over
        }
    }

    // ---- epilogue: the launcher guarantees full tiles in H and W (H % 8 == 0, W % 8 == 0; frames are checked per slab), so a slab's 32 points are a
    //      fixed per-lane offset pattern from one base pointer (no bounds checks, no 64-bit address arithmetic per store)
    int poff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int ih, iw;
        lane_hw((r & 3) + 8 * (r >> 2) + 4 * hh, ih, iw);
        poff[r] = (ih * p.W + iw) * p.N;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / WN) + nt * 32 + l31;
        const bool nok = n < p.N;
        const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
        float ssum = 0.f, ssq = 0.f;
        if (nok) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int f = f0 + wm * 2 + (mt >> 1), h = h0 + 4 * (mt & 1);
                if (f >= p.F) continue;              // partial frame tile
                float* base = p.out + ((((long long)b * p.F + f) * p.H + h) * p.W + w0) * p.N + n;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mt][nt][r] * descale + bv;
                    base[poff[r]] = v;
                    ssum += v;
                    ssq += v * v;
                }
            }
        }
        if (p.gn_part) {
            // GroupNorm statistics of the OUTPUT: this wave's 4 slabs x 32 points of column n (fixed summation order)
            ssum += __shfl_xor(ssum, 32, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (hh == 0 && nok) {
                const long long tile = ((long long)(f0 / TF) * nth + h0 / TH8) * ntw + w0 / TW;
                float* dst = p.gn_part + ((((long long)b * ((long long)ntf * nth * ntw) + tile) * WM + wm) * p.N + n) * 2;
                dst[0] = ssum;
                dst[1] = ssq;
            }
        }
    }
#ifdef DPC_CONV_STAMPS
    {   // per-wave records at a stride of 32 floats: [1..15] stage deltas, [16..21] raw 32-bit words: start, end (s_memtime lo/hi),
        // HW_ID, XCC_ID -- tools/conv_stamps.py rebuilds the per-CU timeline (gaps between consecutive workgroups) from them
        const unsigned long long tend = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            float* rec = p.out + ((long long)wg * 4 + wave) * 32;
            for (int i = 1; i < nstamp; ++i) rec[i] = (float)(tstamp[i] - tstamp[i - 1]);
            rec[15] = (float)(tend - tstamp[nstamp - 1]);         // epilogue
            unsigned* ru = reinterpret_cast<unsigned*>(rec);
            ru[16] = (unsigned)tstamp[0]; ru[17] = (unsigned)(tstamp[0] >> 32);
            ru[18] = (unsigned)tend; ru[19] = (unsigned)(tend >> 32);
            ru[20] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
            ru[21] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
        }
    }
#endif
    } while (PERSIST && (wg += gridDim.x) < nb);   // persistent tile loop
}

// which tiling a launch uses: 0 = (2 x 2)-accumulator kernel on 4x4x8 tiles, 1 = 8x4x8 tiles (64-wide), 2 = big-tile kernel
static int conv3f3_variant(int F, int H, int W, int N, int Npad) {
    static const int big_ok = debug_switch("DPC_CONV3F3_BIG", 1);
    static const int tall_ok = debug_switch("DPC_CONV3F3_TALL", 1);
    const bool wide = Npad % 128 == 0 && N > 64;
    if (big_ok && H % 8 == 0 && W % 8 == 0 && (F % (wide ? 4 : 8) == 0 || F >= 16)) return 2;       // partial frame tiles: F >= 16 only
    if (!wide && tall_ok && F % 8 == 0) return 1;
    return 0;
}

// number of GroupNorm partial-sum entries per sample and channel that the conv epilogue writes ([B][entries][N][2])
long long conv3f3_gn_entries(int F, int H, int W, int N, int Npad) {
    using namespace f3;
    const bool wide = Npad % 128 == 0 && N > 64;
    if (conv3w_shape_ok(F, H, W, N, Npad)) return conv3w_gn_entries(F, H, W);
    const int v = conv3f3_variant(F, H, W, N, Npad);
    if (v == 2) {
        const int tf = wide ? 4 : 8;
        return (long long)((F + tf - 1) / tf) * (H / 8) * (W / 8) * (wide ? 2 : 4);
    }
    const int tf = v == 1 ? 8 : 4;
    return (long long)((F + tf - 1) / tf) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW) * (tf / 2);
}

int launch_conv3f3(const Conv3hParams& p, hipStream_t s) {
    using namespace f3;
    DPC_REQUIRE(p.C0 % 4 == 0 && p.C1 % 4 == 0, "conv3f3: channel counts must be multiples of 4");
    DPC_REQUIRE(p.kchunks == (p.C0 + p.C1 + KC - 1) / KC, "conv3f3: kchunks mismatch");
    DPC_REQUIRE(!(p.in_coef && p.C1 != 0), "conv3f3: fused input normalisation needs a single source");
    if (p.B == 0) return DPC_OK;
    const double M = (double)p.B * p.F * p.H * p.W;
    const double ntap = p.kd == 1 ? 9.0 : 27.0;
    const double flops = 2.0 * M * p.N * ntap * (p.C0 + p.C1);
    const double bytes = 4.0 * (M * p.N + M * (p.C0 + p.C1) + ntap * (p.C0 + p.C1) * p.N);
    const bool wide = p.Npad % 128 == 0 && p.N > 64;
#ifdef DPC_ENABLE_CONV_DBG            // attribution builds only (tools/build_variant.py): these bits make the kernels skip work
    static const int dbg = debug_switch("DPC_CONV_DBG", 0);
#else
    constexpr int dbg = 0;
#endif
    Conv3hParams pd = p;
    pd.dbg = dbg;
    ProfScope prof(wide ? PROF_CONV3X6_128 : PROF_CONV3X6_64, flops, bytes, s);
    const bool flat = p.kd == 1;                 // (1,3,3) convolution: big-tile kernel only
    DPC_REQUIRE(!flat || (p.H % 8 == 0 && p.W % 8 == 0), "conv3f3: the (1,3,3) form needs H % 8 == 0 and W % 8 == 0");
    if (conv3w_supported(pd)) return launch_conv3w(pd, s);        // Winograd F(2,3) over frames (conv3w.hip)
    DPC_REQUIRE(p.wp, "conv3f3: the direct kernels need the plain f16x3 weight pack (only the Winograd pack was supplied, and this "
                      "shape / operand scale does not take the Winograd kernel)");
    DPC_REQUIRE(flat || !p.gn_part || !conv3w_shape_ok(p.F, p.H, p.W, p.N, p.Npad),
                "conv3f3: GroupNorm partial sums are laid out for the Winograd kernel but its weight pack is missing");
    const int variant = flat ? 2 : conv3f3_variant(p.F, p.H, p.W, p.N, p.Npad);
    static const int flat_c = debug_switch("DPC_CONV2D_LOADER_WAVES", 1);
    if (variant == 2 && (!flat || flat_c) && conv3f3c_supported(pd)) return launch_conv3f3c(pd, s);     // loader-wave / persistent form
    DPC_REQUIRE(!flat || !(p.gn_part || p.in_coef), "conv3f3: the per-image GroupNorm hooks of the (1,3,3) form exist in conv3f3c only");
    if (variant == 2) {
        const int tf = wide ? 4 : 8;
        const long long tiles = (long long)p.B * ((p.F + tf - 1) / tf) * (p.H / 8) * (p.W / 8);
        const long long grid = tiles * (p.Npad / (wide ? 128 : 64));
        DPC_REQUIRE(grid < (1ll << 31), "conv3f3: grid too large");
        const size_t lds = (size_t)(tf + (flat ? 0 : 2)) * f3b::HH8 * HWD * PST;
        static DeviceOnce once;
        if (!once) {
            DPC_HIP(hipFuncSetAttribute((const void*)conv3f3b_kernel<64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * 10 * 12 * PST));
            DPC_HIP(hipFuncSetAttribute((const void*)conv3f3b_kernel<128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 10 * 12 * PST));
            DPC_HIP(hipFuncSetAttribute((const void*)conv3f3b_kernel<64, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * 10 * 12 * PST));
            DPC_HIP(hipFuncSetAttribute((const void*)conv3f3b_kernel<128, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 10 * 12 * PST));
            DPC_HIP(hipFuncSetAttribute((const void*)conv3f3b_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 10 * 12 * PST));
            DPC_HIP(hipFuncSetAttribute((const void*)conv3f3b_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 10 * 12 * PST));
            once = true;
        }
        static const int persist = debug_switch("DPC_CONV3F3_PERSIST", 0);
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            hipDeviceProp_t prop;
            DPC_HIP(hipGetDevice(&dev));
            DPC_HIP(hipGetDeviceProperties(&prop, dev));
            ncu = prop.multiProcessorCount / 8 * 8;
        }
        long long grid_l = grid;
        pd.total_wg = 0;
        if (persist && !flat && grid > ncu) { pd.total_wg = (int)grid; grid_l = ncu; }
        if (flat) {
            if (wide) hipLaunchKernelGGL((conv3f3b_kernel<128, 1>), dim3((unsigned)grid), dim3(256), lds, s, pd);
            else hipLaunchKernelGGL((conv3f3b_kernel<64, 1>), dim3((unsigned)grid), dim3(256), lds, s, pd);
        } else {
            if (pd.total_wg) {
                if (wide) hipLaunchKernelGGL((conv3f3b_kernel<128, 3, true>), dim3((unsigned)grid_l), dim3(256), lds, s, pd);
                else hipLaunchKernelGGL((conv3f3b_kernel<64, 3, true>), dim3((unsigned)grid_l), dim3(256), lds, s, pd);
            } else if (wide) hipLaunchKernelGGL((conv3f3b_kernel<128, 3>), dim3((unsigned)grid), dim3(256), lds, s, pd);
            else hipLaunchKernelGGL((conv3f3b_kernel<64, 3>), dim3((unsigned)grid), dim3(256), lds, s, pd);
        }
        DPC_LAUNCH_CHECK();
        return DPC_OK;
    }
    // the 8-frame tile keeps the GroupNorm partial-sum count of the 4-frame tiling (tiles x 2 == tiles8 x 4) iff F % 8 == 0
    const bool tall = variant == 1;
    const int tf = tall ? 8 : 4;
    const long long tiles = (long long)p.B * ((p.F + tf - 1) / tf) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const size_t lds = (size_t)(tf + 2) * HH * HWD * PST;
    if (wide) {
        const long long grid = tiles * (p.Npad / 128);
        DPC_REQUIRE(grid < (1ll << 31), "conv3f3: grid too large");
        hipLaunchKernelGGL((conv3f3_kernel<128, 2, 3, 2>), dim3((unsigned)grid), dim3(256), lds, s, pd);
    } else {
        DPC_REQUIRE(p.Npad % 64 == 0, "conv3f3: Npad must be a multiple of 64");
        const long long grid = tiles * (p.Npad / 64);
        DPC_REQUIRE(grid < (1ll << 31), "conv3f3: grid too large");
        if (tall) hipLaunchKernelGGL((conv3f3_kernel<64, 4, 3, 2>), dim3((unsigned)grid), dim3(256), lds, s, pd);
        else hipLaunchKernelGGL((conv3f3_kernel<64, 2, 3, 2>), dim3((unsigned)grid), dim3(256), lds, s, pd);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

static int* g_ovf_flag = nullptr;
int* f16x3_weight_overflow_flag() {
    if (!g_ovf_flag) {
        if (hipMalloc(&g_ovf_flag, sizeof(int)) != hipSuccess) return nullptr;
        (void)hipMemset(g_ovf_flag, 0, sizeof(int));
    }
    return g_ovf_flag;
}
int f16x3_weight_overflow_check(const char* who) {
    if (!g_ovf_flag) return DPC_OK;
    int v = 0;
    DPC_HIP(hipMemcpy(&v, g_ovf_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (!v) return DPC_OK;
    (void)hipMemset(g_ovf_flag, 0, sizeof(int));
    return fail(DPC_ERR_STATE, std::string(who) + ": a weight exceeds the f16x3 range (|w| > 15.99 after the 2^12 pre-scale); "
                               "set DPC_CONV_MODE / DPC_IGEMM_MODE / DPC_ATTN_MODE / DPC_STEM_MODE to x6 or f32");
}

// ---- weight pre-split: reference [N][K][3][3][3] fp32 -> [27][kchunks][Npad][2 planes][16] fp16 (scaled by 2^12)
__global__ void pack_weights_f3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int N, int Npad, int K,
                                       int kchunks, int ntaps, int* __restrict__ ovf) {
    const long long total = (long long)ntaps * kchunks * Npad * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % 16);
        long long r = i / 16;
        const int n = (int)(r % Npad);
        r /= Npad;
        const int kc = (int)(r % kchunks);
        const int tap = (int)(r / kchunks);
        const int c = kc * 16 + kk;
        float v = 0.f;
        if (n < N && c < K) {
            v = w[((long long)n * K + c) * ntaps + tap] * f3::SW;
            if (!(fabsf(v) <= 65504.f)) atomicOr(ovf, 1);
            v = f3::sat16(v);
        }
        const unsigned p1 = f3::cvt_pk_f16(v, 0.f) & 0xffffu;
        const float h1 = (float)__builtin_bit_cast(f3::f16x2, p1).x;
        const unsigned p2 = f3::cvt_pk_f16(v - h1, 0.f) & 0xffffu;
        unsigned short* dst = wp + (((long long)tap * kchunks + kc) * Npad + n) * 32 + kk;
        dst[0] = (unsigned short)p1;
        dst[16] = (unsigned short)p2;
    }
}

int launch_pack_weights_f3(const float* w, void* wp, int N, int Npad, int K, hipStream_t s, int ntaps) {
    const int kchunks = (K + 15) / 16;
    const long long total = (long long)ntaps * kchunks * Npad * 16;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_f3_kernel, dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), N, Npad,
                       K, kchunks, ntaps, f16x3_weight_overflow_flag());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
