// 7x7x7 stem convolution (init_conv of Unet3D_with_Conv3D, video_diffusion_pytorch_conv3d.py:392: nn.Conv3d(channels, dim,
// 7, padding 3)) with fp32 semantics on the bf16 matrix cores (exact 3-way split, 6 MFMAs per product, see conv3x6.hip).
//
// The gather-per-K-element kernel in igemm.hip (stem_kernel) spends its time issuing 4-byte global loads (r01: 34 % MFMA
// busy, 52 TF/s).  Here a workgroup stages the 10 x 10 x 14 input halo of its 4 x 4 x 8 output tile ONCE in LDS, already
// split into three bf16 planes and padded to 8 channels per point (16 B "slots"), and walks the 49 (df, dh) tap rows; a
// tap row's (dw, c) pairs are then one contiguous run of 8 slots = 64 k-values = 4 MFMA k-steps whose A fragments are plain
// 16-byte LDS reads (k-step s covers dw = 2s, 2s+1; the 8th w-tap and channels >= C carry zero weights: 42 of 64 k-values
// are useful for C = 6).  Rows are 16 slots = 256 B; rows whose (h >> 1) is odd are rotated by 8 slots, so the two rows a
// ds_read_b128 pass touches (h and h+2, see lane_hw in conv3x6.hip) always hit complementary bank halves: conflict-free.
// Weights are pre-split [49][4][n][3 planes][16] bf16 and read as fragments straight from L2/L1, one k-step ahead.
//
// stem7x6_kernel<true> (default, DPC_STEM_MODE=f16x3) is the same kernel with the 2-way fp16 operand split of conv3f3.hip
// (2 planes, 3 MFMAs per product, input pre-scaled by 2^4, weights by 2^12, epilogue rescale 2^-16).
#include <type_traits>

#include "common.h"

namespace dpc {

namespace s7 {
constexpr int TF = 4, TH = 4, TW = 8;
constexpr int HF = TF + 6, HH = TH + 6, HWL = TW + 6;      // 10 x 10 x 14 logical halo
constexpr int SLOTS = 16;                                  // slots per row (15 addressed: w + 2*3 + 1 <= 14)
constexpr int ROWB = SLOTS * 16;                           // 256 B
constexpr int PLANEB = HF * HH * ROWB;                     // 25 600 B per plane
constexpr int NPT = HF * HH * SLOTS;                       // point slots to fill (1600, incl. the zero slots 14, 15)

__device__ __forceinline__ void lane_hw(int i, int& h, int& w) {      // same service-group aware map as conv3x6.hip
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
__device__ __forceinline__ float lo_f32(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_f32(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }
// exact 3-way split of 2 floats -> one packed pair per plane
__device__ __forceinline__ void split3_2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = cvt_pk_bf16(a, b);
    const float r0 = a - lo_f32(p1), r1 = b - hi_f32(p1);
    p2 = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - lo_f32(p2), s1 = r1 - hi_f32(p2);
    p3 = cvt_pk_bf16(s0, s1);
}
typedef _Float16 f16x2_s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sat16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f); }
__device__ __forceinline__ void split2_2(float a, float b, unsigned& p1, unsigned& p2) {
    a = sat16x(a); b = sat16x(b);
    p1 = cvt_pk_f16(a, b);
    p2 = f16_sub_pk(a, b, p1);
}
constexpr float SA = 16.f, SW = 4096.f, DESCALE = 1.f / 65536.f;
}  // namespace s7

typedef __bf16 bf16x8_s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_s __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// CS: channels per point slot.  8 (16-byte slots): a tap row is 4 MFMA k-steps (dw = 2 ks + hh).  4 (8-byte slots, C <= 4 -- the prior
// denoiser's 2-channel stem, which filled 14 of 64 k-values of the 8-channel form): a tap row is 2 k-steps (dw = 4 ks + 2 hh, + 1), half
// the MFMAs; an A fragment is two adjacent 8-byte slots (8-byte aligned: ds_read2_b64), rows are 128 B and not rotated.
template <bool H3, int CS = 8>
__global__ __launch_bounds__(256, 2) void stem7x6_kernel(StemParams p, const unsigned char* __restrict__ wp6) {
    fp16_ovfl_enable();                                 // (common.h: operand conversions saturate in hardware)
    using namespace s7;
    static_assert(CS == 8 || (CS == 4 && H3), "4-channel slots exist for the f16x3 form only");
    constexpr int SLOTB = CS * 2, ROWB = SLOTS * SLOTB, PLANEB = HF * HH * ROWB, KS = CS / 2;     // (shadow the 8-channel constants of s7)
    constexpr int NP = H3 ? 2 : 3;                                // operand planes
    constexpr int WROWB = NP * 32;                                // packed weight bytes per (step, n)
    using frag_t = std::conditional_t<H3, f16x8_s, bf16x8_s>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];
    unsigned char* halo = smem7;                                  // [3 planes][HF][HH][16 slots][8 ch] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / 64;
    const int ntf = (p.F + TF - 1) / TF, nth = (p.H + TH - 1) / TH, ntw = (p.W + TW - 1) / TW;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = (bid % ntn) * 64;
    int t = bid / ntn;
    const int w0 = (t % ntw) * TW; t /= ntw;
    const int h0 = (t % nth) * TH; t /= nth;
    const int f0 = (t % ntf) * TF;
    const int b = t / ntf;
    const long long HWin = (long long)p.H * p.W;

    // ---- stage the halo: one thread per point slot; out-of-range points, slots 14/15 and channels >= C are zero
    for (int q = tid; q < NPT; q += 256) {
        const int slot = q % SLOTS, row = q / SLOTS;              // row = pf * HH + ph
        const int pf = row / HH, ph = row % HH;
        const int f = f0 - 3 + pf, h = h0 - 3 + ph, w = w0 - 3 + slot;
        float v[CS];
#pragma unroll
        for (int c = 0; c < CS; ++c) v[c] = 0.f;
        if (slot < HWL && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) {
            const float* src = p.x + (((long long)b * p.F + f) * p.Ctot + p.c_off) * HWin + (long long)h * p.W + w;
#pragma unroll
            for (int c = 0; c < CS; ++c)
                if (c < p.C) v[c] = src[(long long)c * HWin];
        }
        u32x4 q1 = {0, 0, 0, 0}, q2 = q1, q3 = q1;
#pragma unroll
        for (int c = 0; c < CS / 2; ++c) {
            unsigned a1, a2, a3 = 0;
            if constexpr (H3) split2_2(v[2 * c] * SA, v[2 * c + 1] * SA, a1, a2);
            else split3_2(v[2 * c], v[2 * c + 1], a1, a2, a3);
            q1[c] = a1; q2[c] = a2; q3[c] = a3;
        }
        if constexpr (CS == 8) {
            const int rslot = (slot + (((ph >> 1) & 1) << 3)) & 15;   // bank rotation of rows with odd (h >> 1)
            unsigned char* dst = halo + row * ROWB + rslot * 16;
            *reinterpret_cast<u32x4*>(dst) = q1;
            *reinterpret_cast<u32x4*>(dst + PLANEB) = q2;
            if constexpr (!H3) *reinterpret_cast<u32x4*>(dst + 2 * PLANEB) = q3;
        } else {
            unsigned char* dst = halo + row * ROWB + slot * 8;
            *reinterpret_cast<uint2*>(dst) = uint2{q1[0], q1[1]};
            *reinterpret_cast<uint2*>(dst + PLANEB) = uint2{q2[0], q2[1]};
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    int lh, lw;
    lane_hw(l31, lh, lw);
    // weight fragments: [(df*7+dh)*4 + ks][Npad][3][16] bf16 = 96 B per n
    const unsigned char* wlane = wp6 + ((long long)n0 + wn * 32 + l31) * WROWB + hh * 16;
    const long long wstep = (long long)p.Npad * WROWB;
    // Weight fragments (the MFMAs' B operand) live in a ring of RW register sets with STATIC slot indices: step s multiplies slot
    // s % RW, the set of step s + 1 is requested at the top of step s into slot (s + 1) % RW, and the set of step s - 1 keeps its
    // registers to the end of step s (common.h: mfma_keep -- six or twelve younger MFMAs -- so hipcc cannot hand a B operand's
    // registers to an LDS fragment read issued right behind its last reader: DESIGN.md 6.2).  RW divides the 7 KS steps of one df
    // iteration, so every slot has the same role at the loop's back edge and the loop-carried sets need no copies.
    constexpr int SPI = 7 * KS;                                   // steps per df iteration
    constexpr int RW = SPI % 3 == 0 ? 3 : (SPI % 4 == 0 ? 4 : 7);
    frag_t wr[RW][NP];
    auto ldw = [&](int step, frag_t (&w)[NP]) {
        const unsigned char* src = wlane + (long long)step * wstep;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) w[pl] = *reinterpret_cast<const frag_t*>(src + pl * 32);
    };
    ldw(0, wr[0]);
#pragma unroll
    for (int r = 1; r < RW; ++r)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) wr[r][pl] = wr[0][pl];     // (defined values for the first keeps)
    __syncthreads();

    for (int df = 0; df < 7; ++df) {
#pragma unroll
        for (int dh = 0; dh < 7; ++dh) {
            const int row0 = ((wm * 2 + df) * HH + lh + dh) * ROWB;           // frame wm*2 (+ mt), halo row lh + dh
            const int rot = (((lh + dh) >> 1) & 1) << 3;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int step = (df * 7 + dh) * KS + ks;
                const int sl = (dh * KS + ks) % RW;                // = step % RW (SPI % RW == 0); compile-time after unrolling
                frag_t (&wc)[NP] = wr[sl];
                if (step + 1 < 49 * KS) ldw(step + 1, wr[(sl + 1) % RW]);
                frag_t a[2][NP];
                if constexpr (CS == 8) {
                    const int slot = (lw + 2 * ks + hh + rot) & 15;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl)
                            a[mt][pl] = *reinterpret_cast<const frag_t*>(halo + pl * PLANEB + row0 + mt * (HH * ROWB) + slot * 16);
                } else {
                    const int off = (lw + 4 * ks + 2 * hh) * 8;               // two adjacent 4-channel slots: dw = 4 ks + 2 hh, + 1
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) {
                            const unsigned char* src = halo + pl * PLANEB + row0 + mt * (HH * ROWB) + off;
                            const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
                            a[mt][pl] = __builtin_bit_cast(frag_t, u32x4{lo.x, lo.y, hi.x, hi.y});
                        }
                }
                if constexpr (H3) {
                    constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};                       // small terms first
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][PA[term]], wc[PB[term]], acc[mt], 0, 0, 0);
                } else {
                    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // smallest terms first
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][PA[term]], wc[PB[term]], acc[mt], 0, 0, 0);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    frag_t (&sp)[NP] = wr[(sl + RW - 1) % RW];
                    if constexpr (NP == 2) mfma_keep(acc[mt], sp[0], sp[1]);
                    else mfma_keep(acc[mt], sp[0], sp[1], sp[2]);
                }
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        frag_t (&sp)[NP] = wr[(49 * KS - 1) % RW];
        if constexpr (NP == 2) mfma_keep(acc[mt], sp[0], sp[1]);
        else mfma_keep(acc[mt], sp[0], sp[1], sp[2]);
    }

    const int n = n0 + wn * 32 + l31;
    if (n < p.N) {
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
                    p.out[((((long long)b * p.F + f) * p.H + h) * p.W + w) * p.N + n] = acc[mt][r] * (H3 ? DESCALE : 1.f) + bv;
            }
        }
    }
}

// ---- channel-pair form (r03, f16x3 only; default).  The slot forms above pad a tap row's (dw, c) run to whole 16-byte point slots:
// 42 of 64 k-values useful at C = 6 (14 of 32 for the prior denoiser's C = 2 with 4-channel slots) -- a third to a half of the
// MFMAs multiply zeros, and the kernel is matrix-bound.  Here the halo is stored per CHANNEL PAIR: plane (cp, split plane) holds
// 100 rows of 4-byte points (2 fp16 channels), and one MFMA k-step is  k = (dw 0..7, c in the pair)  = 16 contiguous bytes starting
// at point lw + 4 hh of the tap's row: 14 of 16 k-values useful for every even C (the 8th w-tap carries zero weights), ceil(C / 2)
// k-steps per tap row -- 3 instead of 4 at C = 6, 1 instead of 2 at C = 2.  The fragments are only 4-byte aligned (point pitch 4 B),
// so they are read as four dwords; rows are 24 dwords apart: the four rows of a 32-lane pass start at banks 0 / 24 / 16 / 8 and each
// touches 8 consecutive dwords -- conflict-free without rotation (same-address lanes broadcast).
namespace s7p {
constexpr int ROWB = 96;                                   // 24 dwords: 15 points addressed (w + 7 <= 14), the rest padding
constexpr int PLANEB = s7::HF * s7::HH * ROWB;             // 9 600 B per (channel pair, split plane)
}

template <int NCP>
__global__ __launch_bounds__(256, 2) void stem7p_kernel(StemParams p, const unsigned char* __restrict__ wp6) {
    fp16_ovfl_enable();                                 // (common.h: operand conversions saturate in hardware)
    using namespace s7;
    constexpr int ROWB = s7p::ROWB, PLANEB = s7p::PLANEB;
    constexpr int WROWB = 64;                                     // packed weight bytes per (step, n): 2 planes x 16 fp16
    extern __shared__ __attribute__((aligned(16))) unsigned char smem7[];
    unsigned char* halo = smem7;                                  // [NCP][2 planes][HF][HH][24 dwords]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / 64;
    const int ntf = (p.F + TF - 1) / TF, nth = (p.H + TH - 1) / TH, ntw = (p.W + TW - 1) / TW;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = (bid % ntn) * 64;
    int t = bid / ntn;
    const int w0 = (t % ntw) * TW; t /= ntw;
    const int h0 = (t % nth) * TH; t /= nth;
    const int f0 = (t % ntf) * TF;
    const int b = t / ntf;
    const long long HWin = (long long)p.H * p.W;

    // ---- stage the halo: one thread per point (16 per row: points 14, 15, out-of-range points and channels >= C are zero).
    // (Issuing all 7 rounds of loads before the first conversion, with unconditional loads, was measured SLOWER: 4.05 vs 3.69 ms at
    // C = 6 -- the second workgroup of the CU already hides this phase.)
    for (int q = tid; q < NPT; q += 256) {
        const int slot = q % SLOTS, row = q / SLOTS;              // row = pf * HH + ph
        const int pf = row / HH, ph = row % HH;
        const int f = f0 - 3 + pf, h = h0 - 3 + ph, w = w0 - 3 + slot;
        float v[2 * NCP];
#pragma unroll
        for (int c = 0; c < 2 * NCP; ++c) v[c] = 0.f;
        if (slot < HWL && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) {
            const float* src = p.x + (((long long)b * p.F + f) * p.Ctot + p.c_off) * HWin + (long long)h * p.W + w;
#pragma unroll
            for (int c = 0; c < 2 * NCP; ++c)
                if (c < p.C) v[c] = src[(long long)c * HWin];
        }
        unsigned char* dst = halo + row * ROWB + slot * 4;
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) {
            unsigned a1, a2;
            split2_2(v[2 * cp] * SA, v[2 * cp + 1] * SA, a1, a2);
            *reinterpret_cast<unsigned*>(dst + (2 * cp) * PLANEB) = a1;
            *reinterpret_cast<unsigned*>(dst + (2 * cp + 1) * PLANEB) = a2;
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    int lh, lw;
    lane_hw(l31, lh, lw);
    // weight fragments: [(df*7+dh) * NCP + cp][Npad][2 planes][16] fp16, k = dw * 2 + (c & 1)
    const unsigned char* wlane = wp6 + ((long long)n0 + wn * 32 + l31) * WROWB + hh * 16;
    const long long wstep = (long long)p.Npad * WROWB;
    // ring of weight sets with static slot indices, as in stem7x6_kernel (spent B operands keep their registers for one more k-step)
    constexpr int SPI = 7 * NCP;
    constexpr int RW = SPI % 3 == 0 ? 3 : 7;
    f16x8_s wr[RW][2];
    auto ldw = [&](int step, f16x8_s (&w)[2]) {
        const unsigned char* src = wlane + (long long)step * wstep;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) w[pl] = *reinterpret_cast<const f16x8_s*>(src + pl * 32);
    };
    ldw(0, wr[0]);
#pragma unroll
    for (int r = 1; r < RW; ++r)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wr[r][pl] = wr[0][pl];
    __syncthreads();

    const int abase = (wm * 2 * HH + lh) * ROWB + (lw + 4 * hh) * 4;      // frame wm*2 (+ mt + df), halo row lh (+ dh), point lw + 4 hh
    for (int df = 0; df < 7; ++df) {
#pragma unroll
        for (int dh = 0; dh < 7; ++dh) {
            const unsigned char* arow = halo + abase + (df * HH + dh) * ROWB;
#pragma unroll
            for (int cp = 0; cp < NCP; ++cp) {
                const int step = (df * 7 + dh) * NCP + cp;
                const int sl = (dh * NCP + cp) % RW;               // = step % RW, compile-time after unrolling
                f16x8_s (&wc)[2] = wr[sl];
                if (step + 1 < 49 * NCP) ldw(step + 1, wr[(sl + 1) % RW]);
                f16x8_s a[2][2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const unsigned* src = reinterpret_cast<const unsigned*>(arow + (2 * cp + pl) * PLANEB + mt * (HH * ROWB));
                        a[mt][pl] = __builtin_bit_cast(f16x8_s, u32x4{src[0], src[1], src[2], src[3]});
                    }
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};                       // small terms first
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][PA[term]], wc[PB[term]], acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) mfma_keep(acc[mt], wr[(sl + RW - 1) % RW][0], wr[(sl + RW - 1) % RW][1]);
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) mfma_keep(acc[mt], wr[(49 * NCP - 1) % RW][0], wr[(49 * NCP - 1) % RW][1]);

    const int n = n0 + wn * 32 + l31;
    if (n < p.N) {
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
                    p.out[((((long long)b * p.F + f) * p.H + h) * p.W + w) * p.N + n] = acc[mt][r] * DESCALE + bv;
            }
        }
    }
}


bool stem7x6_supported(int C, int k) { return k == 7 && C >= 1 && C <= 8; }
size_t stem7x6_packed_bytes(int Npad) { return (size_t)196 * Npad * 96; }      // sized for 3 planes; f16x3 uses 64 of the 96 B

static bool stem_h3() { return modes_current().stem != 1; }
// 4-channel slots (half the MFMAs) for stems of at most 4 input channels in the f16x3 form; DPC_STEM_CS4=0 keeps the 8-channel form
static bool stem_cs4(int C) {
    static const int ok = debug_switch("DPC_STEM_CS4", 1);
    return ok && C <= 4 && stem_h3();
}

// channel-pair form (stem7p_kernel) for the f16x3 arithmetic; DPC_STEM_PAIRS=0 keeps the slot forms (A/B)
// It is taken where it saves k-steps -- C <= 2 (1 instead of 2 per tap row) and C = 5, 6 (3 instead of 4); at equal MFMA counts
// (C = 3, 4, 7, 8) the slot forms' aligned 16-byte fragment reads are ~3 % faster (tools/bench_stem.py).  Returns the pair count or 0.
static int stem_ncp(int C) {
    static const int ok = debug_switch("DPC_STEM_PAIRS", 1);
    const int n = (C + 1) / 2;
    return (ok && stem_h3() && (n == 1 || n == 3)) ? n : 0;
}

template <int NCP>
static int launch_p7(const StemParams& p, const void* wp6, unsigned grid, hipStream_t s) {
    constexpr int LDS = 2 * NCP * s7p::PLANEB;
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)stem7p_kernel<NCP>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        once = true;
    }
    hipLaunchKernelGGL(stem7p_kernel<NCP>, dim3(grid), dim3(256), LDS, s, p, (const unsigned char*)wp6);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_stem7x6(const StemParams& p, const void* wp6, hipStream_t s) {
    using namespace s7;
    DPC_REQUIRE(p.Npad % 64 == 0 && wp6, "stem7x6: Npad must be a multiple of 64 and weights packed");
    DPC_REQUIRE(p.C >= 1 && p.C <= 8, "stem7x6: at most 8 input channels");
    if (p.M == 0) return DPC_OK;
    const int B = p.BF / p.F;
    const long long tiles = (long long)B * ((p.F + TF - 1) / TF) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const long long grid = tiles * (p.Npad / 64);
    DPC_REQUIRE(grid < (1ll << 31), "stem7x6: grid too large");
    const bool h3 = stem_h3(), cs4 = stem_cs4(p.C);
    const size_t lds = cs4 ? (size_t)PLANEB : (h3 ? 2 : 3) * (size_t)PLANEB;       // (4-channel slots: 2 planes of half the size)
    ProfScope prof(PROF_STEM, 2.0 * (double)p.M * p.N * 343.0 * p.C, 4.0 * ((double)p.M * p.N + (double)p.M * p.C), s);
    if (const int ncp = stem_ncp(p.C)) return ncp == 1 ? launch_p7<1>(p, wp6, (unsigned)grid, s) : launch_p7<3>(p, wp6, (unsigned)grid, s);
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)stem7x6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PLANEB));
        DPC_HIP(hipFuncSetAttribute((const void*)stem7x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PLANEB));
        once = true;
    }
    if (cs4) hipLaunchKernelGGL((stem7x6_kernel<true, 4>), dim3((unsigned)grid), dim3(256), lds, s, p, (const unsigned char*)wp6);
    else if (h3) hipLaunchKernelGGL(stem7x6_kernel<true>, dim3((unsigned)grid), dim3(256), lds, s, p, (const unsigned char*)wp6);
    else hipLaunchKernelGGL(stem7x6_kernel<false>, dim3((unsigned)grid), dim3(256), lds, s, p, (const unsigned char*)wp6);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// reference weight [N][C][7][7][7] fp32 -> [(df*7+dh)*4 + ks][Npad][3 planes][16] bf16, k = (dw - 2 ks) * 8 + c
// (cs4: [(df*7+dh)*2 + ks][Npad][2 planes][16] fp16, k = (dw - 4 ks) * 4 + c)
// (pairs: [(df*7+dh) * ncp + cp][Npad][2 planes][16] fp16, k = dw * 2 + (c - 2 cp); ncp = ceil(C / 2) k-steps per tap row)
__global__ void pack_stem7x6_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int N, int Npad, int C, int h3, int cs4,
                                    int ncp, int* __restrict__ ovf) {
    const long long total = (ncp ? 49ll * ncp : cs4 ? 98ll : 196ll) * Npad * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % 16);
        long long r = i / 16;
        const int n = (int)(r % Npad);
        const int step = (int)(r / Npad);
        const int ks = ncp ? step % ncp : cs4 ? (step & 1) : (step & 3), tap = ncp ? step / ncp : cs4 ? (step >> 1) : (step >> 2);
        const int df = tap / 7, dh = tap % 7;
        const int dw = ncp ? (kk >> 1) : cs4 ? 4 * ks + (kk >> 2) : 2 * ks + (kk >> 3);
        const int c = ncp ? 2 * ks + (kk & 1) : cs4 ? (kk & 3) : (kk & 7);
        float v = 0.f;
        if (n < N && c < C && dw < 7) v = w[(((long long)n * C + c) * 7 + df) * 49 + dh * 7 + dw];
        if (h3) {
            v = v * s7::SW;
            if (!(fabsf(v) <= 65504.f)) atomicOr(ovf, 1);
            v = s7::sat16(v);
            const unsigned h1 = s7::cvt_pk_f16(v, 0.f) & 0xffffu;
            const unsigned h2 = s7::cvt_pk_f16(v - (float)__builtin_bit_cast(s7::f16x2_s, h1).x, 0.f) & 0xffffu;
            unsigned short* d3 = wp + ((long long)step * Npad + n) * 32 + kk;
            d3[0] = (unsigned short)h1;
            d3[16] = (unsigned short)h2;
            continue;
        }
        const unsigned p1 = s7::cvt_pk_bf16(v, 0.f) & 0xffffu;
        const float r1 = v - __uint_as_float(p1 << 16);
        const unsigned p2 = s7::cvt_pk_bf16(r1, 0.f) & 0xffffu;
        const float r2 = r1 - __uint_as_float(p2 << 16);
        const unsigned p3 = s7::cvt_pk_bf16(r2, 0.f) & 0xffffu;
        unsigned short* dst = wp + ((long long)step * Npad + n) * 48 + kk;
        dst[0] = (unsigned short)p1;
        dst[16] = (unsigned short)p2;
        dst[32] = (unsigned short)p3;
    }
}

int launch_pack_stem7x6(const float* w, void* wp6, int N, int Npad, int C, hipStream_t s) {
    const int cs4 = stem_cs4(C) ? 1 : 0;
    const int ncp = stem_ncp(C);
    const long long total = (ncp ? 49ll * ncp : cs4 ? 98ll : 196ll) * Npad * 16;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_stem7x6_kernel, dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp6), N, Npad, C,
                       stem_h3() ? 1 : 0, cs4, ncp, f16x3_weight_overflow_flag());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
