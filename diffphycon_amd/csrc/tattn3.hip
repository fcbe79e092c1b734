// Fused temporal attention block, persistent and weight-stationary, on the fp16 matrix cores with split operands (f16x3):
//   y = x + to_out( softmax( (q*s) k^T + bias ) v ),  q,k,v = to_qkv(LayerNorm_c(x)),  rotary on q,k
// Reference: video_diffusion_pytorch_conv3d.py:165-184 (LayerNorm/PreNorm), :276-352 (Attention), :382,394,442.
//
// Dataflow of tattn6.hip (one wave = one pixel = one 32-token sequence = one 32-row MFMA tile, every intermediate produced
// in the register layout the next MFMA consumes, qkv never in HBM) with two changes that remove its stalls:
//  * at C = 64 the pre-split weights of ALL four heads (to_qkv 96 KB + to_out 32 KB as 2 fp16 planes), the rotary tables and
//    the relative-position bias fit the 160 KB LDS of a CU together.  One 512-thread workgroup per CU loads them once and
//    its 8 waves then loop over pixels with no barrier and no weight traffic at all (tattn6 re-staged 48 KB per head per
//    4 pixels behind two barriers: 30 % of its time), prefetching the next pixel's rows during the current one;
//  * f16x3 arithmetic (conv3f3.hip): 3 MFMAs per product instead of 6 and a 2-way register split (3 VALU / element instead
//    of 5.5).  Operand scales are powers of two chosen per tensor: LayerNorm output, q, k, v, o: 2^4; softmax numerators
//    (<= 1): 2^10; weights: 2^12; every accumulator is rescaled exactly where it is consumed.
#include "f16x3.h"

// r05 experiment: the two K = 32 contractions of a head on the EXACT fp32 matrix instruction instead of f16x3.
// A 32 x 32 projection accumulator is already an operand of v_mfma_f32_32x32x2_f32: register r of lane (l31, hh) holds element
// (l31, k = rowmap3(r, hh)), the same k in the two accumulators that are contracted, so S^T = sum_r mfma(kT[r], qT[r]) and
// O^T = sum_r mfma(vv[r], P[r]) need NO operand split (32 of a head's 80 split elements per lane each) -- for 16 64-cycle matrix
// instructions instead of 6 32-cycle ones.  Compile-time switches for the interleaved A/B (profiles/r05_*attn*): 0 = f16x3 (r04).
#ifndef DPC_TATTN_QK_F32
#define DPC_TATTN_QK_F32 0
#endif
#ifndef DPC_TATTN_PV_F32
#define DPC_TATTN_PV_F32 0
#endif

namespace dpc {

using namespace h3;

// acc += sum_r A[r] (x) B[r]: 16 exact fp32 products per lane pair on v_mfma_f32_32x32x2_f32 (lane (l31, hh): A[row l31][k = hh])
__device__ __forceinline__ void mfma_f32_k32(f32x16& acc, const f32x16& a, const f32x16& b) {
#ifdef DPC_DBG_NO_MFMA
    asm volatile("" : "+v"(acc) : "v"(a), "v"(b));
    return;
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc, 0, 0, 0);
}

// C = 64: all four heads resident, one launch.  C = 128: the weights of two heads (128 KB) fit, so the block runs as two
// launches over head pairs: pass 1 (heads 0, 1) writes x + its partial to_out sum to a workspace, pass 2 (heads 2, 3) adds its own sum.

template <int C_>
struct T3 {
    static constexpr int C = C_, KS = C / 16, NTC = C / 32;
    static constexpr int NH = C == 64 ? 4 : 2;                // heads resident per launch
    static constexpr int HEAD_QKV = 3 * KS * 2048;            // bytes per head: q | k | v, each KS x 2 planes x 1 KB
    static constexpr int HEAD_OUT = NTC * 2 * 2048;           // bytes per head: NTC column tiles x 2 k-steps x 2 planes x 1 KB
    static constexpr int QKV_RES = NH * HEAD_QKV, OUT_RES = NH * HEAD_OUT;
    static constexpr int TS = 36;                             // padded row stride (floats) of the rotary / bias tables in LDS
    static constexpr int OFF_COS = QKV_RES + OUT_RES, OFF_SIN = OFF_COS + 32 * TS * 4, OFF_BIAS = OFF_SIN + 32 * TS * 4;
    static constexpr int LDS_BYTES = OFF_BIAS + NH * 32 * TS * 4;          // 158720 (C = 64), 149504 (C = 128)
};
namespace t3 {
constexpr int TS = 36;
constexpr float SX = 16.f, SWGT = 4096.f;
constexpr float PROJ_DESCALE = 1.f / (SX * SWGT);  // accumulator -> true value of a projection
constexpr float SQK = 16.f, SP = 1024.f, SV = 16.f, SO = 16.f;
constexpr float LOG2E = 1.4426950408889634f, LOG2_SP = 10.f;          // SP = 2^10
}  // namespace t3

size_t tattn3_qkv_bytes(int C) { return (size_t)4 * 3 * (C / 16) * 2048; }
size_t tattn3_out_bytes(int C) { return (size_t)4 * (C / 32) * 2 * 2048; }

// to_qkv.weight [384][C] -> [head][q|k|v][ks][plane][32 rows][2][8];  to_out.weight [C][128] -> [head][nt][s][plane][32][2][8]
__global__ void pack_tattn3_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int C, int is_out, int total,
                                   int* __restrict__ ovf) {
    using namespace t3;
    const int KS = C / 16, NTC = C / 32;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // one value = 2 planes
    if (idx >= total) return;
    const int i = idx & 7, hh = (idx >> 3) & 1, r = (idx >> 4) & 31;
    int rest = idx >> 9;
    float v;
    long long o;
    if (!is_out) {
        const int ks = rest % KS; rest /= KS;
        const int part = rest % 3, hd = rest / 3;
        v = w[(long long)(part * 128 + hd * 32 + r) * C + 16 * ks + 8 * hh + i];
        o = ((long long)((hd * 3 + part) * KS + ks) * 2) * 512 + r * 16 + hh * 8 + i;
    } else {
        const int s = rest & 1; rest >>= 1;
        const int nt = rest % NTC, hd = rest / NTC;
        const int d = 16 * s + 4 * hh + (i & 3) + 8 * (i >> 2);      // k order of an accumulator operand (f16x3.h)
        v = w[(long long)(nt * 32 + r) * 128 + hd * 32 + d];
        o = ((long long)((hd * NTC + nt) * 2 + s) * 2) * 512 + r * 16 + hh * 8 + i;
    }
    v = v * SWGT;
    if (!(fabsf(v) <= 65504.f)) atomicOr(ovf, 1);
    v = sat16(v);
    const unsigned p1 = cvt_pk(v, 0.f) & 0xffffu;
    const unsigned p2 = cvt_pk(v - (float)__builtin_bit_cast(f16x2, p1).x, 0.f) & 0xffffu;
    dst[o] = (unsigned short)p1;
    dst[o + 512] = (unsigned short)p2;
}

int launch_pack_tattn3(const float* w, unsigned char* dst, int C, bool is_out, hipStream_t s) {
    const int total = is_out ? 4 * (C / 32) * 2 * 512 : 4 * 3 * (C / 16) * 512;
    hipLaunchKernelGGL(pack_tattn3_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(dst),
                       C, is_out ? 1 : 0, total, f16x3_weight_overflow_flag());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

__device__ __forceinline__ int rowmap3(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// FULL: F == 32, no token masks.  PASS: 0 = all heads in one launch (C = 64); 1 / 2 = first / second head pair (C = 128),
// `part` = [rows][C] partial to_out sums written by pass 1 and consumed by pass 2.
template <int C_, bool FULL, int PASS>
__global__ __launch_bounds__(512, 1) void tattn3_kernel(TattnParams p, const unsigned char* __restrict__ wq3,
                                                        const unsigned char* __restrict__ wo3, float* __restrict__ part) {
    using namespace t3;
    using G = T3<C_>;
    constexpr int C = G::C, KS = G::KS, NTC = G::NTC, NH = G::NH, HEAD_QKV = G::HEAD_QKV, HEAD_OUT = G::HEAD_OUT;
    constexpr int QKV_RES = G::QKV_RES, OUT_RES = G::OUT_RES, OFF_COS = G::OFF_COS, OFF_SIN = G::OFF_SIN, OFF_BIAS = G::OFF_BIAS;
    constexpr int HD0 = PASS == 2 ? 2 : 0;             // first resident head
    constexpr bool PREFETCH = C == 64 && FULL;         // C = 128 and the masked (F < 32) form: the row registers are needed elsewhere
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int loff = l31 * 32 + hh * 16;
    const int F = p.F;
    hw_sat_enable();                                   // (f16x3.h: operand conversions saturate in hardware)

    // ---- one-time fill: weight images verbatim, rotary / bias tables with padded rows
    for (int q = tid; q < QKV_RES / 16; q += 512)
        reinterpret_cast<uint4*>(smem3)[q] = reinterpret_cast<const uint4*>(wq3 + (size_t)HD0 * HEAD_QKV)[q];
    for (int q = tid; q < OUT_RES / 16; q += 512)
        reinterpret_cast<uint4*>(smem3 + QKV_RES)[q] = reinterpret_cast<const uint4*>(wo3 + (size_t)HD0 * HEAD_OUT)[q];
    {
        float* cosL = reinterpret_cast<float*>(smem3 + OFF_COS);
        float* sinL = reinterpret_cast<float*>(smem3 + OFF_SIN);
        float* biasL = reinterpret_cast<float*>(smem3 + OFF_BIAS);
        for (int q = tid; q < 32 * 32; q += 512) {
            const int r = q >> 5, c = q & 31;
            cosL[r * TS + c] = r < F ? p.rot_cos[r * 32 + c] : 1.f;
            sinL[r * TS + c] = r < F ? p.rot_sin[r * 32 + c] : 0.f;
        }
        // (r04: the bias enters the softmax as an exponent of two -- scores and bias carry log2(e), one v_exp_f32 per probability)
        for (int q = tid; q < NH * 32 * 32; q += 512) biasL[(q >> 5) * TS + (q & 31)] = p.bias32[HD0 * 1024 + q] * LOG2E;
    }
    __syncthreads();

    const long long HW = p.HW;
    const long long nwaves = (long long)gridDim.x * 8;
    long long gp = (long long)blockIdx.x * 8 + wave;
    const float qscale = 0.17677669529663687f;
    const bool tok = FULL || l31 < F;
    const int ti = tok ? l31 : 0;
    const unsigned char* cosL = smem3 + OFF_COS + (ti * TS + 4 * hh) * 4;
    const unsigned char* sinL = smem3 + OFF_SIN + (ti * TS + 4 * hh) * 4;
    const unsigned char* biasL = smem3 + OFF_BIAS + (l31 * TS + 4 * hh) * 4;

    // rows of a pixel: token f at row0 + f*HW; lane (token l31, half hh) holds channels 16ks + 8hh .. +7 of k-step ks
    auto row0_of = [&](long long g) {
        const unsigned bq = (unsigned)g / (unsigned)HW;             // npix < 2^31 (checked by the launcher)
        return (long long)bq * F * HW + (long long)((unsigned)g - bq * (unsigned)HW);
    };
    f32x4 xr[KS][2];
    auto load_rows = [&](long long row0) {
        const float* src = p.x + (row0 + (long long)ti * HW) * C + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < 2; ++q) xr[ks][q] = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4 * q);
    };
    if (PREFETCH && gp < p.npix) load_rows(row0_of(gp));
    // MFMA B operands that are loaded from LDS or die right after their MFMAs keep their registers until two more 3-MFMA groups were
    // issued (common.h: mfma_keep): the projection order is V, K, Q so that a V-weight fragment is followed by six MFMAs
    // whose B operand (xs) stays live; q / p fragments are released inside the next product; the last two to_out fragments of a head
    // are carried into the first two groups of the next head (or pixel).
    f16x8 wo_carry[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wo_carry[i][pl] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};

    for (; gp < p.npix; gp += nwaves) {
        const long long row0 = row0_of(gp);
        if (!PREFETCH) load_rows(row0);
        // ---- LayerNorm over channels (lane pair), scale, split
        f16x8 xs[KS][2];
        {
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (!tok) xr[ks][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 v = xr[ks][q];
                    s += (v.x + v.y) + (v.z + v.w);
                }
            s += __shfl_xor(s, 32, 64);
            const float mean = s / (float)C;
            float q2 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 d = xr[ks][q] - mean;
                    q2 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            q2 += __shfl_xor(q2, 32, 64);
            const float inv = SX * (1.0f / sqrtf(q2 / (float)C + 1e-5f));      // (SX = 2^4 folded in: exact)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f32x4 n[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + 16 * ks + 8 * hh + 4 * q);
                    n[q] = (xr[ks][q] - mean) * inv * g;           // a masked token has x = mean = 0: stays exactly 0
                }
                split8(sat16h(n[0].x), sat16h(n[0].y), sat16h(n[0].z), sat16h(n[0].w), sat16h(n[1].x), sat16h(n[1].y),
                       sat16h(n[1].z), sat16h(n[1].w), xs[ks]);
            }
        }
        // ---- prefetch the next pixel's rows (consumed at the top of the next iteration)
        if (PREFETCH && gp + nwaves < p.npix) load_rows(row0_of(gp + nwaves));
        asm volatile("" ::: "memory");

        f32x16 y[NTC];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[nt][r] = 0.f;

#pragma unroll 1
        for (int hd = 0; hd < NH; ++hd) {
            const unsigned char* Wq = smem3 + hd * HEAD_QKV;
            const unsigned char* Wo = smem3 + QKV_RES + hd * HEAD_OUT;
            // ---- projections: Q^T, K^T (A = weights: lane = token, regs = head dims), V (A = x: lane = d, regs = token)
            f32x16 qT, kT, vv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { qT[r] = 0.f; kT[r] = 0.f; vv[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f16x8 wv[2], w[2];
                load_w2(Wq + 2 * KS * 2048, ks, loff, wv);
                mfma3(vv, xs[ks], wv);
                if (ks == 0) mfma_keep(vv, wo_carry[0][0], wo_carry[0][1]);
                mfma_order_point();                              // (MFMAs keep their source order across this point: K, Q follow V)
                load_w2(Wq + KS * 2048, ks, loff, w);
                mfma3(kT, w, xs[ks]);
                if (ks == 0) mfma_keep(kT, wo_carry[1][0], wo_carry[1][1]);
                load_w2(Wq, ks, loff, w);
                mfma3(qT, w, xs[ks]);
                mfma_keep(kT, wv[0], wv[1]);
                mfma_keep(qT, wv[0], wv[1]);
            }
            // ---- q * scale, rotary on q and k (pairs (2m, 2m+1) = registers (4jj, 4jj+1), (4jj+2, 4jj+3)); the results carry
            //      the operand pre-scale SQK
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosL + 32 * jj);
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinL + 32 * jj);
                // r04: one multiply + one fused multiply-add per rotated element (was scale, two multiplies, add); the scale
                // q * qscale * SQK / (SX SWGT) follows in the operand split below, which multiplies anyway
                const float q0 = qT[4 * jj], q1 = qT[4 * jj + 1], q2 = qT[4 * jj + 2], q3 = qT[4 * jj + 3];
                qT[4 * jj] = __builtin_fmaf(q0, c4.x, -(q1 * s4.x));
                qT[4 * jj + 1] = __builtin_fmaf(q1, c4.y, q0 * s4.y);
                qT[4 * jj + 2] = __builtin_fmaf(q2, c4.z, -(q3 * s4.z));
                qT[4 * jj + 3] = __builtin_fmaf(q3, c4.w, q2 * s4.w);
                const float k0 = kT[4 * jj], k1 = kT[4 * jj + 1], k2 = kT[4 * jj + 2], k3 = kT[4 * jj + 3];
                kT[4 * jj] = __builtin_fmaf(k0, c4.x, -(k1 * s4.x));
                kT[4 * jj + 1] = __builtin_fmaf(k1, c4.y, k0 * s4.y);
                kT[4 * jj + 2] = __builtin_fmaf(k2, c4.z, -(k3 * s4.z));
                kT[4 * jj + 3] = __builtin_fmaf(k3, c4.w, k2 * s4.w);
            }
            // ---- S^T[j][i] = k_j . q_i   (A = K: lane = key j; B = Q: lane = query i; k index = head dim in register order)
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            f16x8 qs[2][2];
            if constexpr (DPC_TATTN_QK_F32 != 0) {
                // raw accumulators as operands: st = (true k . q) / (PROJ_DESCALE^2): the scale joins the softmax multiplier below
                mfma_f32_k32(st, kT, qT);
                asm volatile("" : "+v"(st) : "v"(kT), "v"(qT));       // (operand registers stay allocated behind the last MFMA: 6.2)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) qs[i][pl] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            } else {
                f16x8 kk[2][2];
                split_acc_h<true>(qT, qscale * (PROJ_DESCALE * SQK), qs);
                split_acc_h<true>(kT, PROJ_DESCALE * SQK, kk);
                mfma3(st, kk[0], qs[0]);
                mfma3(st, kk[1], qs[1]);
            }
            // ---- bias + softmax over keys (lane-local + partner lane); registers 4jj .. 4jj+3 = keys 8jj + 4hh .. +3.
            //      r04: scores in units of log2: s' = s log2(e) + bias log2(e) (one fma), p SP = exp2(s' - (m' - log2 SP)) (sub + v_exp);
            //      the sum l carries SP as well and the operand split of P needs no multiplier
            float m = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(biasL + (hd * 32 * TS + 8 * jj) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    constexpr float SMUL = DPC_TATTN_QK_F32 != 0 ? LOG2E * 0.17677669529663687f * (PROJ_DESCALE * PROJ_DESCALE)
                                                                 : LOG2E / (SQK * SQK);
                    float sv = __builtin_fmaf(st[4 * jj + e], SMUL, b4[e]);
                    if (!FULL && 8 * jj + 4 * hh + e >= F) sv = -INFINITY;
                    st[4 * jj + e] = sv;
                    m = fmaxf(m, sv);
                }
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64)) - LOG2_SP;
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(st[r] - m);
                st[r] = e;
                l += e;
            }
            l += __shfl_xor(l, 32, 64);
            // ---- O^T[d][i] = sum_j V[j][d] P[i][j]   (A = V: lane = d; B = P: lane = query i; k index = key in register order)
            f32x16 oT;
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[r] = 0.f;
            f16x8 ps[2][2];
            if constexpr (DPC_TATTN_PV_F32 != 0) {
                // oT = (true V^T P) SP / PROJ_DESCALE: folded into the multiplier of the O split below
                mfma_f32_k32(oT, vv, st);
                asm volatile("" : "+v"(oT) : "v"(vv), "v"(st));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) ps[i][pl] = qs[i][pl];
            } else {
                f16x8 vs[2][2];
                split_acc_h<true>(vv, PROJ_DESCALE * SV, vs);
                split_acc_h<false>(st, 1.f, ps);                 // (st = p SP already)
                mfma3(oT, vs[0], ps[0]);
                mfma_keep(oT, qs[0][0], qs[0][1]);
                mfma3(oT, vs[1], ps[1]);
                mfma_keep(oT, qs[1][0], qs[1][1]);
            }
            // ---- Y[i][c] += sum_d O[i][d] Wout[c][hd*32+d]   (A = O: lane = token i; B = packed to_out slice); the softmax
            //      normalisation, the descale of P and V and the pre-scale of O are one multiplier
            {
                f16x8 os[2][2];
                // (l = SP x the sum of the probabilities; with the fp32 P V product oT carries 1 / PROJ_DESCALE instead of SV)
                split_acc_h<true>(oT, (DPC_TATTN_PV_F32 != 0 ? SO * PROJ_DESCALE : SO / SV) / l, os);
                f16x8 wo[2 * NTC][2];
#pragma unroll
                for (int g = 0; g < 2 * NTC; ++g) {                 // group g = (column tile g / 2, k-step g % 2)
                    load_w2(Wo, g, loff, wo[g]);
                    mfma3(y[g >> 1], os[g & 1], wo[g]);
                    if (g < 2) mfma_keep(y[g >> 1], ps[g][0], ps[g][1]);
                    else mfma_keep(y[g >> 1], wo[g - 2][0], wo[g - 2][1]);
                    mfma_order_point();
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wo_carry[i][pl] = wo[2 * NTC - 2 + i][pl];
            }
        }
        // ---- residual + store (lane = channel, regs = token)
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {           // residual loads first, unconditional (clamped token), then the stores
                const int i = rowmap3(r, hh);
                const int ic = (FULL || i < F) ? i : 0;
                const long long o = (row0 + (long long)ic * HW) * C + nt * 32 + l31;
                res[r] = PASS == 2 ? part[o] : p.x[o];      // pass 1 stores (its sum + x) into `part`: the same bits as part + x later
            }
            float* dstp = PASS == 1 ? part : p.out;
            unsigned omx = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rowmap3(r, hh);
                const float v = y[nt][r] * (1.f / (SO * SWGT)) + res[r];
                omx = max(omx, abs_bits(v));
                if (FULL || i < F) dstp[(row0 + (long long)i * HW) * C + nt * 32 + l31] = v;
            }
            if (PASS != 1 && p.oflag && omx > F16X3_ACT_LIMIT_BITS) atomicOr(p.oflag, 1);     // activation-range sentinel (common.h)
        }
    }
}


// ------------------------------------------------------------------------------------------------ 32 < F <= 64
// Two 32-token tiles per sequence (BASELINE configs[4], S128: 64 frames).  Same dataflow and arithmetic; what changes:
//  * one wave still owns one pixel, now with two row tiles: K and V of BOTH tiles are projected once per head and kept as
//    split fp16 operands, each query tile then runs Q projection -> S^T against both key tiles -> softmax over 64 keys ->
//    PV -> to_out.  That needs more than 256 registers per lane, so the workgroup has 4 waves (one per SIMD, 512 registers);
//    the two tiles give every MFMA chain an independent partner to interleave with.
//  * the [heads][F][F] bias table no longer fits next to the weights.  The reference's bias is a function of (key - query)
//    only (RelativePositionBias, ...conv3d.py:106-112: bucket(k_pos - q_pos) -> embedding), so the LDS holds the 127-entry
//    Toeplitz vector per head (`brel`, built and VERIFIED against the full table by dpc_unet3d_set_tables; a table that is
//    not Toeplitz keeps the unfused kernels).  Rotary tables have 64 rows.
//  * C = 64 keeps the output accumulators of both query tiles across the head loop (head outer, query tile inner); C = 128
//    would need > 512 registers for that, so it runs query tile outer / head inner and re-projects K, V per query tile.
template <int C_>
struct T3W {
    static constexpr int C = C_, KS = C / 16, NTC = C / 32;
    static constexpr int NH = C == 64 ? 4 : 2;
    static constexpr int HEAD_QKV = 3 * KS * 2048, HEAD_OUT = NTC * 2 * 2048;
    static constexpr int QKV_RES = NH * HEAD_QKV, OUT_RES = NH * HEAD_OUT;
    static constexpr int TS = 36;
    static constexpr int OFF_COS = QKV_RES + OUT_RES, OFF_SIN = OFF_COS + 64 * TS * 4, OFF_BREL = OFF_SIN + 64 * TS * 4;
    static constexpr int LDS_BYTES = OFF_BREL + NH * 128 * 4;              // 151552
};

template <int C_, int PASS>
__global__ __launch_bounds__(256, 1) void tattn3w_kernel(TattnParams p, const unsigned char* __restrict__ wq3,
                                                         const unsigned char* __restrict__ wo3, float* __restrict__ part) {
    using namespace t3;
    using G = T3W<C_>;
    constexpr int C = G::C, KS = G::KS, NTC = G::NTC, NH = G::NH, HEAD_QKV = G::HEAD_QKV, HEAD_OUT = G::HEAD_OUT;
    constexpr int QKV_RES = G::QKV_RES, OUT_RES = G::OUT_RES, OFF_COS = G::OFF_COS, OFF_SIN = G::OFF_SIN, OFF_BREL = G::OFF_BREL;
    constexpr int HD0 = PASS == 2 ? 2 : 0;
    constexpr bool TILE_OUTER = true;      // C = 64 in the head-outer order needs 62 spilled registers per lane (kept for A/B runs)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int loff = l31 * 32 + hh * 16;
    const int F = p.F;
    hw_sat_enable();

    for (int q = tid; q < QKV_RES / 16; q += 256)
        reinterpret_cast<uint4*>(smem3)[q] = reinterpret_cast<const uint4*>(wq3 + (size_t)HD0 * HEAD_QKV)[q];
    for (int q = tid; q < OUT_RES / 16; q += 256)
        reinterpret_cast<uint4*>(smem3 + QKV_RES)[q] = reinterpret_cast<const uint4*>(wo3 + (size_t)HD0 * HEAD_OUT)[q];
    {
        float* cosW = reinterpret_cast<float*>(smem3 + OFF_COS);
        float* sinW = reinterpret_cast<float*>(smem3 + OFF_SIN);
        float* brelW = reinterpret_cast<float*>(smem3 + OFF_BREL);
        for (int q = tid; q < 64 * 32; q += 256) {
            const int r = q >> 5, c = q & 31;
            cosW[r * TS + c] = r < F ? p.rot_cos[r * 32 + c] : 1.f;
            sinW[r * TS + c] = r < F ? p.rot_sin[r * 32 + c] : 0.f;
        }
        for (int q = tid; q < NH * 128; q += 256) brelW[q] = p.brel[HD0 * 128 + q];
    }
    __syncthreads();

    const long long HW = p.HW;
    const long long nwaves = (long long)gridDim.x * 4;
    const float qscale = 0.17677669529663687f;
    const bool tokv[2] = {true, 32 + l31 < F};                 // F > 32 (launcher): tile 0 is always full
    const int ti[2] = {l31, tokv[1] ? 32 + l31 : 0};
    const float* brelL = reinterpret_cast<const float*>(smem3 + OFF_BREL);

    auto row0_of = [&](long long g) {
        const unsigned bq = (unsigned)g / (unsigned)HW;
        return (long long)bq * F * HW + (long long)((unsigned)g - bq * (unsigned)HW);
    };
    // rotary on a 32x32 projection accumulator (lane = token, registers = head dims), tile T; `mul` = descale (* q scale)
    auto rotary = [&](f32x16& a, int T, float mul) {
        const unsigned char* cosL = smem3 + OFF_COS + (ti[T] * TS + 4 * hh) * 4;
        const unsigned char* sinL = smem3 + OFF_SIN + (ti[T] * TS + 4 * hh) * 4;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosL + 32 * jj);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinL + 32 * jj);
            const float a0 = a[4 * jj] * mul, a1 = a[4 * jj + 1] * mul, a2 = a[4 * jj + 2] * mul, a3 = a[4 * jj + 3] * mul;
            a[4 * jj] = __fadd_rn(__fmul_rn(a0, c4.x), __fmul_rn(-a1, s4.x));
            a[4 * jj + 1] = __fadd_rn(__fmul_rn(a1, c4.y), __fmul_rn(a0, s4.y));
            a[4 * jj + 2] = __fadd_rn(__fmul_rn(a2, c4.z), __fmul_rn(-a3, s4.z));
            a[4 * jj + 3] = __fadd_rn(__fmul_rn(a3, c4.w), __fmul_rn(a2, s4.w));
        }
    };

    // spent MFMA B operands keep their registers for two more 3-MFMA groups (as in tattn3_kernel; common.h: mfma_keep)
    f16x8 wo_carry[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wo_carry[i][pl] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (long long gp = (long long)blockIdx.x * 4 + wave; gp < p.npix; gp += nwaves) {
        const long long row0 = row0_of(gp);
        // ---- rows of both tiles: load, LayerNorm over channels (lane pair), scale, split
        f16x8 xs[2][KS][2];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            f32x4 xr[KS][2];
            const float* src = p.x + (row0 + (long long)ti[T] * HW) * C + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) xr[ks][q] = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4 * q);
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (!tokv[T]) xr[ks][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const f32x4 v = xr[ks][q];
                    s += (v.x + v.y) + (v.z + v.w);
                }
            s += __shfl_xor(s, 32, 64);
            const float mean = s / (float)C;
            float q2 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 d = xr[ks][q] - mean;
                    q2 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
                }
            q2 += __shfl_xor(q2, 32, 64);
            const float inv = SX * (1.0f / sqrtf(q2 / (float)C + 1e-5f));      // (SX = 2^4 folded in: exact)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f32x4 n[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + 16 * ks + 8 * hh + 4 * q);
                    n[q] = (xr[ks][q] - mean) * inv * g;
                }
                split8(sat16h(n[0].x), sat16h(n[0].y), sat16h(n[0].z), sat16h(n[0].w), sat16h(n[1].x), sat16h(n[1].y),
                       sat16h(n[1].z), sat16h(n[1].w), xs[T][ks]);
            }
        }

        // K^T (rotary applied) and V of both tiles for resident head hd, as split MFMA operands
        auto project_kv = [&](int hd, f16x8 (&kk)[2][2][2], f16x8 (&vs)[2][2][2]) {
            const unsigned char* Wq = smem3 + hd * HEAD_QKV;
            f32x16 kT[2], vv[2];
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int r = 0; r < 16; ++r) { kT[T][r] = 0.f; vv[T][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f16x8 w[2], wv[2];
                load_w2(Wq + 2 * KS * 2048, ks, loff, wv);       // V first: its weight fragment (B operand) is followed by the six K MFMAs
                mfma3(vv[0], xs[0][ks], wv);
                if (ks == 0) mfma_keep(vv[0], wo_carry[0][0], wo_carry[0][1]);
                mfma3(vv[1], xs[1][ks], wv);
                if (ks == 0) mfma_keep(vv[1], wo_carry[1][0], wo_carry[1][1]);
                mfma_order_point();
                load_w2(Wq + KS * 2048, ks, loff, w);
                mfma3(kT[0], w, xs[0][ks]);
                mfma3(kT[1], w, xs[1][ks]);
                mfma_keep(kT[0], wv[0], wv[1]);
                mfma_keep(kT[1], wv[0], wv[1]);
                mfma_order_point();
            }
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                rotary(kT[T], T, PROJ_DESCALE * SQK);
                split_acc_h<true>(kT[T], 1.f, kk[T]);
                split_acc_h<true>(vv[T], PROJ_DESCALE * SV, vs[T]);
            }
        };
        // query tile `it` of head hd against both key tiles; adds its to_out contribution to y
        auto attend = [&](int hd, int it, const f16x8 (&kk)[2][2][2], const f16x8 (&vs)[2][2][2], f32x16 (&y)[NTC]) {
            const unsigned char* Wq = smem3 + hd * HEAD_QKV;
            const unsigned char* Wo = smem3 + QKV_RES + hd * HEAD_OUT;
            f32x16 qT;
#pragma unroll
            for (int r = 0; r < 16; ++r) qT[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f16x8 w[2];
                load_w2(Wq, ks, loff, w);
                mfma3(qT, w, xs[it][ks]);
                if (ks < 2) mfma_keep(qT, wo_carry[ks][0], wo_carry[ks][1]);     // (the previous attend's last to_out fragments)
            }
            rotary(qT, it, qscale * (PROJ_DESCALE * SQK));
            f32x16 st[2];
            f16x8 qs[2][2];
            {
                split_acc_h<true>(qT, 1.f, qs);
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[jt][r] = 0.f;
                    mfma3(st[jt], kk[jt][0], qs[0]);
                    mfma3(st[jt], kk[jt][1], qs[1]);
                    mfma_order_point();
                }
            }
            // bias[h][i][j] = brel[h][j - i + 63]; registers 4jj .. 4jj+3 of key tile jt = keys 32jt + 8jj + 4hh .. +3
            const float* bq = brelL + hd * 128 + 63 - (32 * it + l31) + 4 * hh;
            float m = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = 32 * jt + 8 * jj + e;             // + 4hh (folded into bq and the mask)
                        float sv = st[jt][4 * jj + e] * (1.f / (SQK * SQK)) + bq[j];
                        if (jt == 1 && j + 4 * hh >= F) sv = -INFINITY;
                        st[jt][4 * jj + e] = sv;
                        m = fmaxf(m, sv);
                    }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __expf(st[jt][r] - m);
                    st[jt][r] = e;
                    l += e;
                }
            l += __shfl_xor(l, 32, 64);
            f32x16 oT;
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[r] = 0.f;
            f16x8 ps[2][2][2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                split_acc_h<false>(st[jt], SP, ps[jt]);
                mfma3(oT, vs[jt][0], ps[jt][0]);
                if (jt == 0) mfma_keep(oT, qs[0][0], qs[0][1]);
                else mfma_keep(oT, ps[0][0][0], ps[0][0][1]);
                mfma3(oT, vs[jt][1], ps[jt][1]);
                if (jt == 0) mfma_keep(oT, qs[1][0], qs[1][1]);
                else mfma_keep(oT, ps[0][1][0], ps[0][1][1]);
            }
            f16x8 os[2][2];
            split_acc_h<true>(oT, (SO / (SP * SV)) / l, os);
            f16x8 wo[2 * NTC][2];
#pragma unroll
            for (int g = 0; g < 2 * NTC; ++g) {                     // group g = (column tile g / 2, k-step g % 2)
                load_w2(Wo, g, loff, wo[g]);
                mfma3(y[g >> 1], os[g & 1], wo[g]);
                if (g < 2) mfma_keep(y[g >> 1], ps[1][g][0], ps[1][g][1]);
                else mfma_keep(y[g >> 1], wo[g - 2][0], wo[g - 2][1]);
                mfma_order_point();
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) wo_carry[i][pl] = wo[2 * NTC - 2 + i][pl];
        };
        auto store_tile = [&](int it, const f32x16 (&y)[NTC]) {
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                float res[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = 32 * it + rowmap3(r, hh);
                    const int ic = i < F ? i : 0;
                    const long long o = (row0 + (long long)ic * HW) * C + nt * 32 + l31;
                    res[r] = PASS == 2 ? part[o] : p.x[o];
                }
                float* dstp = PASS == 1 ? part : p.out;
                unsigned omx = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = 32 * it + rowmap3(r, hh);
                    const float v = y[nt][r] * (1.f / (SO * SWGT)) + res[r];
                    omx = max(omx, abs_bits(v));
                    if (i < F) dstp[(row0 + (long long)i * HW) * C + nt * 32 + l31] = v;
                }
                if (PASS != 1 && p.oflag && omx > F16X3_ACT_LIMIT_BITS) atomicOr(p.oflag, 1);
                asm volatile("" ::: "memory");       // keep the next column tile's residual loads behind these stores (registers)
            }
        };

        if constexpr (!TILE_OUTER) {
            f32x16 y[2][NTC];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[it][nt][r] = 0.f;
#pragma unroll 1
            for (int hd = 0; hd < NH; ++hd) {
                f16x8 kk[2][2][2], vs[2][2][2];
                project_kv(hd, kk, vs);
                attend(hd, 0, kk, vs, y[0]);
                attend(hd, 1, kk, vs, y[1]);
            }
            store_tile(0, y[0]);
            store_tile(1, y[1]);
        } else {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                f32x16 y[NTC];
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[nt][r] = 0.f;
#pragma unroll 1
                for (int hd = 0; hd < NH; ++hd) {
                    f16x8 kk[2][2][2], vs[2][2][2];
                    project_kv(hd, kk, vs);
                    attend(hd, it, kk, vs, y);
                }
                store_tile(it, y);
            }
        }
    }
}

bool tattn3_supported(int C, int F, int heads) { return (C == 64 || C == 128) && F <= 64 && heads == 4; }
size_t tattn3_workspace_bytes(int C, long long rows) { return C == 128 ? (size_t)rows * C * sizeof(float) : 0; }

template <int C, int PASS>
static void launch_t3(const TattnParams& p, const unsigned char* wq3, const unsigned char* wo3, float* part, long long grid,
                      hipStream_t s) {
    constexpr int LDS = T3<C>::LDS_BYTES;
    static DeviceOnce once;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)tattn3_kernel<C, true, PASS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)tattn3_kernel<C, false, PASS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        once = true;
    }
    if (p.F == 32) hipLaunchKernelGGL((tattn3_kernel<C, true, PASS>), dim3((unsigned)grid), dim3(512), LDS, s, p, wq3, wo3, part);
    else hipLaunchKernelGGL((tattn3_kernel<C, false, PASS>), dim3((unsigned)grid), dim3(512), LDS, s, p, wq3, wo3, part);
}

template <int C, int PASS>
static void launch_t3w(const TattnParams& p, const unsigned char* wq3, const unsigned char* wo3, float* part, long long grid,
                       hipStream_t s) {
    constexpr int LDS = T3W<C>::LDS_BYTES;
    static DeviceOnce once;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)tattn3w_kernel<C, PASS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        once = true;
    }
    hipLaunchKernelGGL((tattn3w_kernel<C, PASS>), dim3((unsigned)grid), dim3(256), LDS, s, p, wq3, wo3, part);
}

int launch_tattn3(const TattnParams& p_in, const unsigned char* wq3, const unsigned char* wo3, int C, void* workspace,
                  hipStream_t s) {
    TattnParams p = p_in;
    p.oflag = overflow_flag_current();
    DPC_REQUIRE(tattn3_supported(C, p.F, 4), "tattn3: unsupported shape");
    DPC_REQUIRE(p.F > 32 ? p.brel != nullptr : p.bias32 != nullptr, "tattn3: bias table (padded / Toeplitz form) missing");
    DPC_REQUIRE(C == 64 || workspace, "tattn3: the C = 128 form needs its partial-sum workspace");
    if (p.npix == 0) return DPC_OK;
    DPC_REQUIRE(p.npix < (1ll << 31), "tattn3: too many sequences for one launch");
    const double rows = (double)p.npix * p.F;
    ProfScope prof(PROF_TATTN_FUSED, 2.0 * rows * C * 384 + 4.0 * rows * p.F * 32 * 4 + 2.0 * rows * 128 * C,
                   4.0 * rows * C * (C == 64 ? 2 : 5), s);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        DPC_HIP(hipGetDevice(&dev));
        DPC_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount;
    }
    if (p.F > 32) {
        const long long gridw = std::min<long long>((p.npix + 3) / 4, cu_budget(ncu));
        if (C == 64) {
            launch_t3w<64, 0>(p, wq3, wo3, nullptr, gridw, s);
        } else {
            launch_t3w<128, 1>(p, wq3, wo3, reinterpret_cast<float*>(workspace), gridw, s);
            DPC_LAUNCH_CHECK();
            launch_t3w<128, 2>(p, wq3, wo3, reinterpret_cast<float*>(workspace), gridw, s);
        }
        DPC_LAUNCH_CHECK();
        return DPC_OK;
    }
    const long long grid = std::min<long long>((p.npix + 7) / 8, cu_budget(ncu));
    if (C == 64) {
        launch_t3<64, 0>(p, wq3, wo3, nullptr, grid, s);
    } else {
        launch_t3<128, 1>(p, wq3, wo3, reinterpret_cast<float*>(workspace), grid, s);
        DPC_LAUNCH_CHECK();
        launch_t3<128, 2>(p, wq3, wo3, reinterpret_cast<float*>(workspace), grid, s);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ---- device self-test of the hardware saturation the fused attention kernels rely on (f16x3.h: hw_sat_enable)
__global__ void fp16_clamp_selftest_kernel(const float* __restrict__ x, float* __restrict__ out, int n) {
    hw_sat_enable();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const unsigned hi = cvt_pk(v, 0.f);
    const unsigned lo = f16_sub_pk(v, 0.f, hi);
    out[2 * i] = (float)__builtin_bit_cast(f16x2, hi).x;
    out[2 * i + 1] = (float)__builtin_bit_cast(f16x2, lo).x;
}

}  // namespace dpc

extern "C" int dpc_selftest_fp16_clamp(const float* x, float* out, int n, dpc_stream_t stream) {
    using namespace dpc;
    DPC_REQUIRE(x && out && n >= 0, "selftest_fp16_clamp: bad argument");
    if (n == 0) return DPC_OK;
    hipLaunchKernelGGL(fp16_clamp_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, out, n);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}
