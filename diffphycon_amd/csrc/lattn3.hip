// Fused spatial linear attention block for C = 64, one launch, weight-stationary, f16x3 arithmetic (conv3f3.hip / f16x3.h):
//   q,k,v = to_qkv(LayerNorm_c(x));  q = softmax_d(q) * s;  k = softmax_n(k);  ctx = k v^T;  out = ctx^T q;
//   y = x + to_out(out) + b
// Reference: video_diffusion_pytorch_conv3d.py:232-257 (SpatialLinearAttention), :441 (Residual(PreNorm(...))).
// OUT_LN (r04): the 2-D U-Net's LinearAttention (model/burgers_1d/unet.py:188-229) is the same block with a channel LayerNorm behind
// to_out (to_out = Sequential(Conv2d, LayerNorm), :196-199):  y = x + LayerNorm_c(to_out(out) + b) * g_out.  The token's C outputs
// sit in the 32 lanes of a half-wave x NTC accumulator tiles, so the two reductions (mean, then the variance of the deviations --
// the two-pass form of torch.var) are DPP adds inside the half-wave; nothing else changes.
//
// One 512-thread workgroup per frame image (N = H*W tokens).  All pre-split weights stay in LDS for the whole kernel:
//   region A (64 KB): Wk | Wv of the four heads  -> after phase 1 reused as the cross-wave reduction scratch
//   region B (88 KB): Wq (32 KB) | Wout (32 KB) | the image's four 32x32 contexts as split A-operand planes (16 KB) | m, z (8 KB)
// Phase 1: every wave streams 32-token tiles (rows prefetched one tile ahead): LayerNorm in registers, K and V of all
//          four heads on the MFMA, ONLINE softmax over tokens per head-dim column (lane-local), ctx^T += V^T exp(K).
//          The 8 partial contexts are merged through region A, normalised, and written as fp16 operand planes.
// Phase 2: the same tiles again (L2-resident now): Q^T = Wq xn^T, softmax over head dims (lane-local), out^T = ctx^T q,
//          y += out Wout_h^T, + bias + residual.  No barrier inside either tile loop; the qkv tensor never exists in HBM.
// Operand orientation as in tattn3.hip: every intermediate is produced in the register layout the next MFMA consumes.
#include "f16x3.h"

namespace dpc {

using namespace h3;

// C = 64: phase-1 weights (64 KB), phase-2 weights (64 KB), contexts and merge state are resident together.
// C = 128: both weight sets are 128 KB, so they SHARE region A: Wk | Wv during phase 1, then (after the merge, which also uses
// region A as scratch) the workgroup re-stages Wq | Wout into it for phase 2; contexts and merge state sit behind it.
template <int C_>
struct L3 {
    static constexpr int C = C_, KS = C / 16, NTC = C / 32;
    static constexpr int HEAD_QKV = 3 * KS * 2048;            // global image: [head][q|k|v][ks][plane][1 KB]   (pack_tattn3 layout)
    static constexpr int HEAD_OUT = NTC * 2 * 2048;           // global image: [head][nt][s][plane][1 KB]
    static constexpr int A_BYTES = 4 * 2 * KS * 2048;         // [head][k|v][ks][plane][1 KB]: 64 / 128 KB
    static constexpr bool SHARED = C > 64;                    // phase-2 weights re-staged into region A
    static constexpr int OFF_WQ = SHARED ? 0 : A_BYTES, OFF_WO = OFF_WQ + 4 * KS * 2048;
    static constexpr int OFF_CTX = SHARED ? A_BYTES : OFF_WO + 4 * HEAD_OUT;
    static constexpr int OFF_MZ = OFF_CTX + 4 * 2 * 2048;     // [head][m|z][wave 8][32] floats
    static constexpr int LDS_BYTES = OFF_MZ + 4 * 2 * 8 * 32 * 4;         // 155648 (C = 64 and C = 128)
    static_assert(!SHARED || 4 * KS * 2048 + 4 * HEAD_OUT <= A_BYTES, "phase-2 weights must fit region A");
};
namespace l3 {
constexpr float SX = 16.f, SWGT = 4096.f, PROJ_DESCALE = 1.f / (SX * SWGT);
constexpr float SP = 1024.f, SV = 16.f, SC = 16.f, SQ = 4096.f, SO = 16.f;
constexpr float LOG2E = 1.4426950408889634f, LOG2_SP = 10.f;          // SP = 2^10
}  // namespace l3

__device__ __forceinline__ int rowmap_l3(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// all-reduce over the 32 lanes that share lane >> 5 (fixed order: quad, 8, 16 by DPP, the two 16-lane rows by one ds_bpermute)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum_half_wave(float v) {
    v += dpp_f<0xB1>(v);               // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);               // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);              // row_half_mirror: the other quad of the 8
    v += dpp_f<0x140>(v);              // row_mirror: the other 8 of the row
    v += __shfl_xor(v, 16, 64);
    return v;
}

template <int C_, bool OUT_LN>
__global__ __launch_bounds__(512, 1) void lattn3_kernel(LattnParams p, const unsigned char* __restrict__ wq3,
                                                        const unsigned char* __restrict__ wo3) {
    using namespace l3;
    using G = L3<C_>;
    constexpr int C = G::C, KS = G::KS, NTC = G::NTC, HEAD_QKV = G::HEAD_QKV, HEAD_OUT = G::HEAD_OUT, A_BYTES = G::A_BYTES;
    constexpr int OFF_WQ = G::OFF_WQ, OFF_WO = G::OFF_WO, OFF_CTX = G::OFF_CTX, OFF_MZ = G::OFF_MZ;
    constexpr bool SHARED = G::SHARED, PREFETCH = C == 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int loff = l31 * 32 + hh * 16;
    const long long img = blockIdx.x;
    const int N = p.N;
    const int ntiles = (N + 31) / 32;
    hw_sat_enable();                                           // (f16x3.h: operand conversions saturate in hardware)

    // ---- one-time fill of the weight regions
    for (int q = tid; q < A_BYTES / 16; q += 512) {            // k | v parts of each head: 16 KB per head
        const int hd = q / (2 * KS * 128), r = q % (2 * KS * 128);
        reinterpret_cast<uint4*>(sm)[q] = reinterpret_cast<const uint4*>(wq3 + (size_t)hd * HEAD_QKV + KS * 2048)[r];
    }
    auto fill_phase2 = [&]() {
        for (int q = tid; q < 4 * KS * 128; q += 512) {        // q part of each head
            const int hd = q / (KS * 128), r = q % (KS * 128);
            reinterpret_cast<uint4*>(sm + OFF_WQ)[q] = reinterpret_cast<const uint4*>(wq3 + (size_t)hd * HEAD_QKV)[r];
        }
        for (int q = tid; q < 4 * HEAD_OUT / 16; q += 512)
            reinterpret_cast<uint4*>(sm + OFF_WO)[q] = reinterpret_cast<const uint4*>(wo3)[q];
    };
    if (!SHARED) fill_phase2();
    __syncthreads();

    // rows of a tile: lane (token l31, half hh) holds channels 16ks + 8hh .. +7 of k-step ks
    f32x4 xr[KS][2];
    auto load_rows = [&](int t) {
        const int n = t * 32 + l31;
        const float* src = p.x + (img * N + (n < N ? n : 0)) * C + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < 2; ++q) xr[ks][q] = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4 * q);
    };
    // LayerNorm over channels (lane pair) of the prefetched rows, pre-scale, split
    auto ln_split = [&](int t, f16x8 (&xs)[KS][2]) {
        const bool ok = t * 32 + l31 < N;
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!ok) xr[ks][q] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                n[q] = (xr[ks][q] - mean) * inv * g;               // a masked token has x = mean = 0: stays exactly 0
            }
            split8(sat16h(n[0].x), sat16h(n[0].y), sat16h(n[0].z), sat16h(n[0].w), sat16h(n[1].x), sat16h(n[1].y), sat16h(n[1].z),
                   sat16h(n[1].w), xs[ks]);
        }
    };

    // ================= phase 1: contexts =================
    {
        f32x16 ctxT[4];
        float m[4], z[4];
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            m[hd] = -INFINITY;
            z[hd] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ctxT[hd][r] = 0.f;
        }
        if (PREFETCH && wave < ntiles) load_rows(wave);
        // MFMA B operands that come from LDS (k / v weights) or die with their product (exp k) keep their registers until two more
        // 3-MFMA groups were issued (common.h: mfma_keep / mfma_order_point); the last pair of a head is carried into
        // the first two groups of the next head (and tile)
        f16x8 es_carry[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) es_carry[i][pl] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = wave; t < ntiles; t += 8) {
            f16x8 xs[KS][2];
            if (!PREFETCH) load_rows(t);
            ln_split(t, xs);
            if (PREFETCH && t + 8 < ntiles) load_rows(t + 8);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int hd = 0; hd < 4; ++hd) {
                const unsigned char* Wk = sm + hd * (2 * KS * 2048);
                f32x16 kk, vv;
#pragma unroll
                for (int r = 0; r < 16; ++r) { kk[r] = 0.f; vv[r] = 0.f; }
                f16x8 wk[KS][2], wv[KS][2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    load_w2(Wk, ks, loff, wk[ks]);
                    mfma3(kk, xs[ks], wk[ks]);                 // K[token][d]: lane = d, regs = tokens
                    if (ks == 0) mfma_keep(kk, es_carry[0][0], es_carry[0][1]);
                    else mfma_keep(kk, wk[ks - 1][0], wk[ks - 1][1]);
                    mfma_order_point();
                    load_w2(Wk + KS * 2048, ks, loff, wv[ks]);
                    mfma3(vv, xs[ks], wv[ks]);                 // V[token][e]: lane = e, regs = tokens
                    if (ks == 0) mfma_keep(vv, es_carry[1][0], es_carry[1][1]);
                    else mfma_keep(vv, wv[ks - 1][0], wv[ks - 1][1]);
                    mfma_order_point();
                }
                float tm = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // r04: logits in units of log2 (one multiply, as before), probabilities as exp2(k' - (m' - log2 SP)) = e SP: a
                    // subtraction and one v_exp_f32, and the operand split below needs no multiplier (z carries SP as well)
                    float kv = kk[r] * (PROJ_DESCALE * LOG2E);
                    if (t * 32 + rowmap_l3(r, hh) >= N) kv = -INFINITY;
                    kk[r] = kv;
                    tm = fmaxf(tm, kv);
                }
                tm = fmaxf(tm, __shfl_xor(tm, 32, 64)) - LOG2_SP;
                const float m_new = fmaxf(m[hd], tm);
                const float alpha = (m[hd] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m[hd] - m_new);
                float zs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(kk[r] - m_new);
                    kk[r] = e;
                    zs += e;
                }
                z[hd] = z[hd] * alpha + zs;                    // per half-wave partial; halves are added at the end
                m[hd] = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) ctxT[hd][r] *= alpha;
                f16x8 vs[2][2], es[2][2];
                split_acc_h<true>(vv, PROJ_DESCALE * SV, vs);
                split_acc_h<false>(kk, 1.f, es);                // (kk = e SP already)
                mfma3(ctxT[hd], vs[0], es[0]);                 // ctx^T[e][d]: lane = d, regs = e
                mfma_keep(ctxT[hd], wk[KS - 1][0], wk[KS - 1][1]);
                mfma3(ctxT[hd], vs[1], es[1]);
                mfma_keep(ctxT[hd], wv[KS - 1][0], wv[KS - 1][1]);
                mfma_order_point();
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) es_carry[i][pl] = es[i][pl];
            }
        }
        // ---- merge the 8 partial contexts (two heads per round through region A), normalise, write operand planes
        float* mz = reinterpret_cast<float*>(sm + OFF_MZ);
#pragma unroll
        for (int hd = 0; hd < 4; ++hd) {
            const float zz = z[hd] + __shfl_xor(z[hd], 32, 64);
            if (hh == 0) {
                mz[((hd * 2 + 0) * 8 + wave) * 32 + l31] = m[hd];
                mz[((hd * 2 + 1) * 8 + wave) * 32 + l31] = zz;
            }
        }
        float* scr = reinterpret_cast<float*>(sm);             // [2 heads][8 waves][16][64]
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            __syncthreads();                                   // phase-1 weights (round 0) / previous round's scratch are dead
#pragma unroll
            for (int hs = 0; hs < 2; ++hs)
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[((hs * 8 + wave) * 16 + r) * 64 + lane] = ctxT[round * 2 + hs][r];
            __syncthreads();
            if (wave < 2) {
                const int hd = round * 2 + wave;
                float M = -INFINITY;
#pragma unroll
                for (int w = 0; w < 8; ++w) M = fmaxf(M, mz[((hd * 2 + 0) * 8 + w) * 32 + l31]);
                float sc[8], Z = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    const float mw = mz[((hd * 2 + 0) * 8 + w) * 32 + l31];
                    sc[w] = (mw == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mw - M);
                    Z += mz[((hd * 2 + 1) * 8 + w) * 32 + l31] * sc[w];
                }
                // this lane holds ctx[d = l31][e = rowmap(r, hh)]; operand plane position of (row e, k = d):
                const int d = l31, s = d >> 4, dd = d & 15, kh = (dd >> 2) & 1, ki = (dd & 3) + 4 * (dd >> 3);
                unsigned short* dst = reinterpret_cast<unsigned short*>(sm + OFF_CTX + (hd * 2 + s) * 2048) + kh * 8 + ki;
                const float norm = SC / (Z * SV);              // (Z = SP x the sum of the probabilities; the contexts carry SV SP)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = 0.f;
#pragma unroll
                    for (int w = 0; w < 8; ++w) v += scr[((wave * 8 + w) * 16 + r) * 64 + lane] * sc[w];
                    v = sat16(v * norm);
                    const unsigned p1 = cvt_pk(v, 0.f) & 0xffffu;
                    const unsigned p2 = cvt_pk(v - (float)__builtin_bit_cast(f16x2, p1).x, 0.f) & 0xffffu;
                    const int e = rowmap_l3(r, hh);
                    dst[e * 16] = (unsigned short)p1;
                    dst[e * 16 + 512] = (unsigned short)p2;
                }
            }
        }
        __syncthreads();
        if (SHARED) {                 // the merge scratch is dead: bring in the phase-2 weights
            fill_phase2();
            __syncthreads();
        }
    }

    // ================= phase 2: outputs =================
    const float qscale = 0.17677669529663687f;
    if (PREFETCH && wave < ntiles) load_rows(wave);
    f16x8 wo_carry[2][2];                                      // (as es_carry: the last two to_out fragments of a head)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wo_carry[i][pl] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = wave; t < ntiles; t += 8) {
        f16x8 xs[KS][2];
        if (!PREFETCH) load_rows(t);
        ln_split(t, xs);
        if (PREFETCH && t + 8 < ntiles) load_rows(t + 8);
        asm volatile("" ::: "memory");
        f32x16 y[NTC];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[nt][r] = 0.f;
#pragma unroll 1
        for (int hd = 0; hd < 4; ++hd) {
            const unsigned char* Wq = sm + OFF_WQ + hd * (KS * 2048);
            const unsigned char* Wo = sm + OFF_WO + hd * HEAD_OUT;
            const unsigned char* Cx = sm + OFF_CTX + hd * (2 * 2048);
            f32x16 qT;
#pragma unroll
            for (int r = 0; r < 16; ++r) qT[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f16x8 w[2];
                load_w2(Wq, ks, loff, w);
                mfma3(qT, w, xs[ks]);                          // Q^T[d][token]: lane = token, regs = d
                if (ks < 2) mfma_keep(qT, wo_carry[ks][0], wo_carry[ks][1]);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                qT[r] *= PROJ_DESCALE * LOG2E;
                mx = fmaxf(mx, qT[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(qT[r] - mx);
                qT[r] = e;
                sum += e;
            }
            sum += __shfl_xor(sum, 32, 64);
            // out^T[e][n] = sum_d ctx[d][e] q[n][d]     (A = context planes: lane = e, k = d; B = q regs: lane = token)
            f32x16 oT;
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[r] = 0.f;
            f16x8 qs[2][2];
            {
                split_acc_h<false>(qT, (qscale * SQ) / sum, qs);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    f16x8 cf[2];
                    load_w2(Cx, s, loff, cf);
                    mfma3(oT, cf, qs[s]);
                }
            }
            {
                f16x8 os[2][2];
                split_acc_h<true>(oT, SO / (SC * SQ), os);
                f16x8 wo[2 * NTC][2];
#pragma unroll
                for (int g = 0; g < 2 * NTC; ++g) {             // group g = (column tile g / 2, k-step g % 2)
                    load_w2(Wo, g, loff, wo[g]);
                    mfma3(y[g >> 1], os[g & 1], wo[g]);         // Y[token][c]: lane = channel, regs = tokens
                    if (g < 2) mfma_keep(y[g >> 1], qs[g][0], qs[g][1]);
                    else mfma_keep(y[g >> 1], wo[g - 2][0], wo[g - 2][1]);
                    mfma_order_point();
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) wo_carry[i][pl] = wo[2 * NTC - 2 + i][pl];
            }
        }
        // ---- bias (+ channel LayerNorm) + residual + store (lane = channel, regs = token)
        float ln_mean[16], ln_inv[16];
        if constexpr (OUT_LN) {
            float sm_[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) sm_[r] = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) {
                const float bv = p.bout[nt * 32 + l31];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    y[nt][r] = y[nt][r] * (1.f / (SO * SWGT)) + bv;
                    sm_[r] += y[nt][r];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) ln_mean[r] = sum_half_wave(sm_[r]) / (float)C;
#pragma unroll
            for (int r = 0; r < 16; ++r) sm_[r] = 0.f;
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = y[nt][r] - ln_mean[r];
                    sm_[r] += d * d;
                }
#pragma unroll
            for (int r = 0; r < 16; ++r) ln_inv[r] = 1.0f / sqrtf(sum_half_wave(sm_[r]) / (float)C + 1e-5f);
        }
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const float bv = OUT_LN ? p.gamma_out[nt * 32 + l31] : p.bout[nt * 32 + l31];
            float res[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = t * 32 + rowmap_l3(r, hh);
                res[r] = p.x[(img * N + (nn < N ? nn : 0)) * C + nt * 32 + l31];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = t * 32 + rowmap_l3(r, hh);
                if (nn >= N) continue;
                if constexpr (OUT_LN) p.out[(img * N + nn) * C + nt * 32 + l31] = (y[nt][r] - ln_mean[r]) * ln_inv[r] * bv + res[r];
                else p.out[(img * N + nn) * C + nt * 32 + l31] = (y[nt][r] * (1.f / (SO * SWGT)) + bv) + res[r];
            }
        }
    }
}

bool lattn3_supported(int C, int heads) { return (C == 64 || C == 128) && heads == 4; }

int launch_lattn3(const LattnParams& p, const unsigned char* wq3, const unsigned char* wo3, int C, hipStream_t s) {
    using namespace l3;
    DPC_REQUIRE(lattn3_supported(C, 4), "lattn3: unsupported width");
    if (p.images == 0) return DPC_OK;
    DPC_REQUIRE(p.images < (1ll << 31), "lattn3: grid too large");
    const double rows = (double)p.images * p.N;
    ProfScope prof(PROF_LATTN_FUSED, 2.0 * rows * C * 384 + 4.0 * rows * 32 * 32 * 4 + 2.0 * rows * 128 * C,
                   4.0 * rows * C * 3, s);
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)lattn3_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L3<64>::LDS_BYTES));
        DPC_HIP(hipFuncSetAttribute((const void*)lattn3_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L3<128>::LDS_BYTES));
        DPC_HIP(hipFuncSetAttribute((const void*)lattn3_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L3<64>::LDS_BYTES));
        DPC_HIP(hipFuncSetAttribute((const void*)lattn3_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, L3<128>::LDS_BYTES));
        once = true;
    }
    const dim3 grid((unsigned)p.images), block(512);
    if (p.gamma_out) {                // trailing channel LayerNorm (2-D U-Net)
        if (C == 64) hipLaunchKernelGGL((lattn3_kernel<64, true>), grid, block, L3<64>::LDS_BYTES, s, p, wq3, wo3);
        else hipLaunchKernelGGL((lattn3_kernel<128, true>), grid, block, L3<128>::LDS_BYTES, s, p, wq3, wo3);
    } else {
        if (C == 64) hipLaunchKernelGGL((lattn3_kernel<64, false>), grid, block, L3<64>::LDS_BYTES, s, p, wq3, wo3);
        else hipLaunchKernelGGL((lattn3_kernel<128, false>), grid, block, L3<128>::LDS_BYTES, s, p, wq3, wo3);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
