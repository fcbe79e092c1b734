// Fused temporal attention block on the bf16 matrix cores with fp32 semantics (bf16x6, see bf16x6.h):
//   y = x + to_out( softmax( (q*s) k^T + bias ) v ),  q,k,v = to_qkv(LayerNorm_c(x)),  rotary on q,k
// Reference: video_diffusion_pytorch_conv3d.py:165-184 (LayerNorm/PreNorm), :276-352 (Attention), :382,394,442.
//
// Same dataflow as tattn_fused.hip (one wave = one pixel = one 32-token sequence, every intermediate produced in the
// register layout the next MFMA consumes, qkv never in HBM), but every GEMM is 6 v_mfma_f32_32x32x16_bf16 per 16-deep
// k-step instead of 8 v_mfma_f32_32x32x2_f32: 2.67x fewer matrix-core cycles.  Operands that come out of an
// accumulator (q, k, v, P, O) are split into three bf16 planes in registers; the contraction index of such an operand
// follows the accumulator's register order (bf16x6.h: split_acc), which the pre-split to_out weights are packed for.
// Weights are pre-split once at load time into per-head images that are copied verbatim into LDS.
#include "bf16x6.h"

namespace dpc {

using namespace b6;

size_t attn6_qkv_bytes(int C) { return (size_t)4 * 3 * (C / 16) * 3 * 1024; }
size_t attn6_out_bytes(int C) { return (size_t)4 * (C / 32) * 2 * 3 * 1024; }

// to_qkv.weight [384][C] -> [head][q|k|v][C/16][3][32 rows][2][8];  to_out.weight [C][128] -> [head][C/32][2][3][32][2][8]
__global__ void pack_attn6_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int C, int is_out,
                                  long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one (value) = 3 planes
    if (idx >= total) return;
    const int i = (int)(idx & 7), hh = (int)((idx >> 3) & 1), r = (int)((idx >> 4) & 31);
    long long rest = idx >> 9;
    float v;
    long long o;
    if (!is_out) {
        const int KS = C / 16;
        const int ks = (int)(rest % KS); rest /= KS;
        const int part = (int)(rest % 3), hd = (int)(rest / 3);
        v = w[(long long)(part * 128 + hd * 32 + r) * C + 16 * ks + 8 * hh + i];
        o = ((((long long)(hd * 3 + part) * KS + ks) * 3) * 32 + r) * 16 + hh * 8 + i;
    } else {
        const int NTC = C / 32;
        const int s = (int)(rest & 1); rest >>= 1;
        const int nt = (int)(rest % NTC), hd = (int)(rest / NTC);
        const int d = 16 * s + 4 * hh + (i & 3) + 8 * (i >> 2);
        v = w[(long long)(nt * 32 + r) * 128 + hd * 32 + d];
        o = ((((long long)(hd * NTC + nt) * 2 + s) * 3) * 32 + r) * 16 + hh * 8 + i;
    }
    const unsigned p1 = cvt_pk(v, 0.f) & 0xffffu;
    const float r1 = v - lo_f32(p1);
    const unsigned p2 = cvt_pk(r1, 0.f) & 0xffffu;
    const unsigned p3 = cvt_pk(r1 - lo_f32(p2), 0.f) & 0xffffu;
    dst[o] = (unsigned short)p1;
    dst[o + 512] = (unsigned short)p2;
    dst[o + 1024] = (unsigned short)p3;
}

int launch_pack_attn6(const float* w, unsigned char* dst, int C, bool is_out, hipStream_t s) {
    const long long total = is_out ? (long long)4 * (C / 32) * 2 * 512 : (long long)4 * 3 * (C / 16) * 512;
    hipLaunchKernelGGL(pack_attn6_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w,
                       reinterpret_cast<unsigned short*>(dst), C, is_out ? 1 : 0, total);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

__device__ __forceinline__ int rowmap6(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int C, bool FULL>     // FULL: F == 32, no token masks
__global__ __launch_bounds__(256, (C == 64 ? 2 : 1)) void tattn6_kernel(TattnParams p, const unsigned char* __restrict__ wq6,
                                                                       const unsigned char* __restrict__ wo6) {
    constexpr int KS = C / 16;             // k-steps of the projections
    constexpr int NTC = C / 32;            // output-channel tiles
    constexpr int PART = KS * 3 * 1024;    // bytes of one of q | k | v for one head
    constexpr int QKV_BYTES = 3 * PART, OUT_BYTES = NTC * 2 * 3 * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem6[];
    unsigned char* Ws = smem6;
    unsigned char* Wo = smem6 + QKV_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int loff = l31 * 32 + hh * 16;
    const long long gp = (long long)blockIdx.x * 4 + wave;          // global pixel
    const bool active = gp < p.npix;
    const long long HW = p.HW;
    const unsigned bq = active ? (unsigned)gp / (unsigned)HW : 0u;  // npix < 2^31 (checked by the launcher)
    const long long b = bq, pix = active ? (long long)((unsigned)gp - bq * (unsigned)HW) : 0;
    const long long row0 = b * p.F * HW + pix;                        // row of token 0; token f at row0 + f*HW
    const int F = p.F;
    const float scale = 0.17677669529663687f;

    // ---- LayerNorm'ed x rows, split: lane (token l31, half hh) holds channels 16ks + 8hh .. +7 of k-step ks
    bf16x8 xs[KS][3];
    const bool tok_ok = active && (FULL || l31 < F);
    {
        f32x4 xa[KS][2];
        const float* src = p.x + (row0 + (long long)(tok_ok ? l31 : 0) * HW) * C + 8 * hh;
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4 * q);      // clamped address: always valid
                if (!tok_ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
                xa[ks][q] = v;
                s += (v.x + v.y) + (v.z + v.w);
            }
        s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)C;
        float q2 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 d = xa[ks][q] - mean;
                q2 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        q2 += __shfl_xor(q2, 32, 64);
        const float inv = 1.0f / sqrtf(q2 / (float)C + 1e-5f);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f32x4 n[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + 16 * ks + 8 * hh + 4 * q);
                n[q] = (xa[ks][q] - mean) * inv * g;       // a masked token has x = mean = 0: stays exactly 0
            }
            split8(n[0].x, n[0].y, n[0].z, n[0].w, n[1].x, n[1].y, n[1].z, n[1].w, xs[ks]);
        }
    }

    f32x16 y[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[nt][r] = 0.f;

    for (int hd = 0; hd < 4; ++hd) {
        // ---- stage this head's pre-split weight images (block-cooperative verbatim copy)
        {
            constexpr int NQ = QKV_BYTES / 4096, NO = OUT_BYTES / 4096;       // 16-byte pieces per thread
            const uint4* sq = reinterpret_cast<const uint4*>(wq6 + (size_t)hd * QKV_BYTES) + tid;
            const uint4* so = reinterpret_cast<const uint4*>(wo6 + (size_t)hd * OUT_BYTES) + tid;
            uint4 tq[NQ], to[NO];
#pragma unroll
            for (int q = 0; q < NQ; ++q) tq[q] = sq[q * 256];                 // all loads in flight before the barrier
#pragma unroll
            for (int q = 0; q < NO; ++q) to[q] = so[q * 256];
            __syncthreads();                                                  // previous head's LDS reads are done
#pragma unroll
            for (int q = 0; q < NQ; ++q) reinterpret_cast<uint4*>(Ws)[tid + q * 256] = tq[q];
#pragma unroll
            for (int q = 0; q < NO; ++q) reinterpret_cast<uint4*>(Wo)[tid + q * 256] = to[q];
        }
        __syncthreads();

        // ---- projections: Q^T, K^T (A = weights: lane = token, regs = head dims), V (A = x: lane = d, regs = token)
        f32x16 qT, kT, vv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { qT[r] = 0.f; kT[r] = 0.f; vv[r] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 w[3];
            load_w3(Ws, ks, loff, w);
            mfma6(qT, w, xs[ks]);
            load_w3(Ws + PART, ks, loff, w);
            mfma6(kT, w, xs[ks]);
            load_w3(Ws + 2 * PART, ks, loff, w);
            mfma6(vv, xs[ks], w);
        }
        // ---- scale + rotary (pairs (2m, 2m+1) are registers (4jj+0,4jj+1), (4jj+2,4jj+3) of this lane)
        {
            const int ti = (FULL || l31 < F) ? l31 : 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.rot_cos + ti * 32 + 8 * jj + 4 * hh);
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.rot_sin + ti * 32 + 8 * jj + 4 * hh);
                const float q0 = qT[4 * jj] * scale, q1 = qT[4 * jj + 1] * scale, q2 = qT[4 * jj + 2] * scale,
                            q3 = qT[4 * jj + 3] * scale;
                qT[4 * jj] = __fadd_rn(__fmul_rn(q0, c4.x), __fmul_rn(-q1, s4.x));
                qT[4 * jj + 1] = __fadd_rn(__fmul_rn(q1, c4.y), __fmul_rn(q0, s4.y));
                qT[4 * jj + 2] = __fadd_rn(__fmul_rn(q2, c4.z), __fmul_rn(-q3, s4.z));
                qT[4 * jj + 3] = __fadd_rn(__fmul_rn(q3, c4.w), __fmul_rn(q2, s4.w));
                const float k0 = kT[4 * jj], k1 = kT[4 * jj + 1], k2 = kT[4 * jj + 2], k3 = kT[4 * jj + 3];
                kT[4 * jj] = __fadd_rn(__fmul_rn(k0, c4.x), __fmul_rn(-k1, s4.x));
                kT[4 * jj + 1] = __fadd_rn(__fmul_rn(k1, c4.y), __fmul_rn(k0, s4.y));
                kT[4 * jj + 2] = __fadd_rn(__fmul_rn(k2, c4.z), __fmul_rn(-k3, s4.z));
                kT[4 * jj + 3] = __fadd_rn(__fmul_rn(k3, c4.w), __fmul_rn(k2, s4.w));
            }
        }
        // ---- S^T[j][i] = k_j . q_i   (A = K: lane = key j; B = Q: lane = query i; k index = head dim in register order)
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        {
            bf16x8 qs[2][3], kk[2][3];
            split_acc(qT, qs);
            split_acc(kT, kk);
            mfma6(st, kk[0], qs[0]);
            mfma6(st, kk[1], qs[1]);
        }
        // ---- bias + softmax over keys (lane-local + partner lane)
        float m = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {       // registers 4jj .. 4jj+3 = keys 8jj + 4hh .. +3: one 16-byte load of the padded table
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias32 + (hd * 32 + l31) * 32 + 8 * jj + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sv = st[4 * jj + e] + b4[e];
                if (!FULL && 8 * jj + 4 * hh + e >= F) sv = -INFINITY;
                st[4 * jj + e] = sv;
                m = fmaxf(m, sv);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(st[r] - m);
            st[r] = e;
            l += e;
        }
        l += __shfl_xor(l, 32, 64);
        const float il = 1.0f / l;
        // ---- O^T[d][i] = sum_j V[j][d] P[i][j]   (A = V: lane = d; B = P: lane = query i; k index = key in register order)
        f32x16 oT;
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[r] = 0.f;
        {
            bf16x8 vs[2][3], ps[2][3];
            split_acc(vv, vs);
            split_acc(st, ps);
            mfma6(oT, vs[0], ps[0]);
            mfma6(oT, vs[1], ps[1]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[r] *= il;
        // ---- Y[i][c] += sum_d O[i][d] Wout[c][hd*32+d]   (A = O: lane = token i; B = packed to_out slice)
        {
            bf16x8 os[2][3];
            split_acc(oT, os);
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 w[3];
                    load_w3(Wo, nt * 2 + s, loff, w);
                    mfma6(y[nt], os[s], w);
                }
        }
    }
    // ---- residual + store (lane = channel, regs = token)
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        float res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {           // residual loads first, unconditional (clamped token), then the stores
            const int i = rowmap6(r, hh);
            const int ic = (FULL || i < F) ? i : 0;
            res[r] = p.x[(row0 + (long long)ic * HW) * C + nt * 32 + l31];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = rowmap6(r, hh);
            if (active && (FULL || i < F)) p.out[(row0 + (long long)i * HW) * C + nt * 32 + l31] = y[nt][r] + res[r];
        }
    }
}

int launch_tattn6(const TattnParams& p, const unsigned char* wq6, const unsigned char* wo6, int C, hipStream_t s) {
    DPC_REQUIRE(tattn_fused_supported(C, p.F, 4), "tattn6: unsupported shape");
    if (p.npix == 0) return DPC_OK;
    const long long grid = (p.npix + 3) / 4;
    DPC_REQUIRE(p.npix < (1ll << 31), "tattn6: too many sequences for one launch");
    const double rows = (double)p.npix * p.F;
    ProfScope prof(PROF_TATTN_FUSED, 2.0 * rows * C * 384 + 4.0 * rows * p.F * 32 * 4 + 2.0 * rows * 128 * C,
                   4.0 * rows * C * 2, s);
    const size_t lds = (size_t)(3 * (C / 16) * 3 + (C / 32) * 2 * 3) * 1024;
    DPC_REQUIRE(p.bias32, "tattn6: padded bias table missing");
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)tattn6_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        DPC_HIP(hipFuncSetAttribute((const void*)tattn6_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 49152));
        DPC_HIP(hipFuncSetAttribute((const void*)tattn6_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        DPC_HIP(hipFuncSetAttribute((const void*)tattn6_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
        once = true;
    }
    const dim3 g((unsigned)grid), blk(256);
    if (C == 64) {
        if (p.F == 32) hipLaunchKernelGGL((tattn6_kernel<64, true>), g, blk, lds, s, p, wq6, wo6);
        else hipLaunchKernelGGL((tattn6_kernel<64, false>), g, blk, lds, s, p, wq6, wo6);
    } else {
        if (p.F == 32) hipLaunchKernelGGL((tattn6_kernel<128, true>), g, blk, lds, s, p, wq6, wo6);
        else hipLaunchKernelGGL((tattn6_kernel<128, false>), g, blk, lds, s, p, wq6, wo6);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
