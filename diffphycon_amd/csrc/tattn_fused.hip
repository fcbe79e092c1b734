// Fused temporal attention block: Residual(PreNorm(Attention over frames)) for one pixel per wavefront.
//   y = x + to_out( softmax( (q*s) k^T + bias ) v ),  q,k,v = to_qkv(LayerNorm_c(x)),  rotary on q,k
// Reference: video_diffusion_pytorch_conv3d.py:165-184 (LayerNorm/PreNorm), :276-352 (Attention), :382,394,442.
//
// The qkv tensor never exists in HBM (unfused it is 384 floats per token = 1.6 GB per top-level module at
// micro-batch 8).  One wave owns one pixel = a 32-token sequence = one 32-row MFMA tile, and every intermediate
// stays in registers in exactly the layout the next MFMA wants, by choosing which operand is "A":
//   Q^T = Wq  . xn^T   (A = weight fragment, B = xn fragment)  -> lane = token i, regs = head dims d   [= operand layout]
//   K^T = Wk  . xn^T                                            -> lane = token j, regs = d             [= operand layout]
//   V   = xn  . Wv^T   (A = xn fragment, B = weight fragment)   -> lane = d, regs = token j            [= operand layout]
//   S^T = K Q^T        (A = K regs, B = Q regs)                 -> lane = query i, regs = keys j: softmax is lane-local
//   O^T = V^T P^T      (A = V regs, B = P regs)                 -> lane = query i, regs = d            [= operand layout]
//   Y  += O Wout_h^T   (A = O regs, B = weight fragment)        -> lane = channel c, regs = token      [coalesced store]
// LDS only stages the current head's weight slices (shared by the 4 waves of a workgroup).
#include "common.h"

namespace dpc {

__device__ __forceinline__ int rowmap_t(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int C>
__global__ __launch_bounds__(256, 2) void tattn_fused_kernel(TattnParams p) {
    constexpr int CJ = C / 8;              // float4 fragments per lane along the channel axis
    constexpr int WST = C + 4;             // LDS row stride of the qkv weight slice
    constexpr int NTC = C / 32;            // output-channel tiles
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                      // [96][WST]  rows: q(32) k(32) v(32) of the current head
    float* Wo = Ws + 96 * WST;             // [C][36]    to_out columns of the current head

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const long long gp = (long long)blockIdx.x * 4 + wave;          // global pixel
    const bool active = gp < p.npix;
    const long long HW = p.HW;
    const long long b = active ? gp / HW : 0, pix = active ? gp % HW : 0;
    const long long row0 = b * p.F * HW + pix;                        // row of token 0; token f at row0 + f*HW
    const int F = p.F;
    const float scale = 0.17677669529663687f;

    // ---- x rows as MFMA fragments: lane (token l31, half hh) holds x[token][8j+4hh .. +3]
    f32x4 xa[CJ];
    const bool tok_ok = active && l31 < F;
    {
        const float* src = p.x + (row0 + (long long)(tok_ok ? l31 : 0) * HW) * C + 4 * hh;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (tok_ok) v = *reinterpret_cast<const f32x4*>(src + 8 * j);
            xa[j] = v;
            s += (v.x + v.y) + (v.z + v.w);
        }
        s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const f32x4 d = xa[j] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
        q += __shfl_xor(q, 32, 64);
        const float inv = 1.0f / sqrtf(q / (float)C + 1e-5f);
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + 8 * j + 4 * hh);
            xa[j] = tok_ok ? (xa[j] - mean) * inv * g : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    f32x16 y[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[nt][r] = 0.f;

    for (int hd = 0; hd < 4; ++hd) {
        // ---- stage this head's weight slices (block-cooperative)
        __syncthreads();
        for (int q4 = tid; q4 < 96 * (C / 4); q4 += 256) {
            const int rrow = q4 / (C / 4), c4 = q4 % (C / 4);
            const int grow = (rrow >> 5) * 128 + hd * 32 + (rrow & 31);               // q | k | v block of to_qkv.weight
            *reinterpret_cast<f32x4*>(&Ws[rrow * WST + c4 * 4]) =
                *reinterpret_cast<const f32x4*>(p.wqkv + (long long)grow * C + c4 * 4);
        }
        for (int q4 = tid; q4 < C * 8; q4 += 256) {
            const int c = q4 >> 3, d4 = q4 & 7;
            *reinterpret_cast<f32x4*>(&Wo[c * 36 + d4 * 4]) =
                *reinterpret_cast<const f32x4*>(p.wout + (long long)c * 128 + hd * 32 + d4 * 4);
        }
        __syncthreads();

        // ---- projections
        f32x16 qT, kT, vv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { qT[r] = 0.f; kT[r] = 0.f; vv[r] = 0.f; }
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const f32x4 wq = *reinterpret_cast<const f32x4*>(&Ws[(l31)*WST + 8 * j + 4 * hh]);
            const f32x4 wk = *reinterpret_cast<const f32x4*>(&Ws[(32 + l31) * WST + 8 * j + 4 * hh]);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&Ws[(64 + l31) * WST + 8 * j + 4 * hh]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qT = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s], xa[j][s], qT, 0, 0, 0);
                kT = __builtin_amdgcn_mfma_f32_32x32x2f32(wk[s], xa[j][s], kT, 0, 0, 0);
                vv = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j][s], wv[s], vv, 0, 0, 0);
            }
        }
        // ---- scale + rotary (pairs (2m, 2m+1) are registers (4jj+0,4jj+1), (4jj+2,4jj+3) of this lane)
        {
            const int ti = l31 < F ? l31 : 0;
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
        // ---- S^T[j][i] = k_j . q_i   (A = K regs: lane = key j; B = Q regs: lane = query i)
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kT[r], qT[r], st, 0, 0, 0);
        // ---- bias + softmax over keys (lane-local + partner lane)
        float m = -INFINITY;
        const int qi = l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = rowmap_t(r, hh);
            float sv = st[r];
            if (qi < F && j < F) sv += p.bias[((long long)hd * F + qi) * F + j];
            if (j >= F) sv = -INFINITY;
            st[r] = sv;
            m = fmaxf(m, sv);
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
        // ---- O^T[d][i] = sum_j V[j][d] P[i][j]   (A = V regs: lane = d; B = P regs: lane = query i)
        f32x16 oT;
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r], st[r], oT, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[r] *= il;
        // ---- Y[i][c] += sum_d O[i][d] Wout[c][hd*32+d]   (A = O regs: lane = token i)
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&Wo[(nt * 32 + l31) * 36 + 8 * jj + 4 * hh]);
#pragma unroll
                for (int s = 0; s < 4; ++s) y[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(oT[4 * jj + s], w[s], y[nt], 0, 0, 0);
            }
        }
    }
    // ---- residual + store (lane = channel, regs = token)
    if (active) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = rowmap_t(r, hh);
                if (i < F) {
                    const long long o = (row0 + (long long)i * HW) * C + nt * 32 + l31;
                    p.out[o] = y[nt][r] + p.x[o];
                }
            }
        }
    }
}

bool tattn_fused_supported(int C, int F, int heads) { return (C == 64 || C == 128) && F <= 32 && heads == 4; }

int launch_tattn_fused(const TattnParams& p, int C, hipStream_t s) {
    DPC_REQUIRE(tattn_fused_supported(C, p.F, 4), "tattn_fused: unsupported shape");
    if (p.npix == 0) return DPC_OK;
    const long long grid = (p.npix + 3) / 4;
    DPC_REQUIRE(grid < (1ll << 31), "tattn_fused: grid too large");
    const double rows = (double)p.npix * p.F;
    ProfScope prof(PROF_TATTN_FUSED, 2.0 * rows * C * 384 + 4.0 * rows * p.F * 32 * 4 + 2.0 * rows * 128 * C,
                   4.0 * rows * C * 2, s);
    const size_t lds = (96 * (C + 4) + C * 36) * sizeof(float);
    if (C == 64) {
        hipLaunchKernelGGL(tattn_fused_kernel<64>, dim3((unsigned)grid), dim3(256), lds, s, p);
    } else {
        static DeviceOnce once;
        if (!once) { DPC_HIP(hipFuncSetAttribute((const void*)tattn_fused_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
        hipLaunchKernelGGL(tattn_fused_kernel<128>, dim3((unsigned)grid), dim3(256), lds, s, p);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
