// Operator-level entry points for the jellyfish guidance surrogates: the two learned 2-D nets that sit INSIDE
// the design gradient -- `Unet` (boundary updater, diffusion_2d_jellyfish.py:276-403) and `ForceUnet` (:406-481), differentiated
// by `force_fn` (inference_2d_jellyfish.py:85-114) -- forward AND input-gradient backward on hand-written HIP kernels, so that
// no torch autograd graph is left in the sampling loop.  The graph itself (which op follows which, what is kept for the
// backward pass) lives in Python (diffphycon_amd/model/surrogates_hip.py); every tensor-sized operation is one of:
//   dpc_conv_pack / dpc_conv_run      any 2-D convolution or its input-gradient (the same implicit-GEMM kernel on weights
//                                     flipped / transposed / sliced by the host at load time), virtual concat, fused channel
//                                     LayerNorm prologue, residual accumulate, channels-first / parity-scatter outputs
//   dpc_gn_stats / dpc_gn_apply / dpc_gn_silu_bwd      GroupNorm (+ scale/shift) + SiLU and its backward (norm.hip)
//   dpc_ln_stats / dpc_ln_apply / dpc_ln_bwd            channel LayerNorm and its backward (norm.hip)
//   dpc_linear_attention_core / _bwd, dpc_attention_core / dpc_attention_bwd      attention cores (attn.hip, here)
//   dpc_upsample2x_cl / dpc_downsum2x_cl, dpc_nchw_to_cl / dpc_cl_to_nchw, dpc_mean_rows / dpc_bcast_rows, dpc_small_linear
// Only [N, C]-sized vectors (time embedding, its MLP gradient) are touched by torch elementwise ops on the host side.
#include <cmath>
#include <memory>

#include "common.h"
#include "unet_common.h"

struct dpc_conv_s {
    dpc::PackedConv pc;
    dpc::Modes modes;
};

namespace dpc {

__device__ __forceinline__ int rowmap_s(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// ------------------------------------------------------------------------------------ linear attention backward
// forward (LinearAttention.forward, diffusion_2d_jellyfish.py:232-251; the 1/(h w) on v is folded into to_out by the host):
//   qs = softmax_d(q) * scale, ks = softmax_n(k), ctx[d][e] = sum_n ks[n][d] v[n][e], out[n][e] = sum_d ctx[d][e] qs[n][d]
// backward, given dout[n][e]:
//   dctx[d][e] = sum_n qs[n][d] dout[n][e];   dqs[n][d] = sum_e ctx[d][e] dout[n][e]
//   dq[n][d] = scale * p[n][d] * (dqs[n][d] - sum_d' dqs[n][d'] p[n][d'])          (p = softmax_d(q))
//   dks[n][d] = sum_e dctx[d][e] v[n][e];  dk[n][d] = ks[n][d] * (dks[n][d] - sum_e dctx[d][e] ctx[d][e])
//   dv[n][e] = sum_d ks[n][d] dctx[d][e]
// tape per (image, head), LAB_WS floats: written by the forward (dpc_linear_attention_fwd_save): ctx [d][e] | kmax[32] | Z[32];
// by pass 1 below: rowdot[32] | ctxT [e][d] | dctx [d][e] | dctxT [e][d]   (the transposed copies make every MFMA A-operand of
// pass 2 one contiguous 256-byte read)
constexpr int LAB_KMAX = 1024, LAB_Z = 1056, LAB_ROWDOT = 1088, LAB_CTXT = 1152, LAB_DCTX = 2176, LAB_DCTXT = 3200, LAB_WS = 4224;

// pass 1: dctx of one (image, head); 4 waves split the tokens, loads run 4 token pairs ahead of the MFMAs
__global__ __launch_bounds__(256) void linattn_bwd_ctx_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                             float* __restrict__ ws, int heads, int N) {
    __shared__ float s_acc[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hh = lane >> 5;
    const long long img = blockIdx.x / heads;
    const int head = blockIdx.x % heads;
    const int ld = 3 * heads * 32, HD = heads * 32;
    const float scale = 0.17677669529663687f;
    const float* qbase = qkv + img * N * (long long)ld + head * 32 + l31;
    const float* dbase = dout + img * N * (long long)HD + head * 32 + l31;
    int per = (N + 3) / 4;
    per = (per + 7) & ~7;
    const int n_begin = wave * per, n_end = min(N, n_begin + per);
    f32x16 dacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
    for (int n0 = n_begin; n0 < n_end; n0 += 8) {
        float qv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + 2 * u + hh;
            const bool ok = n < n_end;
            qv[u] = ok ? qbase[(long long)n * ld] : 0.f;
            dv[u] = ok ? dbase[(long long)n * HD] : 0.f;        // (a zero gradient row contributes nothing)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // qs[n][d] for this half-wave's token: softmax over the 32 lanes of the half
            float qm = qv[u];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) qm = fmaxf(qm, __shfl_xor(qm, o, 64));
            const float qe = expf(qv[u] - qm);
            float qs = qe;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) qs += __shfl_xor(qs, o, 64);
            dacc = __builtin_amdgcn_mfma_f32_32x32x2f32((qe / qs) * scale, dv[u], dacc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[wave][r][lane] = dacc[r];
    __syncthreads();
    if (wave == 0) {
        float* dst = ws + ((long long)img * heads + head) * LAB_WS;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = rowmap_s(r, hh);
            const float tot = (s_acc[0][r][lane] + s_acc[1][r][lane]) + (s_acc[2][r][lane] + s_acc[3][r][lane]);
            const float cv = dst[d * 32 + l31];
            dst[LAB_DCTX + d * 32 + l31] = tot;
            dst[LAB_DCTXT + l31 * 32 + d] = tot;
            dst[LAB_CTXT + l31 * 32 + d] = cv;
            float rd = tot * cv;                                 // rowdot[d] = sum_e dctx[d][e] ctx[d][e]: reduce over the 32 lanes
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) rd += __shfl_xor(rd, o, 64);
            if (l31 == 0) dst[LAB_ROWDOT + d] = rd;
        }
    }
}

// pass 2: one workgroup per (image, 32-token tile, group of 4 heads), one wave per head.  The tile's q | k | v | dout rows are
// staged through LDS with row-contiguous 512-byte reads and kept TRANSPOSED ([feature][token], pitch 33): every MFMA B operand
// and every per-token softmax read is then a conflict-free row read; results replace the q | k | v slots and leave as whole rows.
constexpr int LAB_HP = 32 * 33 + 1;              // floats per (matrix, head) slab: +1 spreads the 4 heads over distinct banks
constexpr int LAB_LDS = 4 * 4 * LAB_HP * 4;      // bytes
__global__ __launch_bounds__(256) void linattn_bwd_tok_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                             const float* __restrict__ ws, float* __restrict__ dqkv, int heads,
                                                             int N, int tiles_per_img) {
    extern __shared__ float s_t[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const long long img = blockIdx.x / tiles_per_img;
    const int tile = blockIdx.x % tiles_per_img;
    const int h0 = blockIdx.y * 4, nh = min(4, heads - h0);
    const int ld = 3 * heads * 32, HD = heads * 32;
    const long long row0 = img * N + (long long)tile * 32;
    const int nrows = min(32, N - tile * 32);
    // ---- this wave's MFMA A operands (context blocks of its head) and the k-softmax statistics: issued before the staging so that
    //      their L2 latency hides behind it (inside the MFMA loops each would cost a full round trip)
    const int whead = h0 + (wave < nh ? wave : 0);
    const float* w = ws + ((long long)img * heads + whead) * LAB_WS;
    float a_ct[16], a_dt[16], a_d[16], km2[16], z2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = 2 * i + hh;
        a_ct[i] = w[LAB_CTXT + d * 32 + l31];
        a_dt[i] = w[LAB_DCTXT + d * 32 + l31];
        a_d[i] = w[LAB_DCTX + d * 32 + l31];
        km2[i] = w[LAB_KMAX + d];
        z2[i] = w[LAB_Z + d];
    }
    // ---- stage: matrix m (q, k, v, dout), 32 rows x (nh * 8) float4 chunks
    {
        const int j = tid & 31, rr = tid >> 5;           // chunk within the row, row within the pass (8 rows per pass)
        const int hd = j >> 3, f0 = (j & 7) * 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float* src = m < 3 ? qkv + m * HD + h0 * 32 : dout + h0 * 32;
            const int rl = m < 3 ? ld : HD;
            f32x4 v[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = rr + 8 * it;
                v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < nrows && hd < nh) v[it] = *reinterpret_cast<const f32x4*>(src + (row0 + r) * rl + 4 * j);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                float* dstp = s_t + (m * 4 + hd) * LAB_HP + f0 * 33 + rr + 8 * it;
                dstp[0] = v[it].x; dstp[33] = v[it].y; dstp[66] = v[it].z; dstp[99] = v[it].w;
            }
        }
    }
    __syncthreads();
    if (wave < nh) {
        const float scale = 0.17677669529663687f;
        float* Q = s_t + (0 * 4 + wave) * LAB_HP;
        float* K = s_t + (1 * 4 + wave) * LAB_HP;
        float* V = s_t + (2 * 4 + wave) * LAB_HP;
        const float* D = s_t + (3 * 4 + wave) * LAB_HP;
        f32x4 km[4], zz[4], rd[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            km[jj] = *reinterpret_cast<const f32x4*>(w + LAB_KMAX + 8 * jj + 4 * hh);
            zz[jj] = *reinterpret_cast<const f32x4*>(w + LAB_Z + 8 * jj + 4 * hh);
            rd[jj] = *reinterpret_cast<const f32x4*>(w + LAB_ROWDOT + 8 * jj + 4 * hh);
        }
        // ---- dq: acc[d][n] = sum_e ctx[d][e] dout[n][e]
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // (B operands come straight from LDS, one register per MFMA: all 16 are read up front and each keeps its register until four
        //  younger MFMAs were issued -- common.h: mfma_keep_a)
        {
            float bD[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) bD[i] = D[(2 * i + hh) * 33 + l31];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ct[i], bD[i], acc, 0, 0, 0);
                if (i >= 4) mfma_keep_a(acc, bD[i - 4]);
            }
            mfma_drain(acc);                                     // (the softmax below consumes acc anyway; its LDS reads stay behind the chain)
        }
        {
            float q[16];
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) { q[r] = Q[rowmap_s(r, hh) * 33 + l31]; m = fmaxf(m, q[r]); }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { q[r] = expf(q[r] - m); sum += q[r]; }
            sum += __shfl_xor(sum, 32, 64);
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { q[r] /= sum; dot += acc[r] * q[r]; }
            dot += __shfl_xor(dot, 32, 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) Q[rowmap_s(r, hh) * 33 + l31] = scale * q[r] * (acc[r] - dot);
        }
        // ---- dk: acc[d][n] = sum_e dctx[d][e] v[n][e];   dv: acc2[e][n] = sum_d dctx[d][e] ks[n][d]
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
        {
            float bV[16], bK[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int d = 2 * i + hh;
                bV[i] = V[d * 33 + l31];
                bK[i] = expf(K[d * 33 + l31] - km2[i]) / z2[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {                       // two chains: a B register is released two iterations (4 MFMAs) later
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_dt[i], bV[i], acc, 0, 0, 0);
                if (i >= 2) mfma_keep_a(acc, bV[i - 2]);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_d[i], bK[i], acc2, 0, 0, 0);
                if (i >= 2) mfma_keep_a(acc2, bK[i - 2]);
                mfma_order_point();
            }
            mfma_drain(acc);
            mfma_drain(acc2);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const int o = (8 * jj + 4 * hh + sx) * 33 + l31;
                K[o] = (expf(K[o] - km[jj][sx]) / zz[jj][sx]) * (acc[4 * jj + sx] - rd[jj][sx]);
                V[o] = acc2[4 * jj + sx];
            }
    }
    __syncthreads();
    {
        const int j = tid & 31, rr = tid >> 5;
        const int hd = j >> 3, f0 = (j & 7) * 4;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = rr + 8 * it;
                if (r < nrows && hd < nh) {
                    const float* sp = s_t + (m * 4 + hd) * LAB_HP + f0 * 33 + r;
                    *reinterpret_cast<f32x4*>(dqkv + (row0 + r) * ld + m * HD + h0 * 32 + 4 * j) = f32x4{sp[0], sp[33], sp[66], sp[99]};
                }
            }
    }
}

// ------------------------------------------------------------------------------------ dense attention backward
// Attention.forward (:266-275): S = (q scale) k^T, P = softmax_j(S), O = P v.  One workgroup per (image, head), L <= 256 tokens,
// q | k | v | dO of the head in LDS.  Pass Q (thread = query i): row max / sum, t_i = sum_j P_ij dP_ij with dP_ij = dO_i . v_j,
// dq_i = scale sum_j dS_ij k_j;  pass K (thread = key j): dk_j = scale sum_i dS_ij q_i, dv_j = sum_i P_ij dO_i,
// dS_ij = P_ij (dP_ij - t_i).
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                      float* __restrict__ dqkv, int heads, int L) {
    extern __shared__ float s_att[];
    float* sq = s_att;                 // [L][33]
    float* sk = sq + L * 33;
    float* sv = sk + L * 33;
    float* sd = sv + L * 33;
    float* sm = sd + L * 33;           // [L] row max
    float* sl = sm + L;                // [L] row sum
    float* st = sl + L;                // [L] t_i
    const long long img = blockIdx.x / heads;
    const int head = blockIdx.x % heads;
    const int ld = 3 * heads * 32, HD = heads * 32;
    const float scale = 0.17677669529663687f;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < L * 32; idx += 256) {
        const int i = idx >> 5, d = idx & 31;
        const float* row = qkv + (img * L + i) * (long long)ld + head * 32 + d;
        sq[i * 33 + d] = row[0] * scale;
        sk[i * 33 + d] = row[HD];
        sv[i * 33 + d] = row[2 * HD];
        sd[i * 33 + d] = dout[(img * L + i) * (long long)HD + head * 32 + d];
    }
    __syncthreads();
    const int i = tid;
    if (i < L) {
        float q[32], dO[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) { q[d] = sq[i * 33 + d]; dO[d] = sd[i * 33 + d]; }
        float m = -INFINITY;
        for (int j = 0; j < L; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s += q[d] * sk[j * 33 + d];
            m = fmaxf(m, s);
        }
        float l = 0.f, t = 0.f;
        for (int j = 0; j < L; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += q[d] * sk[j * 33 + d]; dp += dO[d] * sv[j * 33 + d]; }
            const float e = expf(s - m);
            l += e;
            t += e * dp;
        }
        t /= l;
        sm[i] = m; sl[i] = l; st[i] = t;
        float dq[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) dq[d] = 0.f;
        for (int j = 0; j < L; ++j) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += q[d] * sk[j * 33 + d]; dp += dO[d] * sv[j * 33 + d]; }
            const float ds = (expf(s - m) / l) * (dp - t);
#pragma unroll
            for (int d = 0; d < 32; ++d) dq[d] += ds * sk[j * 33 + d];
        }
        float* o = dqkv + (img * L + i) * (long long)ld + head * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = dq[d] * scale;
    }
    __syncthreads();
    const int j = tid;
    if (j < L) {
        float k[32], v[32], dk[32], dv[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) { k[d] = sk[j * 33 + d]; v[d] = sv[j * 33 + d]; dk[d] = 0.f; dv[d] = 0.f; }
        for (int ii = 0; ii < L; ++ii) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s += sq[ii * 33 + d] * k[d]; dp += sd[ii * 33 + d] * v[d]; }
            const float pij = expf(s - sm[ii]) / sl[ii];
            const float ds = pij * (dp - st[ii]);
#pragma unroll
            for (int d = 0; d < 32; ++d) { dk[d] += ds * sq[ii * 33 + d]; dv[d] += pij * sd[ii * 33 + d]; }
        }
        float* o = dqkv + (img * L + j) * (long long)ld + head * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) { o[HD + d] = dk[d]; o[2 * HD + d] = dv[d]; }      // (sq already carries the scale)
    }
}

// ------------------------------------------------------------------------------------ small streaming ops
// y[n][h][w][c] = sum over the 2 x 2 block of x[n][2h+a][2w+b][c]   (backward of the nearest x2 up-sampling)
__global__ __launch_bounds__(256) void downsum2x_cl_kernel(const float* __restrict__ x, float* __restrict__ y, long long total4,
                                                          int H, int W, int C4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long r = i / C4;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const long long n = r / H;
        const f32x4* src = reinterpret_cast<const f32x4*>(x) + ((n * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c4;
        reinterpret_cast<f32x4*>(y)[i] = (src[0] + src[C4]) + (src[(long long)2 * W * C4] + src[(long long)2 * W * C4 + C4]);
    }
}
// [N][C][HW] -> [N*HW][Cpad] (zero padded channels) and back ([N*HW][Cpad] -> [N][C][HW], first C channels)
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int C,
                                                        int Cpad, long long HW) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long long p = i / Cpad, n = p / HW, hw = p % HW;
        y[i] = c < C ? x[(n * C + c) * HW + hw] : 0.f;
    }
}
__global__ __launch_bounds__(256) void cl_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int C,
                                                        int Cpad, int csrc, int Ctot, int cdst, float mul, long long HW) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long hw = i % HW;
        const long long r = i / HW;
        const int c = (int)(r % C);
        const long long n = r / C;
        y[(n * Ctot + cdst + c) * HW + hw] = x[(n * HW + hw) * Cpad + csrc + c] * mul;
    }
}
// y_cl[n][hw][cdst] = a * x[n][csrc][hw] + b     (one channel of a channels-first tensor into one channel of a channels-last one)
__global__ __launch_bounds__(256) void channel_affine_to_cl_kernel(const float* __restrict__ x, float* __restrict__ y, long long total,
                                                                  int Ctot, int csrc, int Cpad, int cdst, float a, float b, long long HW) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, hw = i % HW;
        y[i * Cpad + cdst] = a * x[(n * Ctot + csrc) * HW + hw] + b;
    }
}
// out[n] = mean_hw x[n][csrc][hw]      one block per image, fp64 accumulation in a fixed order
__global__ __launch_bounds__(256) void channel_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int Ctot, int csrc,
                                                          long long HW) {
    __shared__ double red[256];
    const float* src = x + ((long long)blockIdx.x * Ctot + csrc) * HW;
    double s = 0;
    for (long long i = threadIdx.x; i < HW; i += 256) s += (double)src[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] / (double)HW);
}
// y[n][cdst][hw] = v[n] * mul
__global__ __launch_bounds__(256) void channel_fill_kernel(float* __restrict__ y, const float* __restrict__ v, long long total, int Ctot,
                                                          int cdst, float mul, long long HW) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, hw = i % HW;
        y[(n * Ctot + cdst) * HW + hw] = v[n] * mul;
    }
}
// y[row][wpad + w][c] = x[row][w][c], zero borders: rows of W pixels -> rows of W + 2 wpad pixels
__global__ __launch_bounds__(256) void pad_w_cl_kernel(const float* __restrict__ x, float* __restrict__ y, long long total4, int W, int C4,
                                                      int wpad) {
    const int Wp = W + 2 * wpad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long r = i / C4;
        const int wp = (int)(r % Wp);
        const long long row = r / Wp;
        const int w = wp - wpad;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)w < (unsigned)W) v = reinterpret_cast<const f32x4*>(x)[(row * W + w) * C4 + c4];
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}
// dx[row][w][c] = sum_b x[row][w + taps/2 - b][b * C + c]   (second half of the 7x7 stem's backward-data product, see _Init7)
__global__ __launch_bounds__(256) void fold_w_cl_kernel(const float* __restrict__ x, float* __restrict__ dx, long long total, int W, int C,
                                                       int taps) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long r = i / C;
        const int w = (int)(r % W);
        const long long row = r / W;
        float s = 0.f;
        for (int b = 0; b < taps; ++b) {
            const int ws = w + taps / 2 - b;
            if ((unsigned)ws < (unsigned)W) s += x[((row * W + ws) * taps + b) * C + c];
        }
        dx[i] = s;
    }
}
// out[n][c] = mean_r x[n][r][c]   (ForceUnet head, :478-480)      one block per (n, 64-channel slab)
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x, float* __restrict__ out, long long R, int C) {
    __shared__ double red[4][64];
    const long long n = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    double s = 0;
    if (c < C)
        for (long long r = part; r < R; r += 4) s += (double)x[(n * R + r) * C + c];
    red[part][threadIdx.x & 63] = s;
    __syncthreads();
    if (part == 0 && c < C) out[n * C + c] = (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (double)R);
}
// y[n][r][c] = v[n][c] * mul      (backward of the mean over pixels)
__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ v, float* __restrict__ y, long long total4, long long R,
                                                        int C4, float mul) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long n = i / C4 / R;
        reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(v)[n * C4 + c4] * mul;
    }
}
// y += x
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ x, long long total4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4*>(y)[i] += reinterpret_cast<const f32x4*>(x)[i];
}

// out[0] = max(out[0], max |x|)   (bit pattern of a non-negative float orders like an unsigned integer)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, unsigned* __restrict__ out, long long total4) {
    // maximum over the |bits| as unsigned integers: identical to the float maximum for finite values, and Inf / NaN patterns
    // (>= 0x7f800000) win against every finite one -- a non-finite gradient PROPAGATES into the result instead of being dropped by
    // fmaxf, so the caller's isfinite() guard sees it
    unsigned m = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = max(max(m, max(abs_bits(v.x), abs_bits(v.y))), max(abs_bits(v.z), abs_bits(v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

static inline unsigned grid_for(long long total) { return (unsigned)std::min<long long>((total + 255) / 256, 256 * 16); }

}  // namespace dpc

using namespace dpc;

extern "C" {

int dpc_conv_pack(const float* w, int N, int K, int kh, int kw, int sh, int sw, int ph, int pw, int tap_begin, int tap_end,
                  const char* mode, dpc_conv_t* out, dpc_stream_t stream) {
    DPC_REQUIRE(w && out && N >= 1 && K >= 1 && kh >= 1 && kw >= 1, "conv_pack: bad argument");
    DPC_REQUIRE(K % 4 == 0, "conv_pack: input channels must be a multiple of 4 (pad on the host)");
    if (tap_end <= 0) tap_end = kh * kw;
    const int ntaps = tap_end - tap_begin;
    DPC_REQUIRE(tap_begin >= 0 && ntaps >= 1 && ntaps <= 32 && tap_end <= kh * kw, "conv_pack: 1..32 taps per pack (split larger kernels)");
    hipStream_t s = (hipStream_t)stream;
    auto h = std::make_unique<dpc_conv_s>();
    h->modes = modes_global();
    if (mode && mode[0]) {
        const std::string m(mode);
        const int v = m == "f32" ? 0 : (m == "x6" ? 1 : (m == "f16x3" ? 2 : -1));
        DPC_REQUIRE(v >= 0, "conv_pack: unknown arithmetic mode '" + m + "'");
        h->modes.igemm = v;
        h->modes.conv = v;
    }
    ModeScope scope(h->modes);
    PackedConv& pc = h->pc;
    if (tap_begin == 0 && ntaps == kh * kw) {
        // whole window: the U-Nets' own packer (3 x 3 stride-1 convs in f16x3 mode also get the big-tile halo pack, conv3f3.hip)
        if (int rc = pack_conv3d(pc, w, N, K, 1, kh, kw, sh, sw, 0, ph, pw, s)) return rc;
        if (igemm_mode_default() == 2)
            if (int rc = f16x3_weight_overflow_check("conv_pack")) return rc;
        *out = h.release();
        return DPC_OK;
    }
    pc.N = N; pc.K = K; pc.Npad = igemm_npad(N); pc.kchunks = igemm_kchunks(K); pc.ntaps = ntaps;
    pc.sh = sh; pc.sw = sw; pc.halo = false; pc.flat3 = false;
    int off[32];
    for (int t = 0; t < ntaps; ++t) {
        const int tap = tap_begin + t;
        pc.tdf[t] = 0;
        pc.tdh[t] = (signed char)(tap / kw - ph);
        pc.tdw[t] = (signed char)(tap % kw - pw);
        off[t] = tap;
    }
    int rc = pc.wp.alloc((size_t)ntaps * pc.kchunks * pc.Npad * 32 * sizeof(float));
    if (rc) return rc;
    const long long taps_all = (long long)kh * kw;
    if ((rc = launch_pack_weights(w, pc.wp.f(), N, pc.Npad, K, ntaps, (long long)K * taps_all, taps_all, off, s))) return rc;
    if (igemm_mode_default() != 0) {
        if ((rc = pc.wp6g.alloc(igemm6_packed_bytes(pc.Npad, K, ntaps)))) return rc;
        if ((rc = launch_pack_weights_g6(w, pc.wp6g.p, N, pc.Npad, K, ntaps, (long long)K * taps_all, taps_all, off, s))) return rc;
        if (igemm_mode_default() == 2)
            if ((rc = f16x3_weight_overflow_check("conv_pack"))) return rc;
    }
    *out = h.release();
    return DPC_OK;
}

void dpc_conv_free(dpc_conv_t h) { delete h; }

// 3-D form of the operator pair (training path of the space-time U-Net): any Conv3d of the net except the 7x7x7 stem, reference
// weight layout [N][K][kd][kh][kw]; *inout != NULL re-packs into the existing handle (same shape: no allocation -- the weights
// change every optimizer step).  No host synchronisation: the f16x3 weight-range flag is read by dpc_weight_range_check.
int dpc_conv3_pack(const float* w, int N, int K, int kd, int kh, int kw, int sh, int sw, int pd, int ph, int pw, const char* mode,
                   dpc_conv_t* inout, dpc_stream_t stream) {
    DPC_REQUIRE(w && inout && N >= 1 && K >= 1 && kd >= 1 && kh >= 1 && kw >= 1, "conv3_pack: bad argument");
    DPC_REQUIRE(K % 4 == 0, "conv3_pack: input channels must be a multiple of 4 (pad on the host)");
    DPC_REQUIRE(kd * kh * kw <= 32, "conv3_pack: at most 32 taps (the stem has dpc_stem_pack)");
    Modes md = modes_global();
    if (mode && mode[0]) {
        const std::string m(mode);
        const int v = m == "f32" ? 0 : (m == "x6" ? 1 : (m == "f16x3" ? 2 : -1));
        DPC_REQUIRE(v >= 0, "conv3_pack: unknown arithmetic mode '" + m + "'");
        md.igemm = v;
        md.conv = v;
    }
    std::unique_ptr<dpc_conv_s> fresh;
    dpc_conv_s* h = *inout;
    if (h) {
        DPC_REQUIRE(h->pc.N == N && h->pc.K == K && h->pc.ntaps == kd * kh * kw && h->modes.conv == md.conv && h->modes.igemm == md.igemm,
                    "conv3_pack: re-pack into a handle of another shape / mode");
    } else {
        fresh = std::make_unique<dpc_conv_s>();
        h = fresh.get();
        h->modes = md;
    }
    ModeScope scope(h->modes);
    if (int rc = pack_conv3d(h->pc, w, N, K, kd, kh, kw, sh, sw, pd, ph, pw, (hipStream_t)stream)) return rc;
    if (fresh) *inout = fresh.release();
    return DPC_OK;
}

int dpc_conv3_run(dpc_conv_t h, const float* a0, const float* a1, int C0, int C1, const float* bias, const float* resid, float* out,
                  int B, int F, int Hi, int Wi, int Ho, int Wo, const float* ln_stats, const float* ln_gamma, int out_mode, int par_a,
                  int par_b, float act_scale, dpc_stream_t stream) {
    DPC_REQUIRE(h && a0 && out && B >= 1 && F >= 1, "conv3_run: null argument");
    if (act_scale != 0.f) {
        int e = 0;
        DPC_REQUIRE(act_scale > 0.f && std::frexp(act_scale, &e) == 0.5f, "conv3_run: act_scale must be a power of two (or 0)");
    }
    ModeScope scope(h->modes);
    return run_conv(h->pc, a0, a1, C0, C1, bias, resid, out, B * F, F, Hi, Wi, Ho, Wo, ln_stats, ln_gamma, out_mode, par_a, par_b,
                    (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, act_scale, 0);
}

int dpc_weight_range_check(void) { return f16x3_weight_overflow_check("weight_range_check"); }

// 7x7x7 stem (init_conv, ...conv3d.py:392) as an operator: x is the reference-layout state [B][F][ctot][H][W] (channel slice
// [coff, coff + C)), out channels-last [B F H W][N].
struct dpc_stem_s {
    dpc::DevBuf wp, ktab, wp6;
    int N = 0, C = 0, k = 0, npad = 0, kchunks = 0;
    bool has6 = false;
    dpc::Modes modes;
};

int dpc_stem_pack(const float* w, int N, int C, int k, const char* mode, dpc_stem_t* inout, dpc_stream_t stream) {
    DPC_REQUIRE(w && inout && N >= 1 && C >= 1 && k >= 1 && (k & 1), "stem_pack: bad argument");
    Modes md = modes_global();
    if (mode && mode[0]) {
        const std::string m(mode);
        const int v = m == "f32" ? 0 : (m == "x6" ? 1 : (m == "f16x3" ? 2 : -1));
        DPC_REQUIRE(v >= 0, "stem_pack: unknown arithmetic mode '" + m + "'");
        md.stem = v;
    }
    std::unique_ptr<dpc_stem_s> fresh;
    dpc_stem_s* h = *inout;
    if (h) {
        DPC_REQUIRE(h->N == N && h->C == C && h->k == k && h->modes.stem == md.stem, "stem_pack: re-pack into a handle of another shape");
    } else {
        fresh = std::make_unique<dpc_stem_s>();
        h = fresh.get();
        h->modes = md; h->N = N; h->C = C; h->k = k;
    }
    ModeScope scope(h->modes);
    hipStream_t s = (hipStream_t)stream;
    h->npad = (int)align_up(N, 64);
    h->kchunks = igemm_kchunks(k * k * k * C);
    int rc;
    if ((rc = h->wp.alloc((size_t)h->kchunks * h->npad * 32 * sizeof(float)))) return rc;
    if ((rc = h->ktab.alloc((size_t)h->kchunks * 32 * sizeof(int)))) return rc;
    if ((rc = launch_pack_stem(w, h->wp.f(), (int*)h->ktab.p, N, h->npad, C, k, s))) return rc;
    h->has6 = h->modes.stem != 0 && stem7x6_supported(C, k);
    if (h->has6) {
        if ((rc = h->wp6.alloc(stem7x6_packed_bytes(h->npad)))) return rc;
        if ((rc = launch_pack_stem7x6(w, h->wp6.p, N, h->npad, C, s))) return rc;
    }
    if (fresh) *inout = fresh.release();
    return DPC_OK;
}

void dpc_stem_free(dpc_stem_t h) { delete h; }

int dpc_stem_run(dpc_stem_t h, const float* x, int x_channels_total, int x_channel_offset, const float* bias, float* out, int B, int F,
                 int H, int W, dpc_stream_t stream) {
    DPC_REQUIRE(h && x && out && B >= 1 && F >= 1, "stem_run: null argument");
    DPC_REQUIRE(x_channel_offset >= 0 && x_channel_offset + h->C <= x_channels_total, "stem_run: bad channel slice");
    ModeScope scope(h->modes);
    StemParams sp{};
    sp.x = x; sp.wp = h->wp.f(); sp.ktab = (const int*)h->ktab.p; sp.bias = bias; sp.out = out;
    sp.BF = B * F; sp.F = F; sp.C = h->C; sp.H = H; sp.W = W; sp.Ctot = x_channels_total; sp.c_off = x_channel_offset;
    sp.N = h->N; sp.Npad = h->npad; sp.kchunks = h->kchunks; sp.M = (long long)B * F * H * W;
    if (h->has6) return launch_stem7x6(sp, h->wp6.p, (hipStream_t)stream);
    return launch_stem(sp, (hipStream_t)stream);
}

int dpc_conv_run(dpc_conv_t h, const float* a0, const float* a1, int C0, int C1, const float* bias, const float* resid, float* out,
                 int BF, int Hi, int Wi, int Ho, int Wo, const float* ln_stats, const float* ln_gamma, int out_mode, int par_a,
                 int par_b, float act_scale, int a0_stride, dpc_stream_t stream) {
    DPC_REQUIRE(h && a0 && out, "conv_run: null argument");
    DPC_REQUIRE(a0_stride == 0 || (a0_stride % 4 == 0 && a0_stride <= C0 && !a1 && !ln_stats), "conv_run: bad a0_stride");
    if (act_scale != 0.f) {
        int e = 0;
        DPC_REQUIRE(act_scale > 0.f && std::frexp(act_scale, &e) == 0.5f, "conv_run: act_scale must be a power of two (or 0)");
    }
    ModeScope scope(h->modes);
    return run_conv(h->pc, a0, a1, C0, C1, bias, resid, out, BF, 1, Hi, Wi, Ho, Wo, ln_stats, ln_gamma, out_mode, par_a, par_b,
                    (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr, act_scale, a0_stride);
}

// ---- GroupNorm fused around a ResnetBlock's two 3x3 convolutions (r05; see include/dpc.h)
int dpc_conv_gn_fusable(dpc_conv_t h, int H, int W, int C0, int C1) {
    // exactly run_conv's condition for the halo-tile path of a stride-1 3x3 convolution (unet3d.hip: flat_halo) -- a split input whose parts
    // are not multiples of 4 channels, or DPC_CONV2D_HALO=0, must send the caller to its unfused GroupNorm passes, not into a
    // DPC_REQUIRE of dpc_conv_run_gn
    if (!h || C0 + C1 != h->pc.K) return 0;
    ModeScope scope(h->modes);
    return (h->pc.flat3 && C0 % 4 == 0 && C1 % 4 == 0 && conv2d_gn_fusable(h->pc.N, h->pc.Npad, H, W)) ? 1 : 0;
}
int64_t dpc_conv_gn_entries(int H, int W) { return conv3f3c_flat_gn_entries(H, W); }
int dpc_conv_run_gn(dpc_conv_t h, const float* a0, const float* a1, int C0, int C1, const float* bias, float* out, int images, int H, int W,
                    float* gn_part, const float* in_coef, dpc_stream_t stream) {
    DPC_REQUIRE(h && a0 && out && images >= 1, "conv_run_gn: null argument");
    DPC_REQUIRE(gn_part || in_coef, "conv_run_gn: neither statistics output nor input coefficients (use dpc_conv_run)");
    DPC_REQUIRE(!in_coef || (!a1 && C1 == 0), "conv_run_gn: a fused input normalisation needs a single source");
    DPC_REQUIRE(dpc_conv_gn_fusable(h, H, W, C0, a1 ? C1 : 0), "conv_run_gn: this convolution / image size does not take the halo kernel (dpc_conv_gn_fusable)");
    ModeScope scope(h->modes);
    return run_conv(h->pc, a0, a1, C0, C1, bias, nullptr, out, images, 1, H, W, H, W, nullptr, nullptr, 0, 0, 0, (hipStream_t)stream, gn_part,
                    in_coef, nullptr, nullptr, 0.f, 0);
}
int dpc_gn_finalize_fused(const float* part, int images, int64_t entries, int C, int groups, int64_t rows_per_image, const float* gamma,
                          const float* beta, const float* scale_shift, float* stats, float* coef, dpc_stream_t stream) {
    DPC_REQUIRE(part && stats && images >= 1 && entries >= 1, "gn_finalize_fused: bad argument");
    DPC_REQUIRE(!coef || (gamma && beta), "gn_finalize_fused: the coefficient table needs gamma and beta");
    return launch_gn_finalize_fused(part, images, 0, C, groups, rows_per_image, gamma, beta, scale_shift, stats, coef, (hipStream_t)stream, entries);
}

size_t dpc_gn_workspace_bytes(int B, int C) { return std::max(gn_workspace_bytes(B, C), gn_bwd_workspace_bytes(B, C)) + 256; }

int dpc_gn_stats(const float* x, float* stats, int B, int64_t R, int C, int groups, void* ws, size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(x && stats && ws && ws_bytes >= dpc_gn_workspace_bytes(B, C), "gn_stats: bad argument / workspace too small");
    return launch_gn_stats(x, stats, B, R, C, groups, reinterpret_cast<void*>(align_up((size_t)ws, 256)), (hipStream_t)stream);
}

int dpc_gn_apply(const float* x, float* out, const float* resid, const float* stats, const float* gamma, const float* beta,
                 const float* scale_shift, int B, int64_t R, int C, int groups, dpc_stream_t stream) {
    DPC_REQUIRE(x && out && stats && gamma && beta, "gn_apply: null argument");
    return launch_gn_apply(x, out, resid, stats, gamma, beta, scale_shift, B, R, C, groups, (hipStream_t)stream);
}

int dpc_gn_silu_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                    const float* scale_shift, float* dx, float* dss, int B, int64_t R, int C, int groups, void* ws, size_t ws_bytes,
                    dpc_stream_t stream) {
    DPC_REQUIRE(x && dy && stats && gamma && beta && dx && ws && ws_bytes >= dpc_gn_workspace_bytes(B, C),
                "gn_silu_bwd: bad argument / workspace too small");
    return launch_gn_silu_bwd(x, dy, stats, gamma, beta, scale_shift, dx, dss, B, R, C, groups,
                              reinterpret_cast<void*>(align_up((size_t)ws, 256)), (hipStream_t)stream);
}

int dpc_gn_silu_bwd_params(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                           const float* scale_shift, float* dx, float* dss, float* dgamma, float* dbeta, int B, int64_t R, int C,
                           int groups, void* ws, size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(x && dy && stats && gamma && beta && dx && dgamma && dbeta && ws && ws_bytes >= dpc_gn_workspace_bytes(B, C),
                "gn_silu_bwd_params: bad argument / workspace too small");
    return launch_gn_silu_bwd(x, dy, stats, gamma, beta, scale_shift, dx, dss, B, R, C, groups,
                              reinterpret_cast<void*>(align_up((size_t)ws, 256)), (hipStream_t)stream, dgamma, dbeta);
}

int dpc_ln_stats(const float* x, float* stats, int64_t rows, int C, dpc_stream_t stream) {
    DPC_REQUIRE(x && stats, "ln_stats: null argument");
    return launch_ln_stats(x, stats, rows, C, (hipStream_t)stream);
}

int dpc_ln_apply(const float* x, const float* stats, const float* g, const float* resid, float* out, int64_t rows, int C,
                 dpc_stream_t stream) {
    DPC_REQUIRE(x && stats && g && out, "ln_apply: null argument");
    return launch_ln_apply(x, stats, g, resid, out, rows, C, (hipStream_t)stream);
}

int dpc_ln_bwd(const float* x, const float* stats, const float* g, const float* dy, float* dx, int64_t rows, int C, int accumulate,
               dpc_stream_t stream) {
    DPC_REQUIRE(x && stats && g && dy && dx, "ln_bwd: null argument");
    return launch_ln_bwd(x, stats, g, dy, dx, rows, C, accumulate, (hipStream_t)stream);
}

size_t dpc_linear_attention_tape_bytes(int64_t images, int heads) { return (size_t)images * heads * LAB_WS * sizeof(float) + 256; }

int dpc_linear_attention_fwd_save(const float* qkv, float* out, int heads, int64_t images, int N, void* tape, size_t tape_bytes,
                                  dpc_stream_t stream) {
    DPC_REQUIRE(qkv && out && tape && tape_bytes >= dpc_linear_attention_tape_bytes(images, heads),
                "linear_attention_fwd_save: bad argument / tape too small");
    return launch_linear_attention(qkv, out, heads, images, N, reinterpret_cast<void*>(align_up((size_t)tape, 256)), (hipStream_t)stream,
                                   LAB_WS, 1);
}

int dpc_linear_attention_bwd(const float* qkv, const float* dout, float* dqkv, int heads, int64_t images, int N, void* tape,
                             size_t tape_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(qkv && dout && dqkv && tape && tape_bytes >= dpc_linear_attention_tape_bytes(images, heads),
                "linear_attention_bwd: bad argument / tape too small");
    if (images == 0) return DPC_OK;
    const int tiles = (N + 31) / 32;
    DPC_REQUIRE(images * heads < (1ll << 31) && images * tiles < (1ll << 31), "linear_attention_bwd: grid too large");
    hipStream_t s = (hipStream_t)stream;
    float* w = reinterpret_cast<float*>(align_up((size_t)tape, 256));
    const double rows_ = (double)images * N;
    ProfScope prof(PROF_LINATTN, 8.0 * rows_ * 32 * 32 * heads, 4.0 * rows_ * heads * 32 * 10, s);
    hipLaunchKernelGGL(linattn_bwd_ctx_kernel, dim3((unsigned)(images * heads)), dim3(256), 0, s, qkv, dout, w, heads, N);
    DPC_LAUNCH_CHECK();
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)linattn_bwd_tok_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LAB_LDS));
        once = true;
    }
    hipLaunchKernelGGL(linattn_bwd_tok_kernel, dim3((unsigned)(images * tiles), (heads + 3) / 4), dim3(256), LAB_LDS, s, qkv, dout, w, dqkv,
                       heads, N, tiles);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_attention_bwd(const float* qkv, const float* dout, float* dqkv, int heads, int64_t images, int L, dpc_stream_t stream) {
    DPC_REQUIRE(qkv && dout && dqkv && L >= 1 && L <= 256, "attention_bwd: bad argument (1 <= L <= 256 tokens)");
    if (images == 0) return DPC_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)4 * L * 33 + 3 * L) * sizeof(float);
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (4 * 256 * 33 + 3 * 256) * 4));
        once = true;
    }
    ProfScope prof(PROF_ATTN, 10.0 * (double)images * heads * L * L * 32, 0, s);
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(images * heads)), dim3(256), lds, s, qkv, dout, dqkv, heads, L);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_upsample2x_cl(const float* x, float* y, int BF, int H, int W, int C, dpc_stream_t stream) {
    DPC_REQUIRE(x && y && C % 4 == 0, "upsample2x_cl: bad argument");
    return launch_upsample2x_cl(x, y, BF, H, W, C, (hipStream_t)stream);
}

int dpc_downsum2x_cl(const float* x, float* y, int BF, int H, int W, int C, dpc_stream_t stream) {
    DPC_REQUIRE(x && y && C % 4 == 0, "downsum2x_cl: bad argument");
    const long long total4 = (long long)BF * H * W * (C / 4);
    if (total4 == 0) return DPC_OK;
    hipLaunchKernelGGL(downsum2x_cl_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, total4, H, W, C / 4);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_nchw_to_cl(const float* x, float* y, int64_t N, int C, int Cpad, int64_t HW, dpc_stream_t stream) {
    DPC_REQUIRE(x && y && Cpad >= C, "nchw_to_cl: bad argument");
    const long long total = N * HW * Cpad;
    if (total == 0) return DPC_OK;
    hipLaunchKernelGGL(nchw_to_cl_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, C, Cpad, HW);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_cl_to_nchw(const float* x, float* y, int64_t N, int C, int Cpad, int csrc, int Ctot, int cdst, float mul, int64_t HW,
                   dpc_stream_t stream) {
    DPC_REQUIRE(x && y && csrc >= 0 && cdst >= 0 && csrc + C <= Cpad && cdst + C <= Ctot, "cl_to_nchw: bad argument");
    const long long total = N * HW * C;
    if (total == 0) return DPC_OK;
    hipLaunchKernelGGL(cl_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, C, Cpad, csrc, Ctot, cdst,
                       mul, HW);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_channel_affine_to_cl(const float* x, float* y, int64_t N, int Ctot, int csrc, int Cpad, int cdst, float a, float b, int64_t HW,
                             dpc_stream_t stream) {
    DPC_REQUIRE(x && y && csrc >= 0 && csrc < Ctot && cdst >= 0 && cdst < Cpad, "channel_affine_to_cl: bad argument");
    const long long total = N * HW;
    if (total == 0) return DPC_OK;
    hipLaunchKernelGGL(channel_affine_to_cl_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, total, Ctot, csrc, Cpad,
                       cdst, a, b, HW);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_channel_mean(const float* x, float* out, int64_t N, int Ctot, int csrc, int64_t HW, dpc_stream_t stream) {
    DPC_REQUIRE(x && out && csrc >= 0 && csrc < Ctot, "channel_mean: bad argument");
    if (N == 0) return DPC_OK;
    hipLaunchKernelGGL(channel_mean_kernel, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, x, out, Ctot, csrc, HW);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_channel_fill(float* y, const float* v, int64_t N, int Ctot, int cdst, float mul, int64_t HW, dpc_stream_t stream) {
    DPC_REQUIRE(y && v && cdst >= 0 && cdst < Ctot, "channel_fill: bad argument");
    const long long total = N * HW;
    if (total == 0) return DPC_OK;
    hipLaunchKernelGGL(channel_fill_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, v, total, Ctot, cdst, mul, HW);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_mean_rows(const float* x, float* out, int64_t N, int64_t R, int C, dpc_stream_t stream) {
    DPC_REQUIRE(x && out, "mean_rows: null argument");
    if (N == 0) return DPC_OK;
    hipLaunchKernelGGL(mean_rows_kernel, dim3((C + 63) / 64, (unsigned)N), dim3(256), 0, (hipStream_t)stream, x, out, R, C);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_bcast_rows(const float* v, float* y, int64_t N, int64_t R, int C, float mul, dpc_stream_t stream) {
    DPC_REQUIRE(v && y && C % 4 == 0, "bcast_rows: bad argument");
    const long long total4 = N * R * (C / 4);
    if (total4 == 0) return DPC_OK;
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, v, y, total4, R, C / 4, mul);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_add_inplace(float* y, const float* x, int64_t n, dpc_stream_t stream) {
    DPC_REQUIRE(y && x && n % 4 == 0, "add_inplace: bad argument");
    if (n == 0) return DPC_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, y, x, n / 4);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_pad_w_cl(const float* x, float* y, int64_t rows, int W, int C, int wpad, dpc_stream_t stream) {
    DPC_REQUIRE(x && y && C % 4 == 0 && wpad >= 0, "pad_w_cl: bad argument");
    const long long total4 = rows * (W + 2 * wpad) * (C / 4);
    if (total4 == 0) return DPC_OK;
    hipLaunchKernelGGL(pad_w_cl_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, total4, W, C / 4, wpad);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_fold_w_cl(const float* x, float* dx, int64_t rows, int W, int C, int taps, dpc_stream_t stream) {
    DPC_REQUIRE(x && dx && taps >= 1, "fold_w_cl: bad argument");
    const long long total = rows * W * C;
    if (total == 0) return DPC_OK;
    hipLaunchKernelGGL(fold_w_cl_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, dx, total, W, C, taps);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_absmax(const float* x, int64_t n, float* out, dpc_stream_t stream) {
    DPC_REQUIRE(x && out && n % 4 == 0, "absmax: bad argument");
    if (n == 0) return DPC_OK;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<unsigned*>(out), n / 4);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_small_linear(const float* in, const float* W, const float* bias, float* out, int B, int K, int N, int in_act, int out_act,
                     dpc_stream_t stream) {
    DPC_REQUIRE(in && W && out, "small_linear: null argument");
    return launch_small_linear(in, W, bias, out, B, K, N, in_act, out_act, (hipStream_t)stream);
}

}  // extern "C"
