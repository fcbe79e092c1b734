// Fused spatial linear attention block: Residual(PreNorm(SpatialLinearAttention)) per frame.
//   q,k,v = to_qkv(LayerNorm_c(x));  q = softmax_d(q) * s;  k = softmax_n(k);  ctx = k v^T;  out = ctx^T q;
//   y = x + to_out(out) + b
// Reference: video_diffusion_pytorch_conv3d.py:232-257 (SpatialLinearAttention), :441 (Residual(PreNorm(...))).
//
// Two launches, the qkv tensor never reaches HBM:
//   lattn_ctx_kernel : one workgroup per (frame, head).  Each wave streams 32-token tiles: LayerNorm in registers,
//                      K = xn Wk^T and V = xn Wv^T on the MFMA, ONLINE softmax over tokens per head-dim column
//                      (the column lives in one lane pair, so max / sum / rescale are lane-local), and the
//                      32x32 context accumulated as ctx^T = V^T exp(K) (A = V regs, B = exp(K) regs).
//   lattn_out_kernel : one wave per 32-token tile: Q^T = Wq xn^T (softmax over head dims is lane-local),
//                      out^T = ctx^T-fragment . q, then y += out Wout_h^T, + bias + residual.
// Operand orientation trick as in tattn_fused.hip: every intermediate is produced in the register layout the next
// MFMA consumes, so there are no LDS transposes; LDS only stages the current head's weight slices.
#include "common.h"

namespace dpc {

__device__ __forceinline__ int rowmap_l(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <int C>
__device__ __forceinline__ void load_ln_rows(const float* x, long long row, bool ok, const float* gamma, int hh,
                                             f32x4 (&xa)[C / 8]) {
    constexpr int CJ = C / 8;
    const float* src = x + row * C + 4 * hh;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(src + 8 * j);
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
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 8 * j + 4 * hh);
        xa[j] = ok ? (xa[j] - mean) * inv * g : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int C>
__global__ __launch_bounds__(256, 2) void lattn_ctx_kernel(LattnParams p) {
    constexpr int CJ = C / 8, WST = C + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                         // [64][WST]: rows 0..31 = Wk_h, 32..63 = Wv_h
    float* s_m = Ws + 64 * WST;               // [4][32]
    float* s_z = s_m + 128;                   // [4][32]
    float* s_ctx = s_z + 128;                 // [4][16][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const long long img = blockIdx.x / 4;
    const int hd = blockIdx.x % 4;
    const int N = p.N;
    for (int q4 = tid; q4 < 64 * (C / 4); q4 += 256) {
        const int rrow = q4 / (C / 4), c4 = q4 % (C / 4);
        const int grow = 128 + (rrow >> 5) * 128 + hd * 32 + (rrow & 31);       // k | v blocks of to_qkv.weight
        *reinterpret_cast<f32x4*>(&Ws[rrow * WST + c4 * 4]) =
            *reinterpret_cast<const f32x4*>(p.wqkv + (long long)grow * C + c4 * 4);
    }
    __syncthreads();

    f32x16 ctxT;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctxT[r] = 0.f;
    float m = -INFINITY, z = 0.f;
    const int ntiles = (N + 31) / 32;
    for (int t = wave; t < ntiles; t += 4) {
        const int n = t * 32 + l31;
        f32x4 xa[CJ];
        load_ln_rows<C>(p.x, img * N + (n < N ? n : 0), n < N, p.gamma, hh, xa);
        f32x16 kk, vv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { kk[r] = 0.f; vv[r] = 0.f; }
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const f32x4 wk = *reinterpret_cast<const f32x4*>(&Ws[l31 * WST + 8 * j + 4 * hh]);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&Ws[(32 + l31) * WST + 8 * j + 4 * hh]);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                kk = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j][s], wk[s], kk, 0, 0, 0);     // [token][d]: lane = d
                vv = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[j][s], wv[s], vv, 0, 0, 0);     // [token][e]: lane = e
            }
        }
        float tm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (t * 32 + rowmap_l(r, hh) >= N) kk[r] = -INFINITY;
            tm = fmaxf(tm, kk[r]);
        }
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
        const float m_new = fmaxf(m, tm);
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - m_new);
        float zs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(kk[r] - m_new);
            kk[r] = e;
            zs += e;
        }
        z = z * alpha + zs;                 // per half-wave partial; halves are added at the end
        m = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctxT[r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) ctxT = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r], kk[r], ctxT, 0, 0, 0);   // [e][d]: lane = d
    }
    z += __shfl_xor(z, 32, 64);
    if (hh == 0) { s_m[wave * 32 + l31] = m; s_z[wave * 32 + l31] = z; }
#pragma unroll
    for (int r = 0; r < 16; ++r) s_ctx[(wave * 16 + r) * 64 + lane] = ctxT[r];
    __syncthreads();
    if (wave == 0) {
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, s_m[w * 32 + l31]);
        float sc[4], Z = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = s_m[w * 32 + l31];
            sc[w] = (mw == -INFINITY) ? 0.f : expf(mw - M);
            Z += s_z[w * 32 + l31] * sc[w];
        }
        float* dst = p.ctx + ((long long)img * 4 + hd) * 1024;          // stored transposed: [e][d]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += s_ctx[(w * 16 + r) * 64 + lane] * sc[w];
            dst[rowmap_l(r, hh) * 32 + l31] = v / Z;
        }
    }
}

template <int C>
__global__ __launch_bounds__(256, 2) void lattn_out_kernel(LattnParams p) {
    constexpr int CJ = C / 8, WST = C + 4, NTC = C / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                         // [32][WST]  Wq_h
    float* Wo = Ws + 32 * WST;                // [C][36]    to_out columns of head h
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
    const int N = p.N;
    const int tpi = (N + 31) / 32;            // tiles per image
    const long long tile = (long long)blockIdx.x * 4 + wave;
    const bool active = tile < p.images * tpi;
    const long long img = active ? tile / tpi : 0;
    const int t = active ? (int)(tile % tpi) : 0;
    const int n = t * 32 + l31;
    const bool ok = active && n < N;
    const float scale = 0.17677669529663687f;
    f32x4 xa[CJ];
    load_ln_rows<C>(p.x, img * N + (ok ? n : 0), ok, p.gamma, hh, xa);

    f32x16 y[NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[nt][r] = 0.f;

    for (int hd = 0; hd < 4; ++hd) {
        __syncthreads();
        for (int q4 = tid; q4 < 32 * (C / 4); q4 += 256) {
            const int rrow = q4 / (C / 4), c4 = q4 % (C / 4);
            *reinterpret_cast<f32x4*>(&Ws[rrow * WST + c4 * 4]) =
                *reinterpret_cast<const f32x4*>(p.wqkv + (long long)(hd * 32 + rrow) * C + c4 * 4);
        }
        for (int q4 = tid; q4 < C * 8; q4 += 256) {
            const int c = q4 >> 3, d4 = q4 & 7;
            *reinterpret_cast<f32x4*>(&Wo[c * 36 + d4 * 4]) =
                *reinterpret_cast<const f32x4*>(p.wout + (long long)c * 128 + hd * 32 + d4 * 4);
        }
        __syncthreads();
        f32x16 qT;
#pragma unroll
        for (int r = 0; r < 16; ++r) qT[r] = 0.f;
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const f32x4 wq = *reinterpret_cast<const f32x4*>(&Ws[l31 * WST + 8 * j + 4 * hh]);
#pragma unroll
            for (int s = 0; s < 4; ++s) qT = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[s], xa[j][s], qT, 0, 0, 0);   // [d][token]: lane = token
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, qT[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(qT[r] - mx);
            qT[r] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) qT[r] = (qT[r] / sum) * scale;
        // out^T[e][n] = sum_d ctx[d][e] q[n][d]     (A = ctx^T fragment: lane = e, k = d; B = q regs: lane = token)
        const float* cbase = p.ctx + ((long long)img * 4 + hd) * 1024 + l31 * 32 + 4 * hh;
        f32x16 oT;
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[r] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const f32x4 cf = *reinterpret_cast<const f32x4*>(cbase + 8 * jj);
#pragma unroll
            for (int s = 0; s < 4; ++s) oT = __builtin_amdgcn_mfma_f32_32x32x2f32(cf[s], qT[4 * jj + s], oT, 0, 0, 0);
        }
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
    if (active) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const float bv = p.bout[nt * 32 + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = t * 32 + rowmap_l(r, hh);
                if (nn < N) {
                    const long long o = (img * N + nn) * C + nt * 32 + l31;
                    p.out[o] = (y[nt][r] + bv) + p.x[o];
                }
            }
        }
    }
}

bool lattn_fused_supported(int C, int heads) { return (C == 64 || C == 128) && heads == 4; }

size_t lattn_fused_workspace_bytes(long long images) { return (size_t)images * 4 * 1024 * sizeof(float); }

template <int C>
static int launch_lattn_t(const LattnParams& p, hipStream_t s) {
    const size_t lds1 = (64 * (C + 4) + 256 + 4 * 16 * 64) * sizeof(float);
    const size_t lds2 = (32 * (C + 4) + C * 36) * sizeof(float);
    hipLaunchKernelGGL(lattn_ctx_kernel<C>, dim3((unsigned)(p.images * 4)), dim3(256), lds1, s, p);
    DPC_LAUNCH_CHECK();
    const long long tiles = p.images * ((p.N + 31) / 32);
    hipLaunchKernelGGL(lattn_out_kernel<C>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), lds2, s, p);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_lattn_fused(const LattnParams& p, int C, hipStream_t s) {
    DPC_REQUIRE(lattn_fused_supported(C, 4), "lattn_fused: unsupported shape");
    if (p.images == 0) return DPC_OK;
    DPC_REQUIRE(p.images * 4 < (1ll << 31), "lattn_fused: grid too large");
    const double rows = (double)p.images * p.N;
    ProfScope prof(PROF_LATTN_FUSED, 2.0 * rows * C * 384 + 4.0 * rows * 32 * 32 * 4 + 2.0 * rows * 128 * C,
                   4.0 * rows * C * 3, s);
    return C == 64 ? launch_lattn_t<64>(p, s) : launch_lattn_t<128>(p, s);
}

}  // namespace dpc
