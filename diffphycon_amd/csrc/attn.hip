// Attention cores on a packed qkv tensor [rows][3*heads*32] (head dim 32 = one fp32 MFMA k-tile).
//
// attention_kernel  : softmax(q k^T * scale + bias) v for sequences of L tokens addressed by strides, optional
//                     interleaved-pair rotary on q,k.  Serves the temporal attention (L = frames, one sequence
//                     per pixel) and the bottleneck dense spatial attention (L = H*W, one sequence per frame).
//                     One wave per (sequence, head, 32-query tile); flash-style loop over 32-key tiles.
//                     The score tile is computed TRANSPOSED (S^T = K Q^T) so that a query's 32 scores sit in
//                     the registers of one lane pair (l, l^32): row max / sum are in-lane + one cross-half swap,
//                     and the same registers are directly the A operand of the P V product.
// linattn_ctx/out   : SpatialLinearAttention core: context = softmax_n(k)^T v (32x32 per image,head), then
//                     out = context^T softmax_d(q) * scale.
// Reference: video_diffusion_pytorch_conv3d.py:311-351 (Attention), :246-255 (SpatialLinearAttention).
#include "common.h"

namespace dpc {

__device__ __forceinline__ int rowmap(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__global__ __launch_bounds__(256) void attention_kernel(AttnParams p, long long total_waves, int qtiles) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total_waves) return;
    const int head = (int)(wid % p.heads);
    long long rest = wid / p.heads;
    const int qt = (int)(rest % qtiles);
    const long long seq = rest / qtiles;
    const int ld = 3 * p.heads * 32;
    const int HD = p.heads * 32;
    const long long base_row = (seq / p.seq_inner) * p.seq_outer_stride + (seq % p.seq_inner) * p.seq_inner_stride;
    const float scale = 0.17677669529663687f;   // 32^-0.5, as fp32(dim_head ** -0.5)

    // ---- this lane's query row (B operand of S^T = K Q^T): q[i][8jj+4hh+0..3]
    const int qi = qt * 32 + l31;
    const bool qok = qi < p.L;
    f32x4 q[4];
    {
        const float* src = p.qkv + (base_row + (long long)(qok ? qi : 0) * p.token_stride) * ld + head * 32 + 4 * hh;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x4 v = {0, 0, 0, 0};
            if (qok) v = *reinterpret_cast<const f32x4*>(src + 8 * jj);
            v = v * scale;
            if (p.rot_cos && qok) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.rot_cos + qi * 32 + 8 * jj + 4 * hh);
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.rot_sin + qi * 32 + 8 * jj + 4 * hh);
                f32x4 o;
                o.x = __fadd_rn(__fmul_rn(v.x, c.x), __fmul_rn(-v.y, s.x));
                o.y = __fadd_rn(__fmul_rn(v.y, c.y), __fmul_rn(v.x, s.y));
                o.z = __fadd_rn(__fmul_rn(v.z, c.z), __fmul_rn(-v.w, s.z));
                o.w = __fadd_rn(__fmul_rn(v.w, c.w), __fmul_rn(v.z, s.w));
                v = o;
            }
            q[jj] = v;
        }
    }

    f32x16 o_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int nkt = (p.L + 31) / 32;

    for (int kt = 0; kt < nkt; ++kt) {
        // ---- key rows (A operand): k[j][8jj+4hh+0..3], rotary applied
        const int kj = kt * 32 + l31;
        const bool kok = kj < p.L;
        f32x4 k[4];
        {
            const float* src = p.qkv + (base_row + (long long)(kok ? kj : 0) * p.token_stride) * ld + HD + head * 32 + 4 * hh;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                f32x4 v = {0, 0, 0, 0};
                if (kok) v = *reinterpret_cast<const f32x4*>(src + 8 * jj);
                if (p.rot_cos && kok) {
                    const f32x4 c = *reinterpret_cast<const f32x4*>(p.rot_cos + kj * 32 + 8 * jj + 4 * hh);
                    const f32x4 s = *reinterpret_cast<const f32x4*>(p.rot_sin + kj * 32 + 8 * jj + 4 * hh);
                    f32x4 o;
                    o.x = __fadd_rn(__fmul_rn(v.x, c.x), __fmul_rn(-v.y, s.x));
                    o.y = __fadd_rn(__fmul_rn(v.y, c.y), __fmul_rn(v.x, s.y));
                    o.z = __fadd_rn(__fmul_rn(v.z, c.z), __fmul_rn(-v.w, s.z));
                    o.w = __fadd_rn(__fmul_rn(v.w, c.w), __fmul_rn(v.z, s.w));
                    v = o;
                }
                k[jj] = v;
            }
        }
        // ---- value column (A operand of O = P V is P; B operand rows j, col d = l31)
        float vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = kt * 32 + rowmap(r, hh);
            vv[r] = (j < p.L) ? p.qkv[(base_row + (long long)j * p.token_stride) * ld + 2 * HD + head * 32 + l31] : 0.f;
        }
        // ---- S^T tile: rows = keys, cols = queries
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(k[jj][s], q[jj][s], st, 0, 0, 0);
        // ---- bias, mask, online softmax over keys (in-lane + partner lane l^32)
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = kt * 32 + rowmap(r, hh);
            float sv = st[r];
            if (p.bias && qok && j < p.L) sv += p.bias[((long long)head * p.L + qi) * p.L + j];
            if (j >= p.L) sv = -INFINITY;
            st[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (kt == 0) ? 0.f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = expf(st[r] - m_new);
            st[r] = pv;
            psum += pv;
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (kt > 0) {
            // rescale O rows: row i of the accumulator belongs to query lane i
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[r] *= __shfl(alpha, rowmap(r, hh), 64);
        }
        // ---- O[i][d] += sum_j P[i][j] V[j][d]  (A = P from the S^T registers, B = V column)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vv[r], o_acc, 0, 0, 0);
    }
    // ---- normalise and store: out[token i][head*32 + d], d = l31
    const float inv_l = 1.0f / l_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = rowmap(r, hh);
        const float il = __shfl(inv_l, i, 64);
        const int ti = qt * 32 + i;
        if (ti < p.L) p.out[(base_row + (long long)ti * p.token_stride) * HD + head * 32 + l31] = o_acc[r] * il;
    }
}

int launch_attention(const AttnParams& p, hipStream_t s) {
    DPC_REQUIRE(p.heads >= 1 && p.L >= 1, "attention: heads, L");
    const int qtiles = (p.L + 31) / 32;
    const long long total = p.n_seq * qtiles * p.heads;
    if (total == 0) return DPC_OK;
    const long long grid = (total + 3) / 4;
    DPC_REQUIRE(grid < (1ll << 31), "attention: grid too large");
    const double rows = (double)p.n_seq * p.L;
    ProfScope prof(PROF_ATTN, 4.0 * rows * p.L * 32 * p.heads, 4.0 * rows * 4 * p.heads * 32, s);
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)grid), dim3(256), 0, s, p, total, qtiles);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------------------------ linear attention
size_t linattn_workspace_bytes(long long images, int heads) { return (size_t)images * heads * 1024 * sizeof(float); }

// ctx_stride: floats between the [32][32] context blocks of consecutive (image, head) pairs (1024: packed).  save != 0 (the
// backward pass's tape, surr.hip): the column maximum of k and its exp-sum follow the context at +1024 / +1056.
__global__ __launch_bounds__(256) void linattn_ctx_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                          int heads, int N, int ctx_stride, int save) {
    __shared__ float s_max[4][32];
    __shared__ float s_z[4][32];
    __shared__ float s_acc[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hh = lane >> 5;
    const long long img = blockIdx.x / heads;
    const int head = blockIdx.x % heads;
    const int ld = 3 * heads * 32, HD = heads * 32;
    const float* kbase = qkv + img * N * (long long)ld + HD + head * 32 + l31;
    const float* vbase = kbase + HD;
    int per = (N + 3) / 4;
    per += per & 1;
    const int n_begin = wave * per, n_end = min(N, n_begin + per);

    // phase 1: column max of k over tokens
    float m = -INFINITY;
    for (int n = n_begin + hh; n < n_end; n += 2) m = fmaxf(m, kbase[(long long)n * ld]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if (hh == 0) s_max[wave][l31] = m;
    __syncthreads();
    const float kmax = fmaxf(fmaxf(s_max[0][l31], s_max[1][l31]), fmaxf(s_max[2][l31], s_max[3][l31]));

    // phase 2: ctx[d][e] = sum_n exp(k[n][d]-kmax[d]) v[n][e],  Z[d] = sum_n exp(...)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float z = 0.f;
#pragma unroll 4
    for (int n0 = n_begin; n0 < n_end; n0 += 2) {
        const int n = n0 + hh;
        const bool ok = n < n_end;
        const float a = ok ? expf(kbase[(long long)n * ld] - kmax) : 0.f;
        const float b = ok ? vbase[(long long)n * ld] : 0.f;
        z += a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    z += __shfl_xor(z, 32, 64);
    if (hh == 0) s_z[wave][l31] = z;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
        float* dst = ctx + ((long long)img * heads + head) * ctx_stride;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = rowmap(r, hh);
            const float tot = (s_acc[0][r][lane] + s_acc[1][r][lane]) + (s_acc[2][r][lane] + s_acc[3][r][lane]);
            const float zz = (s_z[0][d] + s_z[1][d]) + (s_z[2][d] + s_z[3][d]);
            dst[d * 32 + l31] = tot / zz;
        }
        if (save && hh == 0) {
            dst[1024 + l31] = kmax;
            dst[1056 + l31] = (s_z[0][l31] + s_z[1][l31]) + (s_z[2][l31] + s_z[3][l31]);
        }
    }
}

__global__ __launch_bounds__(256) void linattn_out_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                          float* __restrict__ out, int heads, int N,
                                                          long long total_waves, int tiles_per_img, int ctx_stride) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= total_waves) return;
    const int head = (int)(wid % heads);
    const long long rest = wid / heads;
    const int tile = (int)(rest % tiles_per_img);
    const long long img = rest / tiles_per_img;
    const int ld = 3 * heads * 32, HD = heads * 32;
    const int n = tile * 32 + l31;
    const bool ok = n < N;
    const float scale = 0.17677669529663687f;
    f32x4 q[4];
    const float* src = qkv + (img * N + (ok ? n : 0)) * (long long)ld + head * 32 + 4 * hh;
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        q[jj] = *reinterpret_cast<const f32x4*>(src + 8 * jj);
        m = fmaxf(fmaxf(m, fmaxf(q[jj].x, q[jj].y)), fmaxf(q[jj].z, q[jj].w));
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float e = expf(q[jj][s] - m);
            q[jj][s] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float* cbase = ctx + ((long long)img * heads + head) * ctx_stride + l31;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a = (q[jj][s] / sum) * scale;
            const float b = cbase[(8 * jj + 4 * hh + s) * 32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int nn = tile * 32 + rowmap(r, hh);
        if (nn < N) out[(img * N + nn) * (long long)HD + head * 32 + l31] = acc[r];
    }
}

int launch_linear_attention(const float* qkv, float* out, int heads, long long images, int N, void* ws,
                            hipStream_t s, int ctx_stride, int save) {
    if (images == 0) return DPC_OK;
    float* ctx = reinterpret_cast<float*>(ws);
    DPC_REQUIRE(images * heads < (1ll << 31), "linear attention: grid too large");
    const double rows_ = (double)images * N;
    ProfScope prof(PROF_LINATTN, 4.0 * rows_ * 32 * 32 * heads, 4.0 * rows_ * heads * 32 * 6, s);
    hipLaunchKernelGGL(linattn_ctx_kernel, dim3((unsigned)(images * heads)), dim3(256), 0, s, qkv, ctx, heads, N, ctx_stride, save);
    DPC_LAUNCH_CHECK();
    const int tiles = (N + 31) / 32;
    const long long total = images * tiles * heads;
    hipLaunchKernelGGL(linattn_out_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, qkv, ctx, out, heads, N,
                       total, tiles, ctx_stride);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
