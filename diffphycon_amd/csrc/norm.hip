// Normalisation kernels on channels-last activations (HBM-bound streaming; float4 per lane).
//   ln_stats       : per-row (mean, 1/sqrt(var+eps)) over C      -> consumed by igemm's A-operand prologue
//   gn_partial     : per (sample, row-chunk, channel) sum / sum of squares, fp64 partials (deterministic)
//   gn_apply_silu  : GroupNorm affine -> (scale+1, shift) -> SiLU, in place
#include "common.h"

namespace dpc {

// ------------------------------------------------------------------ channel LayerNorm statistics
// Reference: LayerNorm.forward (video_diffusion_pytorch_conv3d.py:171-174): biased variance over dim=1.
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                       long long rows, int C, int G /*lanes per row*/) {
    const int lane = threadIdx.x & 63;
    const int rows_per_wave = 64 / G;
    const long long wave_id = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int sub = lane / G, gl = lane % G;
    const int c4n = C >> 2;
    for (long long r0 = wave_id * rows_per_wave; r0 < rows; r0 += nwaves * rows_per_wave) {
        const long long r = r0 + sub;
        const bool ok = r < rows;
        float s = 0.f;
        // first pass: mean
        for (int c4 = gl; c4 < c4n; c4 += G) {
            if (ok) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c4 * 4);
                s += (v.x + v.y) + (v.z + v.w);
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)C;
        float q = 0.f;
        for (int c4 = gl; c4 < c4n; c4 += G) {
            if (ok) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c4 * 4);
                const f32x4 d = v - mean;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if (ok && gl == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = 1.0f / sqrtf(q / (float)C + 1e-5f);
        }
    }
}

int launch_ln_stats(const float* x, float* stats, long long rows, int C, hipStream_t s) {
    DPC_REQUIRE(C % 4 == 0, "ln_stats: C % 4");
    int G = 1;
    while (G < 64 && G < C / 4) G <<= 1;   // power-of-two lanes per row
    const int rows_per_block = 4 * (64 / G);
    ProfScope prof(PROF_LN, 0, 4.0 * (double)rows * (C + 2), s);
    const long long nb = (rows + rows_per_block - 1) / rows_per_block;
    const int grid = (int)std::min<long long>(nb, 256 * 16);
    if (grid == 0) return DPC_OK;
    hipLaunchKernelGGL(ln_stats_kernel, dim3(grid), dim3(256), 0, s, x, stats, rows, C, G);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// Channel LayerNorm applied from precomputed (mean, rstd) + residual: the trailing LayerNorm of the 2-D LinearAttention
// (model/burgers_1d/unet.py:59-69, 203-206) followed by Residual (:22-28).
__global__ __launch_bounds__(256) void ln_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* resid,
                                                       float* out, long long rows, int C) {
    const int c4n = C >> 2;
    const long long total = rows * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        float mean = stats[2 * r], rstd = stats[2 * r + 1];
        // r02: with the (mean, rstd) pair used straight from its dwordx2 load the compiler folds the broadcast of rstd into
        // `v_pk_mul_f32 ..., v[m:m+1] op_sel:[0,1]`, and that form gave wrong LOW lanes (components 0 / 2 of a row's float4s;
        // results like (v - mean) * 0) in ~5 % of launches whenever a second process kept the GPU busy -- never on an idle GPU,
        // and never once rstd sits in a register of its own (tools/det_ops.py: 0 of 300 launches vs 11 of 200).  r04 found the
        // same failure in every packed fp32 op of the library whenever a second kernel is resident and builds
        // without them (-packed-fp32-ops); the two copies stay as they are (one VALU op per float4).
        asm volatile("v_mov_b32 %0, %0" : "+v"(rstd));
        asm volatile("v_mov_b32 %0, %0" : "+v"(mean));
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        f32x4 o = (v - mean) * rstd * g;
        if (resid) o = o + *reinterpret_cast<const f32x4*>(resid + r * C + c);
        *reinterpret_cast<f32x4*>(out + r * C + c) = o;
    }
}

int launch_ln_apply(const float* x, const float* stats, const float* gamma, const float* resid, float* out,
                    long long rows, int C, hipStream_t s) {
    DPC_REQUIRE(C % 4 == 0, "ln_apply: C % 4");
    if (rows == 0) return DPC_OK;
    ProfScope prof(PROF_LN, 0, 4.0 * (double)rows * C * (resid ? 3 : 2), s);
    const long long total = rows * (C / 4);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(ln_apply_kernel, dim3(grid), dim3(256), 0, s, x, stats, gamma, resid, out, rows, C);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------ GroupNorm + scale/shift + SiLU
// Reference: Block.forward (…conv3d.py:196-204) with nn.GroupNorm(groups, C, eps=1e-5).
constexpr int GN_MAX_CHUNKS = 128;

static int gn_chunks(long long R, int C) {
    const int tpr = C / 4;                 // threads per row
    const int rows_per_pass = 256 / tpr;
    long long n = R / ((long long)rows_per_pass * 32);
    if (n < 1) n = 1;
    if (n > GN_MAX_CHUNKS) n = GN_MAX_CHUNKS;
    return (int)n;
}

size_t gn_workspace_bytes(int B, int C) {
    // fp64 partial sums [B][GN_MAX_CHUNKS][C][2] + finished (mean, rstd) [B][groups<=C][2] floats
    return (size_t)B * GN_MAX_CHUNKS * C * 2 * sizeof(double) + (size_t)B * C * 2 * sizeof(float);
}

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, double* __restrict__ part,
                                                         long long R, int C, int nchunk) {
    __shared__ double red[256][8];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int tpr = C >> 2, rpp = 256 / tpr;
    const int c4 = tid % tpr, rsub = tid / tpr;
    const long long rpc = (R + nchunk - 1) / nchunk;
    const long long r_begin = chunk * rpc, r_end = min(R, r_begin + rpc);
    const float* xb = x + (long long)b * R * C;
    f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    for (long long r = r_begin + rsub; r < r_end; r += rpp) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + r * C + c4 * 4);
        s += v;
        q += v * v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[tid][i] = (double)s[i];
        red[tid][4 + i] = (double)q[i];
    }
    __syncthreads();
    if (tid < tpr) {
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < rpp; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += red[tid + k * tpr][i];
        double* dst = part + (((long long)b * nchunk + chunk) * C + tid * 4) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[2 * i] = acc[i];
            dst[2 * i + 1] = acc[4 + i];
        }
    }
}

// one wave per (sample, group): fold the fp64 partials into (mean, rstd)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part, float* __restrict__ stats,
                                                          long long R, int C, int groups, int nchunk, int B) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= B * groups) return;
    const int b = wid / groups, g = wid % groups;
    const int cpg = C / groups;
    double s = 0, q = 0;
    for (int i = lane; i < nchunk * cpg; i += 64) {
        const int k = i / cpg, c = i % cpg;
        const double* src = part + (((long long)b * nchunk + k) * C + g * cpg + c) * 2;
        s += src[0];
        q += src[1];
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) {
        const double n = (double)R * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        stats[2 * wid] = (float)mean;
        stats[2 * wid + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// rows [r_begin, r_end) of sample b: out = SiLU((x - mean_g) * rstd_g * gamma + beta) * (scale + 1) + shift) (+ resid); shared by the
// streaming apply kernel and the one-pass kernel below (identical per-element arithmetic)
__device__ __forceinline__ void gn_apply_rows(const float* x, float* out, const float* resid, const float* s_mean, const float* s_rstd,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const float* __restrict__ scale_shift, long long R, int C, int groups, int b,
                                              long long r_begin, long long r_end, int* oflag) {
    const int tid = threadIdx.x;
    const int cpg = C / groups;
    const int tpr = C >> 2, rpp = 256 / tpr;
    const int c4 = tid % tpr, rsub = tid / tpr;
    float mu[4], ga[4], be[4], sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c4 * 4 + i, g = c / cpg;
        mu[i] = s_mean[g];
        ga[i] = s_rstd[g] * gamma[c];
        be[i] = beta[c];
        sc[i] = scale_shift ? scale_shift[(long long)b * 2 * C + c] + 1.0f : 1.0f;
        sh[i] = scale_shift ? scale_shift[(long long)b * 2 * C + C + c] : 0.0f;
    }
    const float* xb = x + (long long)b * R * C;
    float* ob = out + (long long)b * R * C;
    const float* rb = resid ? resid + (long long)b * R * C : nullptr;
    unsigned omx = 0;
    // four rows per thread and iteration, all loads issued before the first use (rows past the end re-read the first row and are
    // not stored); r03: no gain on the 64 x 64 level -- x + residual + out already move at 5.6 TB/s
    for (long long r = r_begin + rsub; r < r_end; r += 4 * rpp) {
        f32x4 v[4], q[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long ru = r + (long long)u * rpp;
            ok[u] = ru < r_end;
            const long long o = (ok[u] ? ru : r) * C + c4 * 4;
            // (non-temporal: every byte of this pass is touched once -- 12.0 -> 11.3 ms per S64 step for the class, r03_bs)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xb + o));
            if (rb) q[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rb + o));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = (v[u][i] - mu[i]) * ga[i] + be[i];
                if (scale_shift) y = y * sc[i] + sh[i];
                v[u][i] = y / (1.0f + expf(-y));
            }
            if (rb) v[u] += q[u];
            if (ok[u]) {
                omx = max(omx, max(max(abs_bits(v[u][0]), abs_bits(v[u][1])), max(abs_bits(v[u][2]), abs_bits(v[u][3]))));
                __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(ob + (r + (long long)u * rpp) * C + c4 * 4));
            }
        }
    }
    if (oflag && omx > F16X3_ACT_LIMIT_BITS) atomicOr(oflag, 1);     // f16x3 activation-range sentinel (common.h)
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* x, float* out, const float* resid,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ scale_shift, long long R, int C,
                                                       int groups, int nblk, int* oflag) {
    __shared__ float s_mean[1024], s_rstd[1024];   // per group (groups <= 1024)
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int g = tid; g < groups; g += 256) {
        s_mean[g] = stats[2 * (b * groups + g)];
        s_rstd[g] = stats[2 * (b * groups + g) + 1];
    }
    __syncthreads();
    const long long rpb = (R + nblk - 1) / nblk;
    const long long r_begin = blockIdx.x * rpb, r_end = min(R, r_begin + rpb);
    gn_apply_rows(x, out, resid, s_mean, s_rstd, gamma, beta, scale_shift, R, C, groups, b, r_begin, r_end, oflag);
}

// One launch per GroupNorm for samples whose statistics take ONE row chunk (gn_chunks == 1: the 4 x 32, 2 x 16, 1 x 8 levels of the
// Burgers U-Net -- 64 of its 84 GroupNorms per step ran as three launches of 8 + 5 + 13-18 us each).  One workgroup per sample does
// what gn_partial_kernel (one chunk), gn_finalize_kernel and gn_apply_kernel do, with the SAME thread mapping, fp32 row sums, fp64
// folds and butterfly order: the statistics and the outputs are bit-identical to the three-launch path (checked A/B: DPC_GN_ONEPASS=0).
// The second read of x comes from L2 (<= 128 KB per sample).
__global__ __launch_bounds__(256) void gn_onepass_kernel(const float* x, float* out, const float* resid,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ scale_shift, long long R, int C, int groups,
                                                         int* oflag) {
    __shared__ double red[256][8];
    __shared__ double chs[1024][2];                // per channel (sum, sum of squares)
    __shared__ float s_mean[1024], s_rstd[1024];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tpr = C >> 2, rpp = 256 / tpr;
    const int c4 = tid % tpr, rsub = tid / tpr;
    const float* xb = x + (long long)b * R * C;
    {   // gn_partial_kernel, chunk 0 of 1
        f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
        for (long long r = rsub; r < R; r += 4 * rpp) {           // four loads in flight, accumulated in row order (same sums)
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long ru = r + (long long)u * rpp;
                v[u] = *reinterpret_cast<const f32x4*>(xb + (ru < R ? ru : r) * C + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + (long long)u * rpp < R) {
                    s += v[u];
                    q += v[u] * v[u];
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[tid][i] = (double)s[i];
            red[tid][4 + i] = (double)q[i];
        }
        __syncthreads();
        if (tid < tpr) {
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; k < rpp; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += red[tid + k * tpr][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                chs[tid * 4 + i][0] = acc[i];
                chs[tid * 4 + i][1] = acc[4 + i];
            }
        }
        __syncthreads();
    }
    {   // gn_finalize_kernel: one wave per group, lane-strided fp64 sums + xor butterfly
        const int cpg = C / groups;
        for (int g = wave; g < groups; g += 4) {
            double s = 0, q = 0;
            for (int i = lane; i < cpg; i += 64) {
                s += chs[g * cpg + i][0];
                q += chs[g * cpg + i][1];
            }
            for (int o = 32; o > 0; o >>= 1) {
                s += __shfl_xor(s, o, 64);
                q += __shfl_xor(q, o, 64);
            }
            if (lane == 0) {
                const double n = (double)R * cpg;
                const double mean = s / n;
                double var = q / n - mean * mean;
                if (var < 0) var = 0;
                s_mean[g] = (float)mean;
                s_rstd[g] = (float)(1.0 / sqrt(var + 1e-5));
            }
        }
        __syncthreads();
    }
    gn_apply_rows(x, out, resid, s_mean, s_rstd, gamma, beta, scale_shift, R, C, groups, b, 0, R, oflag);
}

// ---- fused-statistics path: the conv3x6 epilogue already produced per-tile channel sums (Conv3hParams::gn_part)
// one wave per (sample, group): fold [tiles][2][C][2] fp32 partials in fp64 (fixed order), emit (mean, rstd) and the
// per-channel coefficient table [B][C/4][5][4] = (mu, rstd*gamma, beta, scale+1, shift) for the consumer's halo load, followed by
// its folded form [B][C/4][2][4] (coef must hold B * C * 7 floats)
__global__ __launch_bounds__(1024) void gn_finalize_fused_kernel(const float* __restrict__ part, float* __restrict__ stats,
                                                                float* __restrict__ coef, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                const float* __restrict__ scale_shift, long long nslab, int C,
                                                                int groups, long long R, int B) {
    // one WORKGROUP per (sample, group): 1024 threads stream the partials with 4 independent accumulators each (the kernel
    // sits on the conv1 -> conv2 dependency chain and is pure load latency: one wave per group took 48 us, 256 threads 11 us)
    __shared__ double red_s[1024], red_q[1024];
    const int tid = threadIdx.x;
    const int wid = blockIdx.x;
    const int b = wid / groups, g = wid % groups;
    const int cpg = C / groups;
    const long long total = nslab * cpg;
    const float* base = part + ((long long)b * nslab * C + (long long)g * cpg) * 2;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    auto at = [&](long long i) -> const float* {
        const long long k = i / cpg;
        const int c = (int)(i - k * cpg);
        return base + (k * C + c) * 2;
    };
    long long i = tid;
    for (; i + 3072 < total; i += 4096) {
        const float *p0 = at(i), *p1 = at(i + 1024), *p2 = at(i + 2048), *p3 = at(i + 3072);
        const float a0 = p0[0], b0 = p0[1], a1 = p1[0], b1 = p1[1], a2 = p2[0], b2 = p2[1], a3 = p3[0], b3 = p3[1];
        s0 += (double)a0; q0 += (double)b0; s1 += (double)a1; q1 += (double)b1;
        s2 += (double)a2; q2 += (double)b2; s3 += (double)a3; q3 += (double)b3;
    }
    for (; i < total; i += 1024) {
        const float* p0 = at(i);
        s0 += (double)p0[0];
        q0 += (double)p0[1];
    }
    red_s[tid] = (s0 + s1) + (s2 + s3);
    red_q[tid] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) { red_s[tid] += red_s[tid + o]; red_q[tid] += red_q[tid + o]; }
        __syncthreads();
    }
    const double n = (double)R * cpg;
    const double mean = red_s[0] / n;
    double var = red_q[0] / n - mean * mean;
    if (var < 0) var = 0;
    const float mu = (float)mean, rstd = (float)(1.0 / sqrt(var + 1e-5));
    if (tid == 0) {
        stats[2 * wid] = mu;
        stats[2 * wid + 1] = rstd;
    }
    if (coef) {
        for (int c = tid; c < cpg; c += 1024) {
            const int ch = g * cpg + c;
            float* dst = coef + ((long long)b * (C >> 2) + (ch >> 2)) * 20 + (ch & 3);
            dst[0] = mu;
            dst[4] = rstd * gamma[ch];
            dst[8] = beta[ch];
            dst[12] = scale_shift ? scale_shift[(long long)b * 2 * C + ch] + 1.0f : 1.0f;
            dst[16] = scale_shift ? scale_shift[(long long)b * 2 * C + C + ch] : 0.0f;
            // folded form behind the table, [B][C/4][2][4] at coef + B C 5, for the Winograd conv's loader (conv3w.hip):
            // GroupNorm -> x (scale + 1) + shift as ONE multiply-add y = x A + B, both times log2(e) (the exponent of its SiLU)
            const double ga = (double)dst[4], sc = (double)dst[12];
            float* d2 = coef + (long long)B * C * 5 + ((long long)b * (C >> 2) + (ch >> 2)) * 8 + (ch & 3);
            d2[0] = (float)(ga * sc * 1.4426950408889634);
            d2[4] = (float)((((double)dst[8] - (double)mu * ga) * sc + (double)dst[16]) * 1.4426950408889634);
        }
    }
}

int launch_gn_finalize_fused(const float* part, int B, long long tiles, int C, int groups, long long R,
                             const float* gamma, const float* beta, const float* scale_shift, float* stats, float* coef,
                             hipStream_t s, long long entries) {
    DPC_REQUIRE(groups >= 1 && C % groups == 0 && C % 4 == 0, "gn_finalize_fused: groups must divide C, C % 4 == 0");
    if (B == 0) return DPC_OK;
    const long long nslab = entries > 0 ? entries : tiles * 2;        // partial-sum entries per sample
    ProfScope prof(PROF_GN, 0, 4.0 * (double)B * nslab * C * 2, s);
    hipLaunchKernelGGL(gn_finalize_fused_kernel, dim3(B * groups), dim3(1024), 0, s, part, stats, coef, gamma, beta,
                       scale_shift, nslab, C, groups, R, B);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_gn_apply(const float* x, float* out, const float* resid, const float* stats, const float* gamma,
                    const float* beta, const float* scale_shift, int B, long long R, int C, int groups, hipStream_t s) {
    DPC_REQUIRE(C % 4 == 0 && C <= 1024 && (256 % (C / 4)) == 0, "groupnorm: C/4 must divide 256");
    if (B == 0 || R == 0) return DPC_OK;
    ProfScope prof(PROF_GN, 0, 4.0 * (double)B * R * C * (resid ? 3 : 2), s);
    const int rpp = 256 / (C / 4);
    long long nblk = R / ((long long)rpp * 8);
    if (nblk < 1) nblk = 1;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((int)nblk, B), dim3(256), 0, s, x, out, resid, stats, gamma, beta, scale_shift, R,
                       C, groups, (int)nblk, overflow_flag_current());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_groupnorm_silu(const float* x, float* out, const float* resid, const float* gamma, const float* beta,
                          const float* scale_shift, int B, long long R, int C, int groups, void* ws, hipStream_t s) {
    DPC_REQUIRE(C % 4 == 0 && C <= 1024 && (256 % (C / 4)) == 0, "groupnorm: C/4 must divide 256");
    DPC_REQUIRE(groups >= 1 && C % groups == 0 && groups <= 1024, "groupnorm: groups must divide C");
    if (B == 0 || R == 0) return DPC_OK;
    const int nchunk = gn_chunks(R, C);
    ProfScope prof(PROF_GN, 0, 4.0 * (double)B * R * C * (resid ? 4 : 3), s);
    static const int onepass = debug_switch("DPC_GN_ONEPASS", 1);
    if (onepass && nchunk == 1 && out != nullptr) {       // shape-only rule; bit-identical to the three launches below either way
        hipLaunchKernelGGL(gn_onepass_kernel, dim3(B), dim3(256), 0, s, x, out, resid, gamma, beta, scale_shift, R, C, groups,
                           overflow_flag_current());
        DPC_LAUNCH_CHECK();
        return DPC_OK;
    }
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, part, R, C, nchunk);
    DPC_LAUNCH_CHECK();
    const int rpp = 256 / (C / 4);
    long long nblk = R / ((long long)rpp * 8);
    if (nblk < 1) nblk = 1;
    if (nblk > 1024) nblk = 1024;
    float* stats = reinterpret_cast<float*>(part + (size_t)B * GN_MAX_CHUNKS * C * 2);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * groups + 3) / 4), dim3(256), 0, s, part, stats, R, C, groups, nchunk,
                       B);
    DPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_apply_kernel, dim3((int)nblk, B), dim3(256), 0, s, x, out, resid, stats, gamma, beta,
                       scale_shift, R, C, groups, (int)nblk, overflow_flag_current());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}


// ------------------------------------------------------------------ GroupNorm statistics only / GroupNorm+SiLU backward
// (the jellyfish guidance surrogates: forward keeps the raw conv output + statistics on the tape, the backward pass
//  recomputes the activation; diffusion_2d_jellyfish.py:122-148 Block / ResnetBlock)
int launch_gn_stats(const float* x, float* stats, int B, long long R, int C, int groups, void* ws, hipStream_t s) {
    DPC_REQUIRE(C % 4 == 0 && C <= 1024 && (256 % (C / 4)) == 0, "groupnorm: C/4 must divide 256");
    DPC_REQUIRE(groups >= 1 && C % groups == 0 && groups <= 1024, "groupnorm: groups must divide C");
    if (B == 0 || R == 0) return DPC_OK;
    const int nchunk = gn_chunks(R, C);
    ProfScope prof(PROF_GN, 0, 4.0 * (double)B * R * C, s);
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, part, R, C, nchunk);
    DPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * groups + 3) / 4), dim3(256), 0, s, part, stats, R, C, groups, nchunk, B);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// y = silu(z) (+ resid), z = (xhat * gamma + beta) * (scale + 1) + shift, xhat = (x - mean_g) * rstd_g.  Given dy:
//   dz = dy * silu'(z);  per (sample, channel): A = sum_r dz, Bs = sum_r dz * xhat     (pass 1, fp64 partials per row chunk)
//   dshift = A, dscale = gamma * Bs + beta * A;  with dxhat = dz * (scale + 1) * gamma and the group means
//   M1 = mean_g(dxhat), M2 = mean_g(dxhat * xhat):  dx = rstd_g * (dxhat - M1 - xhat * M2)                    (pass 2)
__device__ __forceinline__ float silu_grad(float z) {
    const float sg = 1.0f / (1.0f + expf(-z));
    return sg * (1.0f + z * (1.0f - sg));
}

__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ scale_shift,
                                                            double* __restrict__ part, long long R, int C, int groups, int nchunk) {
    __shared__ double red[256][8];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int cpg = C / groups, tpr = C >> 2, rpp = 256 / tpr;
    const int c4 = tid % tpr, rsub = tid / tpr;
    float mu[4], rs[4], ga[4], be[4], sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c4 * 4 + i, g = c / cpg;
        mu[i] = stats[2 * (b * groups + g)];
        rs[i] = stats[2 * (b * groups + g) + 1];
        ga[i] = gamma[c];
        be[i] = beta[c];
        sc[i] = scale_shift ? scale_shift[(long long)b * 2 * C + c] + 1.0f : 1.0f;
        sh[i] = scale_shift ? scale_shift[(long long)b * 2 * C + C + c] : 0.0f;
    }
    const float* xb = x + (long long)b * R * C;
    const float* db = dy + (long long)b * R * C;
    const long long rpc = (R + nchunk - 1) / nchunk;
    const long long r_begin = chunk * rpc, r_end = min(R, r_begin + rpc);
    double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (long long r = r_begin + rsub; r < r_end; r += rpp) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + r * C + c4 * 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(db + r * C + c4 * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (v[i] - mu[i]) * rs[i];
            const float z = (xh * ga[i] + be[i]) * sc[i] + sh[i];
            const float dz = d[i] * silu_grad(z);
            a[i] += (double)dz;
            q[i] += (double)(dz * xh);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[tid][i] = a[i]; red[tid][4 + i] = q[i]; }
    __syncthreads();
    if (rsub == 0) {                         // fixed order over the row sub-groups: deterministic
        double ta[4] = {0, 0, 0, 0}, tq[4] = {0, 0, 0, 0};
        for (int k = 0; k < rpp; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) { ta[i] += red[k * tpr + c4][i]; tq[i] += red[k * tpr + c4][4 + i]; }
        double* dst = part + (((long long)b * nchunk + chunk) * C + c4 * 4) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dst[2 * i] = ta[i]; dst[2 * i + 1] = tq[i]; }
    }
}

// one wave per (sample, group): channel sums over the chunks, d(scale, shift), group means -> coef [B][C][4] = (k, M1, M2, 0)
// with k = (scale + 1) * gamma so that pass 2 is dx = rstd * (dz * k - M1 - xhat * M2)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ part, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ scale_shift,
                                                             float* __restrict__ coef, float* __restrict__ dss, long long R, int C,
                                                             int groups, int nchunk, int B) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= B * groups) return;
    const int b = wid / groups, g = wid % groups;
    const int cpg = C / groups;
    double m1 = 0, m2 = 0;
    for (int j = lane; j < cpg; j += 64) {
        const int c = g * cpg + j;
        double A = 0, Bs = 0;
        for (int k = 0; k < nchunk; ++k) {
            const double* src = part + (((long long)b * nchunk + k) * C + c) * 2;
            A += src[0];
            Bs += src[1];
        }
        const double sc = scale_shift ? (double)scale_shift[(long long)b * 2 * C + c] + 1.0 : 1.0;
        const double kk = sc * (double)gamma[c];
        m1 += kk * A;
        m2 += kk * Bs;
        if (dss) {
            dss[(long long)b * 2 * C + c] = (float)((double)gamma[c] * Bs + (double)beta[c] * A);     // d scale
            dss[(long long)b * 2 * C + C + c] = (float)A;                                             // d shift
        }
        coef[((long long)b * C + c) * 4] = (float)kk;
    }
    for (int o = 32; o > 0; o >>= 1) {
        m1 += __shfl_xor(m1, o, 64);
        m2 += __shfl_xor(m2, o, 64);
    }
    const double n = (double)R * cpg;
    for (int j = lane; j < cpg; j += 64) {
        const int c = g * cpg + j;
        coef[((long long)b * C + c) * 4 + 1] = (float)(m1 / n);
        coef[((long long)b * C + c) * 4 + 2] = (float)(m2 / n);
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ scale_shift,
                                                          const float* __restrict__ coef, float* __restrict__ dx, long long R, int C,
                                                          int groups, int nblk) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cpg = C / groups, tpr = C >> 2, rpp = 256 / tpr;
    const int c4 = tid % tpr, rsub = tid / tpr;
    float mu[4], rs[4], ga[4], be[4], sc[4], sh[4], kk[4], m1[4], m2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c4 * 4 + i, g = c / cpg;
        mu[i] = stats[2 * (b * groups + g)];
        rs[i] = stats[2 * (b * groups + g) + 1];
        ga[i] = gamma[c];
        be[i] = beta[c];
        sc[i] = scale_shift ? scale_shift[(long long)b * 2 * C + c] + 1.0f : 1.0f;
        sh[i] = scale_shift ? scale_shift[(long long)b * 2 * C + C + c] : 0.0f;
        const float* cf = coef + ((long long)b * C + c) * 4;
        kk[i] = cf[0]; m1[i] = cf[1]; m2[i] = cf[2];
    }
    const float* xb = x + (long long)b * R * C;
    const float* db = dy + (long long)b * R * C;
    float* ob = dx + (long long)b * R * C;
    const long long rpb = (R + nblk - 1) / nblk;
    const long long r_begin = blockIdx.x * rpb, r_end = min(R, r_begin + rpb);
    for (long long r = r_begin + rsub; r < r_end; r += rpp) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + r * C + c4 * 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(db + r * C + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float xh = (v[i] - mu[i]) * rs[i];
            const float z = (xh * ga[i] + be[i]) * sc[i] + sh[i];
            const float dz = d[i] * silu_grad(z);
            o[i] = rs[i] * (dz * kk[i] - m1[i] - xh * m2[i]);
        }
        *reinterpret_cast<f32x4*>(ob + r * C + c4 * 4) = o;
    }
}

size_t gn_bwd_workspace_bytes(int B, int C) {
    return (size_t)B * GN_MAX_CHUNKS * C * 2 * sizeof(double) + (size_t)B * C * 4 * sizeof(float);
}

int launch_gn_silu_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                       const float* scale_shift, float* dx, float* dss, int B, long long R, int C, int groups, void* ws,
                       hipStream_t s, float* dgamma, float* dbeta) {
    DPC_REQUIRE(C % 4 == 0 && C <= 1024 && (256 % (C / 4)) == 0, "groupnorm bwd: C/4 must divide 256");
    DPC_REQUIRE(groups >= 1 && C % groups == 0 && groups <= 1024, "groupnorm bwd: groups must divide C");
    if (B == 0 || R == 0) return DPC_OK;
    const int nchunk = gn_chunks(R, C);
    ProfScope prof(PROF_GN, 0, 4.0 * (double)B * R * C * 5, s);
    double* part = reinterpret_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(part + (size_t)B * GN_MAX_CHUNKS * C * 2);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, B), dim3(256), 0, s, x, dy, stats, gamma, beta, scale_shift, part, R, C,
                       groups, nchunk);
    DPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((B * groups + 3) / 4), dim3(256), 0, s, part, gamma, beta, scale_shift, coef,
                       dss, R, C, groups, nchunk, B);
    DPC_LAUNCH_CHECK();
    if (dgamma || dbeta) {
        DPC_REQUIRE(dgamma && dbeta, "groupnorm bwd: dgamma and dbeta go together");
        if (int rc = launch_gn_param_grad(part, scale_shift, dgamma, dbeta, B, C, nchunk, s)) return rc;
    }
    const int rpp = 256 / (C / 4);
    long long nblk = R / ((long long)rpp * 8);
    if (nblk < 1) nblk = 1;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((int)nblk, B), dim3(256), 0, s, x, dy, stats, gamma, beta, scale_shift, coef, dx, R,
                       C, groups, (int)nblk);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// Channel LayerNorm backward (diffusion_2d_jellyfish.py LayerNorm: y = (x - mean) * rstd * g over the channel axis), G lanes per
// row as ln_stats:  dx = rstd * (dy g - mean_c(dy g) - xhat mean_c(dy g xhat));  accum != 0: dx += (gradient accumulation)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                    const float* __restrict__ g, const float* __restrict__ dy, float* dx,
                                                    long long rows, int C, int G, int accum) {
    const int lane = threadIdx.x & 63;
    const int rows_per_wave = 64 / G;
    const long long wave_id = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int sub = lane / G, gl = lane % G;
    const int c4n = C >> 2;
    for (long long r0 = wave_id * rows_per_wave; r0 < rows; r0 += nwaves * rows_per_wave) {
        const long long r = r0 + sub;
        const bool ok = r < rows;
        const float mean = ok ? stats[2 * r] : 0.f, rstd = ok ? stats[2 * r + 1] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        for (int c4 = gl; c4 < c4n; c4 += G) {
            if (ok) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c4 * 4);
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * C + c4 * 4) * *reinterpret_cast<const f32x4*>(g + c4 * 4);
                const f32x4 xh = (v - mean) * rstd;
                s1 += (d.x + d.y) + (d.z + d.w);
                s2 += (d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w);
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        s1 /= (float)C;
        s2 /= (float)C;
        for (int c4 = gl; c4 < c4n; c4 += G) {
            if (ok) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c4 * 4);
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * C + c4 * 4) * *reinterpret_cast<const f32x4*>(g + c4 * 4);
                const f32x4 xh = (v - mean) * rstd;
                f32x4 o = (d - s1 - xh * s2) * rstd;
                if (accum) o += *reinterpret_cast<const f32x4*>(dx + r * C + c4 * 4);
                *reinterpret_cast<f32x4*>(dx + r * C + c4 * 4) = o;
            }
        }
    }
}

int launch_ln_bwd(const float* x, const float* stats, const float* g, const float* dy, float* dx, long long rows, int C, int accum,
                  hipStream_t s) {
    DPC_REQUIRE(C % 4 == 0, "ln_bwd: C % 4");
    int G = 1;
    while (G < 64 && G < C / 4) G <<= 1;
    const int rows_per_block = 4 * (64 / G);
    ProfScope prof(PROF_LN, 0, 4.0 * (double)rows * C * 4, s);
    const long long nb = (rows + rows_per_block - 1) / rows_per_block;
    const int grid = (int)std::min<long long>(nb, 256 * 16);
    if (grid == 0) return DPC_OK;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid), dim3(256), 0, s, x, stats, g, dy, dx, rows, C, G, accum);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
