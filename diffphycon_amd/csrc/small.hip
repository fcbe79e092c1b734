// Small host-latency-class ops of the denoiser: time-embedding MLP pieces and layout converters.
#include "common.h"

namespace dpc {

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == 1) return v / (1.0f + expf(-v));                               // SiLU
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // exact GELU (nn.GELU default)
    return v;
}

// out[b][n] = out_act(bias[n] + sum_k in_act(in[b][k]) W[n][k]).  One wave per (output column n, block of BB rows b; BB = 16 for B >= 64): the
// weight row is read once into registers and reused for the 16 rows (one wave per ELEMENT re-read it B times: 172 us for the
// 256 x 256 -> 2048 time-embedding projections of the Burgers net).  Per element the arithmetic is unchanged: lane-strided
// partial sums in ascending k, then the xor butterfly.
// Reference: time_mlp (video_diffusion_pytorch_conv3d.py:404-409), ResnetBlock.mlp (:209-212).
template <int KR>      // K <= 64 * KR
__global__ __launch_bounds__(256) void small_linear_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int B, int K, int N, int in_act, int out_act, int nbb, int BB) {
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)nbb * N) return;
    const int n = (int)(wid % N), b0 = (int)(wid / N) * BB;
    float w[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) {
        const int k = lane + 64 * j;
        w[j] = k < K ? W[(long long)n * K + k] : 0.f;
    }
    const float bn = bias ? bias[n] : 0.f;
    for (int b = b0; b < b0 + BB && b < B; ++b) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + 64 * j;
            if (k < K) s += act_apply(in[(long long)b * K + k], in_act) * w[j];
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[(long long)b * N + n] = act_apply(s + bn, out_act);
    }
}

int launch_small_linear(const float* in, const float* W, const float* bias, float* out, int B, int K, int N,
                        int in_act, int out_act, hipStream_t s) {
    const long long total = (long long)B * N;
    if (total == 0) return DPC_OK;
    DPC_REQUIRE(K <= 2048, "small_linear: K <= 2048");
    ProfScope prof(PROF_SMALL, 2.0 * B * (double)N * K, 4.0 * ((double)N * K + (double)B * (N + K)), s);
    const int bb = B >= 64 ? 16 : 1;            // small batches (the 3-D nets' micro-batches): one wave per element is faster
    const int nbb = (B + bb - 1) / bb;
    const dim3 grid((unsigned)(((long long)nbb * N + 3) / 4)), blk(256);
    if (K <= 256) hipLaunchKernelGGL(small_linear_kernel<4>, grid, blk, 0, s, in, W, bias, out, B, K, N, in_act, out_act, nbb, bb);
    else if (K <= 1024) hipLaunchKernelGGL(small_linear_kernel<16>, grid, blk, 0, s, in, W, bias, out, B, K, N, in_act, out_act, nbb, bb);
    else hipLaunchKernelGGL(small_linear_kernel<32>, grid, blk, 0, s, in, W, bias, out, B, K, N, in_act, out_act, nbb, bb);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// The same op for up to 32 weight sets on ONE input in one launch (blockIdx.y = set): every ResnetBlock's time projection
// mlp = Sequential(SiLU, Linear(dim * 4, 2 * dim_out)) (...conv3d.py:209-212) depends only on the time embedding, so a forward
// computes all of them up front instead of 14 (3-D nets) or 19-23 (Burgers nets) launches scattered through the network.  Per element the arithmetic is
// that of small_linear_kernel (bit-identical results).
template <int KR>
__global__ __launch_bounds__(256) void small_linear_multi_kernel(const float* __restrict__ in, SmallLinearBatch d, int B, int K,
                                                                 int in_act, int out_act, int nbb, int BB) {
    const int set = blockIdx.y;
    const int N = d.N[set];
    const float* __restrict__ W = d.W[set];
    const float* __restrict__ bias = d.bias[set];
    float* __restrict__ out = d.out[set];
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long long)nbb * N) return;
    const int n = (int)(wid % N), b0 = (int)(wid / N) * BB;
    float w[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) {
        const int k = lane + 64 * j;
        w[j] = k < K ? W[(long long)n * K + k] : 0.f;
    }
    const float bn = bias ? bias[n] : 0.f;
    for (int b = b0; b < b0 + BB && b < B; ++b) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + 64 * j;
            if (k < K) s += act_apply(in[(long long)b * K + k], in_act) * w[j];
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) out[(long long)b * N + n] = act_apply(s + bn, out_act);
    }
}

// in_act applied once: the multi kernel evaluated SiLU(in[b][k]) again for every output column (N_total x B x K expf: 1.2 ms of the
// Burgers step at B = 256, r03 trace); the activated input goes to a scratch buffer first, the products see the same floats
__global__ __launch_bounds__(256) void small_act_kernel(const float* __restrict__ in, float* __restrict__ out, long long n, int act) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = act_apply(in[i], act);
}

// r04: the same sets on the fp32 matrix pipe.  The wave-per-column form above spends 6 shuffles and a 4-byte store per output and took
// 0.45 ms for the 19 time projections of the Burgers joint net at B = 256 (12 416 columns x 256 rows x K = 256: 1.6 GFLOP; r04_q trace).
// Here one wave owns a 32-row x 32-column tile of one set: both operands come straight from global memory as 16-byte lane loads
// (lane (l31, hh) holds k = 8 j + 4 hh .. + 3 of its row / column -- the same permutation of the k slots on both operands), four
// v_mfma_f32_32x32x2_f32 per load pair, exact fp32 products, fixed k order: a row's result does not depend on the batch it is in.
struct SmallLinearTiles { int first[33]; int total; };        // first column tile of every set (prefix sums of ceil(N / 32))

__global__ __launch_bounds__(256) void small_linear_multi_mfma_kernel(const float* __restrict__ in, SmallLinearBatch d, SmallLinearTiles tl,
                                                                      int B, int K, int out_act, int rowtiles) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
    const int t = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (t >= tl.total * rowtiles) return;
    const int ctg = t / rowtiles, rt = t - ctg * rowtiles;
    int set = 0;
    while (set + 1 < d.count && tl.first[set + 1] <= ctg) ++set;
    const int N = d.N[set], ct = ctg - tl.first[set];
    const float* __restrict__ W = d.W[set];
    const int row = min(rt * 32 + l31, B - 1), col = min(ct * 32 + l31, N - 1);
    const float* ap = in + (long long)row * K + 4 * hh;
    const float* bp = W + (long long)col * K + 4 * hh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 a = *reinterpret_cast<const f32x4*>(ap), b = *reinterpret_cast<const f32x4*>(bp);
    for (int k = 8; k <= K; k += 8) {
        f32x4 an = a, bn = b;
        if (k < K) {
            an = *reinterpret_cast<const f32x4*>(ap + k);
            bn = *reinterpret_cast<const f32x4*>(bp + k);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        a = an; b = bn;
    }
    const int n = ct * 32 + l31;
    if (n >= N) return;
    const float bias = d.bias[set] ? d.bias[set][n] : 0.f;
    float* __restrict__ out = d.out[set];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rb = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (rb < B) out[(long long)rb * N + n] = act_apply(acc[r] + bias, out_act);
    }
}

int launch_small_linear_multi(const float* in, const SmallLinearBatch& d, int B, int K, int in_act, int out_act, hipStream_t s) {
    if (B == 0 || d.count == 0) return DPC_OK;
    DPC_REQUIRE(d.count <= 32 && K <= 1024, "small_linear_multi: at most 32 sets, K <= 1024");
    int nmax = 0;
    double nsum = 0;
    for (int i = 0; i < d.count; ++i) { nmax = std::max(nmax, d.N[i]); nsum += d.N[i]; }
    ProfScope prof(PROF_SMALL, 2.0 * B * nsum * K, 4.0 * (nsum * K + (double)B * (nsum + K)), s);
    if (in_act != 0) {
        // scratch per (device, stream): two forwards on different streams never share it (api.hip: stream_scratch)
        float* act = nullptr;
        if (int rc = stream_scratch(SCRATCH_SMALL_ACT, s, (size_t)B * K * sizeof(float), &act)) return rc;
        const long long n = (long long)B * K;
        hipLaunchKernelGGL(small_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, act, n, in_act);
        DPC_LAUNCH_CHECK();
        in = act;
        in_act = 0;
    }
    static const int use_mfma = debug_switch("DPC_SMALL_LINEAR_MFMA", 1);
    if (use_mfma && K % 8 == 0 && K >= 8) {        // shape-only rule (never the batch): every set of every net has K = 4 dim
        SmallLinearTiles tl{};
        int tiles = 0;
        for (int i = 0; i < d.count; ++i) { tl.first[i] = tiles; tiles += (d.N[i] + 31) / 32; }
        tl.first[d.count] = tiles;
        tl.total = tiles;
        const int rowtiles = (B + 31) / 32;
        const long long waves = (long long)tiles * rowtiles;
        hipLaunchKernelGGL(small_linear_multi_mfma_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, in, d, tl, B, K, out_act, rowtiles);
        DPC_LAUNCH_CHECK();
        return DPC_OK;
    }
    const int bb = B >= 64 ? 16 : 1;
    const int nbb = (B + bb - 1) / bb;
    const dim3 grid((unsigned)(((long long)nbb * nmax + 3) / 4), (unsigned)d.count), blk(256);
    if (K <= 256) hipLaunchKernelGGL(small_linear_multi_kernel<4>, grid, blk, 0, s, in, d, B, K, in_act, out_act, nbb, bb);
    else hipLaunchKernelGGL(small_linear_multi_kernel<16>, grid, blk, 0, s, in, d, B, K, in_act, out_act, nbb, bb);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// SinusoidalPosEmb (…conv3d.py:144-151): emb = t[:,None]*freqs[None,:]; cat(sin, cos)
__global__ void sinusoidal_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs,
                                  float* __restrict__ out, int B, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half;
    const float e = __fmul_rn((float)t[b], freqs[j]);
    out[(long long)b * 2 * half + j] = sinf(e);
    out[(long long)b * 2 * half + half + j] = cosf(e);
}

// 1x1x1 convolution to a handful of output channels, written per-frame channels-first (the denoiser's final_conv.1,
// video_diffusion_pytorch_conv3d.py:478-481: dim -> channels / out_dim).  As an implicit GEMM it pads 6 (or 2, 4, 1) output channels to a
// 64-wide MFMA tile and ran at 1.2 TB/s; it is a row-streaming dot product: 64 rows per workgroup are staged through LDS with coalesced 16-byte
// loads (row pitch K + 1 floats: conflict-free for the row-per-thread pass), each thread then reduces its row against the N weight
// rows in plain fp32 (ascending k, exact products: no operand split in any arithmetic mode) and writes out[bf][n][hw].
template <int K, int ROWS, int N>      // N compile-time: the weight rows become batched scalar loads, the N x K fused multiply-adds unroll
__global__ __launch_bounds__(ROWS) void conv1x1_rows_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ out, long long M,
                                                           long long HW) {
    __shared__ float tile[ROWS * (K + 1)];
    const int tid = threadIdx.x;
    for (long long r0 = (long long)blockIdx.x * ROWS; r0 < M; r0 += (long long)gridDim.x * ROWS) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < K / 4; ++i) {
            const int idx = tid + ROWS * i, row = idx / (K / 4), q = idx % (K / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r0 + row < M) v = *reinterpret_cast<const f32x4*>(x + (r0 + row) * K + q * 4);
            float* d = tile + row * (K + 1) + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        const long long r = r0 + tid;
        if (r < M) {
            float acc[N];
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = 0.f;
            const float* t = tile + tid * (K + 1);
#pragma unroll 16
            for (int k = 0; k < K; ++k) {
                const float xv = t[k];
#pragma unroll
                for (int n = 0; n < N; ++n) acc[n] = fmaf(xv, W[n * K + k], acc[n]);
            }
            const long long bf = r / HW, hw = r - bf * HW;
#pragma unroll
            for (int n = 0; n < N; ++n) out[(bf * N + n) * HW + hw] = acc[n] + (bias ? bias[n] : 0.f);
        }
    }
}

bool conv1x1_rows_supported(int K, int N) { return K == 64 && N >= 1 && N <= 8; }

int launch_conv1x1_rows(const float* x, const float* W, const float* bias, float* out, long long M, long long HW, int K, int N,
                        hipStream_t s) {
    DPC_REQUIRE(conv1x1_rows_supported(K, N), "conv1x1_rows: K == 64, N <= 8");
    if (M == 0) return DPC_OK;
    ProfScope prof(PROF_IGEMM64, 2.0 * (double)M * K * N, 4.0 * ((double)M * K + (double)M * N), s);
    // 64-row tiles of one wave each (16.6 KB of LDS: nine workgroups per CU overlap each other's load / compute phases; a 256-row
    // tile left two workgroups per CU taking turns: 179 us for 268 MB)
    const int grid = (int)std::min<long long>((M + 63) / 64, 256 * 36);
#define DPC_ROWS_CASE(n) case n: hipLaunchKernelGGL((conv1x1_rows_kernel<64, 64, n>), dim3(grid), dim3(64), 0, s, x, W, bias, out, M, HW); break;
    switch (N) {
        DPC_ROWS_CASE(1) DPC_ROWS_CASE(2) DPC_ROWS_CASE(3) DPC_ROWS_CASE(4) DPC_ROWS_CASE(5) DPC_ROWS_CASE(6) DPC_ROWS_CASE(7) DPC_ROWS_CASE(8)
    }
#undef DPC_ROWS_CASE
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_sinusoidal(const int64_t* t, const float* freqs, float* out, int B, int half, hipStream_t s) {
    const int n = B * half;
    if (n == 0) return DPC_OK;
    hipLaunchKernelGGL(sinusoidal_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, freqs, out, B, half);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// channels-last [BF][HW][C] -> channels-first [B][C][F][HW]   (debug taps / tests only)
__global__ void cl_to_cf_kernel(const float* __restrict__ x, float* __restrict__ y, int BF, int C, long long HW,
                                int F) {
    const long long total = (long long)BF * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long hw = i % HW;
        long long r = i / HW;
        const int f = (int)(r % F);
        r /= F;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        y[i] = x[(((long long)b * F + f) * HW + hw) * C + c];
    }
}
int launch_cl_to_cf(const float* x_cl, float* x_cf, int BF, int C, long long HW, int F, hipStream_t s) {
    const long long total = (long long)BF * HW * C;
    if (total == 0) return DPC_OK;
    hipLaunchKernelGGL(cl_to_cf_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, s,
                       x_cl, x_cf, BF, C, HW, F);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// per-frame channels-first [BF][C][HW] -> channels-last [BF][HW][C]
__global__ void cf_to_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int BF, int C, long long HW) {
    const long long total = (long long)BF * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const long long hw = r % HW;
        const long long bf = r / HW;
        y[i] = x[(bf * C + c) * HW + hw];
    }
}
int launch_cf_to_cl(const float* x_cf, float* x_cl, int BF, int C, long long HW, hipStream_t s) {
    const long long total = (long long)BF * HW * C;
    if (total == 0) return DPC_OK;
    hipLaunchKernelGGL(cf_to_cl_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, s,
                       x_cf, x_cl, BF, C, HW);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// nn.Upsample(scale_factor=2, mode='nearest') on channels-last data (model/burgers_1d/unet.py:40-44)
__global__ __launch_bounds__(256) void upsample2x_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int BF,
                                                            int H, int W, int C) {
    const int c4n = C >> 2;
    const long long total = (long long)BF * 4 * H * W * c4n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        long long r = i / c4n;
        const int wo = (int)(r % (2 * W));
        r /= 2 * W;
        const int ho = (int)(r % (2 * H));
        const long long bf = r / (2 * H);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((bf * H + (ho >> 1)) * W + (wo >> 1)) * C + c4 * 4);
        *reinterpret_cast<f32x4*>(y + ((bf * 2 * H + ho) * 2 * W + wo) * C + c4 * 4) = v;
    }
}

int launch_upsample2x_cl(const float* x, float* y, int BF, int H, int W, int C, hipStream_t s) {
    DPC_REQUIRE(C % 4 == 0, "upsample2x: C % 4");
    const long long total = (long long)BF * 4 * H * W * (C / 4);
    if (total == 0) return DPC_OK;
    ProfScope prof(PROF_SMALL, 0, 4.0 * (double)BF * H * W * C * 5, s);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(upsample2x_cl_kernel, dim3(grid), dim3(256), 0, s, x, y, BF, H, W, C);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ---------------------------------------------------------------- f16x3 activation range check (common.h: RangeCheck)
__global__ __launch_bounds__(256) void range_check_kernel(const float* __restrict__ x, long long n4, int C4,
                                                          const float* __restrict__ in_coef, long long rows_per_sample,
                                                          float limit, int* __restrict__ flag, int id) {
    bool bad = false;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        if (in_coef) {       // the value the halo staging splits: SiLU((GN(x)) * (scale + 1) + shift)  (conv3f3.hip store_halo)
            const long long row = i / C4;
            const int c4 = (int)(i - row * C4);
            const long long b = row / rows_per_sample;
            const f32x4* cf = reinterpret_cast<const f32x4*>(in_coef) + (b * C4 + c4) * 5;
            f32x4 y = (v - cf[0]) * cf[1] + cf[2];
            y = y * cf[3] + cf[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.0f + expf(-y[e]));
            v = y;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) bad |= !(fabsf(v[e]) <= limit);          // also catches NaN / Inf
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicMin(flag, id);      // flag starts at INT_MAX: keeps the FIRST offending op
}

int launch_range_check(const float* x, long long rows, int C, const float* in_coef, long long rows_per_sample, float limit,
                       int* flag, int id, hipStream_t s) {
    if (rows == 0 || C == 0) return DPC_OK;
    DPC_REQUIRE(C % 4 == 0, "range_check: channel count must be a multiple of 4");
    const long long n4 = rows * (C / 4);
    const int grid = (int)std::min<long long>((n4 + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(range_check_kernel, dim3(grid), dim3(256), 0, s, x, n4, C / 4, in_coef, rows_per_sample, limit, flag, id);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
