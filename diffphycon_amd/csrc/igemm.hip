// fp32 implicit-GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TF peak).
//
// One kernel family serves every GEMM-shaped op of the denoiser: Conv3d 3x3x3, the strided
// (1,4,4) down-conv, the four parity classes of ConvTranspose3d (1,4,4), 1x1x1 convs and the
// attention projections (ntaps == 1), all on channels-last activations, so the reduction
// (channel) axis is contiguous in HBM and a tap is just a row offset.
//
// Tiling: 256 threads = 4 waves (2x2), block tile BM=128 rows x BN={64,128} cols x BK=32.
// LDS tiles are [rows][36] floats (stride 36 words => ds_read_b128 conflict-free for any 16 rows
// distinct mod 16).  Each lane reads float4 = 4 consecutive k; the two half-waves
// take k = 8j+0..3 / 8j+4..7, which is just a permutation of the MFMA k-slots shared by A and B.
// Global->LDS goes through registers (prefetch of tile i+1 is issued before the MFMAs of tile i).
#include "common.h"

namespace dpc {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_STRIDE = 36;

int igemm_npad(int N) { return N <= 64 ? 64 : (int)align_up(N, 128); }
int igemm_kchunks(int K) { return (K + BK - 1) / BK; }

__device__ __forceinline__ int xcd_remap(int id, int nblocks) {
    // give each XCD (block id % 8) a contiguous range of tiles so neighbouring tiles share an L2
    const int q = nblocks >> 3, r = nblocks & 7, xcd = id & 7, idx = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int BN>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Bs,
                                         f32x16 (&acc)[2][BN / 64], int wm, int wn, int l31, int hh) {
    constexpr int NT = BN / 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 a[2], b[NT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            a[mt] = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + mt * 32 + l31) * LDS_STRIDE + 8 * j + 4 * hh]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            b[nt] = *reinterpret_cast<const f32x4*>(&Bs[(wn * (BN / 2) + nt * 32 + l31) * LDS_STRIDE + 8 * j + 4 * hh]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], b[nt][s], acc[mt][nt], 0, 0, 0);
    }
}

template <int BN>
__device__ __forceinline__ void epilogue(const f32x16 (&acc)[2][BN / 64], long long m0, int n0, int wm, int wn,
                                         int l31, int hh, long long M, int N, const float* bias,
                                         const float* resid, float* out, int out_mode, int HoWo, int Wo, int par_a,
                                         int par_b) {
    constexpr int NT = BN / 64;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 32 + l31;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (m >= M) continue;
                float v = acc[mt][nt][r] + bv;
                if (resid) v += resid[m * N + n];
                long long o;
                if (out_mode == 0) {
                    o = m * N + n;
                } else if (out_mode == 1) {
                    const long long bf = m / HoWo, hw = m - bf * HoWo;
                    o = (bf * N + n) * (long long)HoWo + hw;
                } else {
                    const long long bf = m / HoWo;
                    const int hw = (int)(m - bf * HoWo), ho = hw / Wo, wo = hw - ho * Wo;
                    o = ((bf * (2 * (HoWo / Wo)) + 2 * ho + par_a) * (long long)(2 * Wo) + 2 * wo + par_b) * N + n;
                }
                out[o] = v;
            }
        }
    }
}

template <int BN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmParams p) {
    constexpr int NT = BN / 64;
    constexpr int BROWS = BN / 32;   // B rows per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As0 = smem;
    float* Bs0 = As0 + BM * LDS_STRIDE;
    float* As1 = Bs0 + BN * LDS_STRIDE;
    float* Bs1 = As1 + BM * LDS_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    const int bid = xcd_remap(blockIdx.x, mtiles * ntn);
    const long long m0 = (long long)(bid / ntn) * BM;
    const int n0 = (bid % ntn) * BN;

    // ---- per-thread A rows: 4 rows (tid/8 + 32 i), one float4 column (tid%8)*4
    const int arow = tid >> 3, acol = (tid & 7) * 4;
    const int HoWo = p.Ho * p.Wo;
    int r_bf[4], r_f[4], r_h[4], r_w[4];
    bool r_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = m0 + arow + 32 * i;
        r_ok[i] = m < p.M;
        const long long mm = r_ok[i] ? m : 0;
        const int bf = (int)(mm / HoWo);
        const int hw = (int)(mm - (long long)bf * HoWo);
        const int ho = hw / p.Wo;
        r_bf[i] = bf;
        r_f[i] = bf % p.F;
        r_h[i] = ho * p.sh;
        r_w[i] = (hw - ho * p.Wo) * p.sw;
    }
    const int K = p.C0 + p.C1;

    f32x4 ra[4], rb[BROWS];
    long long roff[4];
    bool rvalid[4];
    int cur_tap = -1;

    auto load_tiles = [&](int it) {
        const int tap = it / p.kchunks, kc = it - tap * p.kchunks;
        if (tap != cur_tap) {
            cur_tap = tap;
            const int df = p.tdf[tap], dh = p.tdh[tap], dw = p.tdw[tap];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int fi = r_f[i] + df, hi = r_h[i] + dh, wi = r_w[i] + dw;
                rvalid[i] = r_ok[i] && (unsigned)fi < (unsigned)p.F && (unsigned)hi < (unsigned)p.Hi &&
                            (unsigned)wi < (unsigned)p.Wi;
                roff[i] = ((long long)(r_bf[i] + df) * p.Hi + hi) * p.Wi + wi;
            }
        }
        const int c = kc * BK + acol;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = p.a0; cs = p.cs0; cc = c; }
        else { src = p.a1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && rvalid[i]) {
                v = *reinterpret_cast<const f32x4*>(src + roff[i] * cs + cc);
                if (p.ln_stats) {
                    const float mean = p.ln_stats[2 * roff[i]], inv = p.ln_stats[2 * roff[i] + 1];
                    const f32x4 g = *reinterpret_cast<const f32x4*>(p.ln_gamma + c);
                    v = (v - mean) * inv * g;
                }
            }
            ra[i] = v;
        }
        const float* wsrc = p.wp + ((long long)it * p.Npad + n0 + arow) * BK + acol;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wsrc + (long long)i * 32 * BK);
    };
    auto store_tiles = [&](float* As, float* Bs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&As[(arow + 32 * i) * LDS_STRIDE + acol]) = ra[i];
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *reinterpret_cast<f32x4*>(&Bs[(arow + 32 * i) * LDS_STRIDE + acol]) = rb[i];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int niter = p.ntaps * p.kchunks;
    load_tiles(0);
    store_tiles(As0, Bs0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const bool more = it + 1 < niter;
        if (more) load_tiles(it + 1);
        if (it & 1) mma_tile<BN>(As1, Bs1, acc, wm, wn, l31, hh);
        else mma_tile<BN>(As0, Bs0, acc, wm, wn, l31, hh);
        if (more) {
            if (it & 1) store_tiles(As0, Bs0);
            else store_tiles(As1, Bs1);
        }
        __syncthreads();
    }
    epilogue<BN>(acc, m0, n0, wm, wn, l31, hh, p.M, p.N, p.bias, p.resid, p.out, p.out_mode, HoWo, p.Wo, p.par_a,
                 p.par_b);
}

int launch_igemm(const IgemmParams& p_in, hipStream_t s) {
    IgemmParams p = p_in;
    p.cs0 = p.a0_stride ? p.a0_stride : p.C0;
    DPC_REQUIRE(p.C0 % 4 == 0 && p.C1 % 4 == 0, "igemm: channel counts must be multiples of 4");
    DPC_REQUIRE(p.ntaps >= 1 && p.ntaps <= 32, "igemm: 1..32 taps");
    DPC_REQUIRE(!(p.ln_stats && (p.ntaps != 1 || p.C1 != 0)), "igemm: LayerNorm prologue needs a 1-tap single-source op");
    DPC_REQUIRE(p.kchunks == igemm_kchunks(p.C0 + p.C1), "igemm: kchunks mismatch");
    if (p.M == 0) return DPC_OK;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    const double flops = 2.0 * (double)p.M * p.N * (double)p.ntaps * (p.C0 + p.C1);
    const double bytes = 4.0 * ((double)p.M * (p.N + (p.resid ? p.N : 0)) + (double)p.BF * p.Hi * p.Wi * (p.C0 + p.C1) +
                                (double)p.ntaps * (p.C0 + p.C1) * p.N);
    ProfScope prof((p.Npad % 128 == 0 && p.N > 64) ? PROF_IGEMM128 : PROF_IGEMM64, flops, bytes, s);
    if (p.Npad % 128 == 0 && p.N > 64) {
        const int grid = mtiles * (p.Npad / 128);
        const size_t lds = 2 * (BM + 128) * LDS_STRIDE * sizeof(float);
        static DeviceOnce once;
        if (!once) { DPC_HIP(hipFuncSetAttribute((const void*)igemm_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
        hipLaunchKernelGGL(igemm_kernel<128>, dim3(grid), dim3(256), lds, s, p);
    } else {
        DPC_REQUIRE(p.Npad % 64 == 0, "igemm: Npad must be a multiple of 64");
        const int grid = mtiles * (p.Npad / 64);
        const size_t lds = 2 * (BM + 64) * LDS_STRIDE * sizeof(float);
        static DeviceOnce once;
        if (!once) { DPC_HIP(hipFuncSetAttribute((const void*)igemm_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
        hipLaunchKernelGGL(igemm_kernel<64>, dim3(grid), dim3(256), lds, s, p);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------------------------ weight packing
struct PackTaps { int off[32]; };
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int N, int Npad, int K,
                                    int kchunks, int ntaps, long long stride_n, long long stride_c, PackTaps t, int bk) {
    const long long total = (long long)ntaps * kchunks * Npad * bk;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % bk);
        long long r = i / bk;
        const int n = (int)(r % Npad);
        r /= Npad;
        const int kc = (int)(r % kchunks);
        const int tap = (int)(r / kchunks);
        const int c = kc * bk + kk;
        float v = 0.f;
        if (n < N && c < K) v = w[n * stride_n + c * stride_c + t.off[tap]];
        wp[i] = v;
    }
}

// wp[tap][kc][n][bk] with kc over ceil(K/bk) chunks (bk = 32 for igemm, 16 for conv3h)
int launch_pack_weights(const float* w, float* wp, int N, int Npad, int K, int ntaps, long long stride_n,
                        long long stride_c, const int* tap_off_host, hipStream_t s, int bk) {
    DPC_REQUIRE(ntaps <= 32, "pack: at most 32 taps");
    PackTaps t;
    for (int i = 0; i < 32; ++i) t.off[i] = i < ntaps ? tap_off_host[i] : 0;
    const int kchunks = (K + bk - 1) / bk;
    const long long total = (long long)ntaps * kchunks * Npad * bk;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, s, w, wp, N, Npad, K, kchunks, ntaps, stride_n,
                       stride_c, t, bk);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------------------------ 7x7x7 stem (gather)
// K axis = flattened (tap, c).  Reads the reference-layout input [BF][C][H][W] directly: within a wave the
// lanes walk consecutive output points (consecutive w), so each gathered column is a coalesced read.
__global__ __launch_bounds__(256, 2) void stem_kernel(StemParams p) {
    constexpr int BN = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As0 = smem;
    float* Bs0 = As0 + BM * LDS_STRIDE;
    float* As1 = Bs0 + BN * LDS_STRIDE;
    float* Bs1 = As1 + BM * LDS_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    const int bid = xcd_remap(blockIdx.x, mtiles * ntn);
    const long long m0 = (long long)(bid / ntn) * BM;
    const int n0 = (bid % ntn) * BN;

    const int HW = p.H * p.W;
    // thread -> one A row (tid & 127), 16 k columns (tid >> 7) + 2 j
    const int arow = tid & 127, ak0 = tid >> 7;
    const long long m = m0 + arow;
    const bool rok = m < p.M;
    const long long mm = rok ? m : 0;
    const int bf = (int)(mm / HW), hw = (int)(mm - (long long)bf * HW);
    const int f = bf % p.F, h = hw / p.W, w = hw - h * p.W;
    // B: rows (tid>>3)+32 i, col (tid&7)*4
    const int brow = tid >> 3, bcol = (tid & 7) * 4;

    float ra[16];
    f32x4 rb[2];
    auto load_tiles = [&](int it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kk = ak0 + 2 * j;
            const int e = p.ktab[it * BK + kk];   // wave-uniform -> scalar load
            float v = 0.f;
            if (e >= 0 && rok) {
                const int df = ((e >> 24) & 0xff) - 64, dh = ((e >> 16) & 0xff) - 64, dw = ((e >> 8) & 0xff) - 64,
                          c = e & 0xff;
                const int fi = f + df, hi = h + dh, wi = w + dw;
                if ((unsigned)fi < (unsigned)p.F && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    v = p.x[((long long)(bf + df) * p.Ctot + p.c_off + c) * HW + hi * p.W + wi];
            }
            ra[j] = v;
        }
        const float* wsrc = p.wp + ((long long)it * p.Npad + n0 + brow) * BK + bcol;
        rb[0] = *reinterpret_cast<const f32x4*>(wsrc);
        rb[1] = *reinterpret_cast<const f32x4*>(wsrc + 32 * BK);
    };
    auto store_tiles = [&](float* As, float* Bs) {
#pragma unroll
        for (int j = 0; j < 16; ++j) As[arow * LDS_STRIDE + ak0 + 2 * j] = ra[j];
        *reinterpret_cast<f32x4*>(&Bs[brow * LDS_STRIDE + bcol]) = rb[0];
        *reinterpret_cast<f32x4*>(&Bs[(brow + 32) * LDS_STRIDE + bcol]) = rb[1];
    };

    f32x16 acc[2][1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][0][r] = 0.f;

    const int niter = p.kchunks;
    load_tiles(0);
    store_tiles(As0, Bs0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const bool more = it + 1 < niter;
        if (more) load_tiles(it + 1);
        if (it & 1) mma_tile<BN>(As1, Bs1, acc, wm, wn, l31, hh);
        else mma_tile<BN>(As0, Bs0, acc, wm, wn, l31, hh);
        if (more) {
            if (it & 1) store_tiles(As0, Bs0);
            else store_tiles(As1, Bs1);
        }
        __syncthreads();
    }
    epilogue<BN>(acc, m0, n0, wm, wn, l31, hh, p.M, p.N, p.bias, nullptr, p.out, 0, HW, p.W, 0, 0);
}

int launch_stem(const StemParams& p, hipStream_t s) {
    DPC_REQUIRE(p.Npad % 64 == 0, "stem: Npad must be a multiple of 64");
    if (p.M == 0) return DPC_OK;
    const int mtiles = (int)((p.M + BM - 1) / BM);
    const int grid = mtiles * (p.Npad / 64);
    const size_t lds = 2 * (BM + 64) * LDS_STRIDE * sizeof(float);
    ProfScope prof(PROF_STEM, 2.0 * (double)p.M * p.N * p.kchunks * 32, 4.0 * ((double)p.M * p.N + (double)p.M * p.C), s);
    static DeviceOnce once;
    if (!once) { DPC_HIP(hipFuncSetAttribute((const void*)stem_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
    hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(256), lds, s, p);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

__global__ void pack_stem_kernel(const float* __restrict__ w, float* __restrict__ wp, int* __restrict__ ktab, int N,
                                 int Npad, int C, int k, int kd, int kchunks) {
    const int taps = kd * k * k, K = taps * C, pad = k / 2, padd = kd / 2;
    const long long total = (long long)kchunks * Npad * BK;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % BK);
        long long r = i / BK;
        const int n = (int)(r % Npad);
        const int kc = (int)(r / Npad);
        const int kg = kc * BK + kk;
        float v = 0.f;
        if (kg < K && n < N) {
            const int tap = kg / C, c = kg - tap * C;
            v = w[((long long)n * C + c) * taps + tap];
        }
        wp[i] = v;
        if (n == 0) {
            int e = -1;
            if (kg < K) {
                const int tap = kg / C, c = kg - tap * C;
                const int a = tap / (k * k), b = (tap / k) % k, cc = tap % k;
                e = ((a - padd + 64) << 24) | ((b - pad + 64) << 16) | ((cc - pad + 64) << 8) | c;
            }
            ktab[kg] = e;
        }
    }
}

int launch_pack_stem(const float* w, float* wp, int* ktab, int N, int Npad, int C, int k, hipStream_t s, int kd) {
    if (kd <= 0) kd = k;                                  // cubic kernel (Unet3D); kd = 1: Conv2d 7x7 (Unet2D)
    const int kchunks = igemm_kchunks(kd * k * k * C);
    const long long total = (long long)kchunks * Npad * BK;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_stem_kernel, dim3(grid), dim3(256), 0, s, w, wp, ktab, N, Npad, C, k, kd, kchunks);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
