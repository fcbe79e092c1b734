// Conv3d 3x3x3 / stride 1 / pad 1 on channels-last activations: LDS-staged 3-D stencil tile + fp32 MFMA.
//
// A workgroup owns a 4(frames) x 4(rows) x 8(cols) block of output points (128 GEMM rows).  For each chunk of
// 16 input channels it stages the 6 x 6 x 10 halo of that block ONCE in LDS (zero-filled outside the tensor)
// and runs all 27 taps out of it: a tap is only an LDS address offset of the MFMA A-fragment read, so the
// activation is fetched from L2/HBM 2.8x (halo overhead) instead of 27x.  Only the weights of the current
// (tap, chunk) stream through a small double-buffered LDS tile (register prefetch one tap ahead).
// 256 threads = 4 waves as 2(M) x 2(N); BN = 64 or 128 output channels per workgroup; 39/49 KB LDS -> 3
// workgroups per CU, which is what hides the once-per-chunk halo load behind other workgroups' MFMAs.
// Same k-slot permutation as igemm.hip: a lane reads float4 = 4 consecutive channels; LDS row stride is 20
// floats (== 4 mod 8) so the b128 fragment reads of 16 consecutive halo points hit 16 distinct bank groups.
// Reference op: nn.Conv3d(dim, dim_out, (3,3,3), padding=(1,1,1)) in Block (video_diffusion_pytorch_conv3d.py:192).
#include "common.h"

namespace dpc {

constexpr int TF = 4, TH = 4, TW = 8;
constexpr int HF = TF + 2, HH = TH + 2, HWL = TW + 2;      // logical halo extent 6 x 6 x 10
constexpr int HWD = 12;                    // halo row PITCH in LDS: 12 (not 10) makes every ds_read_b128 lane group of
                                           // the A fragment hit 16 distinct bank groups for all 27 taps (see lane_hw)
constexpr int NLOG = HF * HH * HWL;        // 360 points loaded
constexpr int NHALO = HF * HH * HWD;       // 432 point slots allocated
constexpr int KC = 16;
constexpr int AST = 20;                    // LDS floats per halo point / per weight row
constexpr int HLOADS = (NLOG * 4 + 255) / 256;    // float4 loads per thread per chunk (6)

// MFMA tile row i (0..31) -> output point (h, w) of the 4 x 8 tile.  ds_read_b128 services lanes in the groups
// {0-3,12-15,20-27} and {4-11,16-19,28-31}; with pitch 12 the rows {0,2} / {1,3} each cover all 16 residues of
// (12 h + w) mod 16, so assigning rows 0,2 to the first group and 1,3 to the second is conflict-free under any tap shift.
__device__ __forceinline__ void lane_hw(int i, int& h, int& w) {
    if (i < 4) { h = 0; w = i; }
    else if (i < 12) { h = 1; w = i - 4; }
    else if (i < 16) { h = 0; w = i - 8; }
    else if (i < 20) { h = 3; w = i - 16; }
    else if (i < 28) { h = 2; w = i - 20; }
    else { h = 3; w = i - 24; }
}

template <int BN, bool BDIRECT>
__global__ __launch_bounds__(256, 3) void conv3h_kernel(Conv3hParams p) {
    constexpr int NT = BN / 64;
    constexpr int BL = BN / 64;            // weight float4 loads per thread per (tap, chunk)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* halo = smem;                    // [NHALO][AST]
    float* Bs0 = halo + NHALO * AST;       // [BN][AST]
    float* Bs1 = Bs0 + BN * AST;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int ntf = (p.F + TF - 1) / TF, nth = (p.H + TH - 1) / TH, ntw = (p.W + TW - 1) / TW;
    int bid = blockIdx.x;
    {   // XCD-aware remap: consecutive tiles (neighbours in w, h, f) share one XCD's L2
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n0 = (bid % ntn) * BN;
    int t = bid / ntn;
    const int w0 = (t % ntw) * TW; t /= ntw;
    const int h0 = (t % nth) * TH; t /= nth;
    const int f0 = (t % ntf) * TF;
    const int b = t / ntf;
    const int K = p.C0 + p.C1;

    // ---- halo load descriptors (loop invariant): float4 q = tid + 256 i  -> halo point q>>2, channel slot q&3
    long long hoff[HLOADS];
    bool hok[HLOADS];
#pragma unroll
    for (int i = 0; i < HLOADS; ++i) {
        const int q = tid + 256 * i;
        const int pt = q >> 2;
        const int pf = pt / (HH * HWL), ph = (pt / HWL) % HH, pw = pt % HWL;
        const int f = f0 - 1 + pf, h = h0 - 1 + ph, w = w0 - 1 + pw;
        hok[i] = pt < NLOG && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
        hoff[i] = (((long long)b * p.F + f) * p.H + h) * p.W + w;
    }
    const int hslot = (tid & 3) * 4;

    f32x4 hreg[HLOADS];
    auto load_halo = [&](int kc) {
        const int c = kc * KC + hslot;
        const float* src;
        int cs, cc;
        if (c < p.C0) { src = p.a0; cs = p.C0; cc = c; }
        else { src = p.a1; cs = p.C1; cc = c - p.C0; }
        const bool cok = c < K;
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (cok && hok[i]) v = *reinterpret_cast<const f32x4*>(src + hoff[i] * cs + cc);
            hreg[i] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < HLOADS; ++i) {
            const int q = tid + 256 * i;
            const int pt = q >> 2;
            const int phys = (pt / HWL) * HWD + pt % HWL;          // (pf*HH + ph) * pitch + pw
            if (q < NLOG * 4) *reinterpret_cast<f32x4*>(&halo[phys * AST + hslot]) = hreg[i];
        }
    };
    // ---- weights: wp[tap][kc][n][16]; thread -> row (tid>>2) + 64 i, float4 slot tid&3
    f32x4 breg[BL];
    const int brow = tid >> 2;
    auto load_b = [&](int tap, int kc) {
        const float* src = p.wp + (((long long)tap * p.kchunks + kc) * p.Npad + n0 + brow) * KC + hslot;
#pragma unroll
        for (int i = 0; i < BL; ++i) breg[i] = *reinterpret_cast<const f32x4*>(src + (long long)i * 64 * KC);
    };
    auto store_b = [&](float* Bs) {
#pragma unroll
        for (int i = 0; i < BL; ++i) *reinterpret_cast<f32x4*>(&Bs[(brow + 64 * i) * AST + hslot]) = breg[i];
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // A fragment base (halo coordinates of this lane's output point, tap (0,0,0)): frame wm*2+mt, row l31>>3, col l31&7
    int lh, lw;
    lane_hw(l31, lh, lw);
    const int abase = ((wm * 2) * HH + lh) * HWD + lw;
    const int a_lane = abase * AST + 4 * hh;
    const int b_lane = (wn * (BN / 2) + l31) * AST + 4 * hh;

    if constexpr (BDIRECT) {
        // Variant: weight fragments straight from global/L2 into registers (4 KB per (tap, chunk), shared by every
        // workgroup, L1/L2 resident), prefetched one tap ahead: no LDS staging and NO barrier inside the tap loop.
        const float* wlane = p.wp + ((long long)n0 + wn * (BN / 2) + l31) * KC + 4 * hh;
        f32x4 wc[NT][2], wn_[NT][2];
        auto ldw = [&](int tap, int kc, f32x4 (&w)[NT][2]) {
            const float* src = wlane + ((long long)tap * p.kchunks + kc) * p.Npad * KC;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 2; ++j) w[nt][j] = *reinterpret_cast<const f32x4*>(src + nt * 32 * KC + 8 * j);
        };
        load_halo(0);
        ldw(0, 0, wc);
        store_halo();
        __syncthreads();
        for (int kc = 0; kc < p.kchunks; ++kc) {
            const bool more_kc = kc + 1 < p.kchunks;
            if (more_kc) load_halo(kc + 1);
#pragma unroll 3
            for (int tap = 0; tap < 27; ++tap) {
                const bool last_tap = tap == 26;
                if (!last_tap || more_kc) ldw(last_tap ? 0 : tap + 1, last_tap ? kc + 1 : kc, wn_);
                const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
                const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * AST;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 a[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        a[mt] = *reinterpret_cast<const f32x4*>(&halo[aoff + mt * (HH * HWD * AST) + 8 * j]);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], wc[nt][j][s], acc[mt][nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) wc[nt][j] = wn_[nt][j];
            }
            if (more_kc) {
                __syncthreads();
                store_halo();
                __syncthreads();
            }
        }
    } else {
    load_halo(0);
    load_b(0, 0);
    store_halo();
    store_b(Bs0);
    __syncthreads();
    int it = 0;
    for (int kc = 0; kc < p.kchunks; ++kc) {
        const bool more_kc = kc + 1 < p.kchunks;
        if (more_kc) load_halo(kc + 1);
        for (int tap = 0; tap < 27; ++tap, ++it) {
            const bool last_tap = tap == 26;
            const bool has_next = !last_tap || more_kc;
            if (has_next) load_b(last_tap ? 0 : tap + 1, last_tap ? kc + 1 : kc);
            const float* Bs = (it & 1) ? Bs1 : Bs0;
            const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            const int aoff = a_lane + ((df * HH + dh) * HWD + dw) * AST;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 a[2], bfr[NT];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    a[mt] = *reinterpret_cast<const f32x4*>(&halo[aoff + mt * (HH * HWD * AST) + 8 * j]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bfr[nt] = *reinterpret_cast<const f32x4*>(&Bs[b_lane + nt * 32 * AST + 8 * j]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][s], bfr[nt][s], acc[mt][nt], 0, 0, 0);
            }
            if (has_next) store_b((it & 1) ? Bs0 : Bs1);
            __syncthreads();
            if (last_tap && more_kc) {
                store_halo();
                __syncthreads();
            }
        }
    }
    }   // !BDIRECT

    // ---- epilogue: + bias, channels-last store
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 32 + l31;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int f = f0 + wm * 2 + mt;
            if (f >= p.F) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * hh;     // row inside the 32-row MFMA tile = (h, w)
                int ih, iw;
                lane_hw(i, ih, iw);
                const int h = h0 + ih, w = w0 + iw;
                if (h < p.H && w < p.W)
                    p.out[((((long long)b * p.F + f) * p.H + h) * p.W + w) * p.N + n] = acc[mt][nt][r] + bv;
            }
        }
    }
}

int launch_conv3h(const Conv3hParams& p, hipStream_t s) {
    DPC_REQUIRE(p.C0 % 4 == 0 && p.C1 % 4 == 0, "conv3h: channel counts must be multiples of 4");
    DPC_REQUIRE(p.kchunks == (p.C0 + p.C1 + KC - 1) / KC, "conv3h: kchunks mismatch");
    if (p.B == 0) return DPC_OK;
    const long long tiles = (long long)p.B * ((p.F + TF - 1) / TF) * ((p.H + TH - 1) / TH) * ((p.W + TW - 1) / TW);
    const double M = (double)p.B * p.F * p.H * p.W;
    const double flops = 2.0 * M * p.N * 27.0 * (p.C0 + p.C1);
    const double bytes = 4.0 * (M * p.N + M * (p.C0 + p.C1) + 27.0 * (p.C0 + p.C1) * p.N);
    const bool wide = p.Npad % 128 == 0 && p.N > 64;
    ProfScope prof(wide ? PROF_CONV3H128 : PROF_CONV3H64, flops, bytes, s);
    // weights: BN=128 reads fragments straight from L2 (keeps LDS at 34.5 KB -> 4 workgroups/CU by LDS); BN=64 stages
    // them through LDS.  Both variants measure the same throughput; DPC_CONV3H_BDIRECT=0/1 forces one (A/B tests).
    static const int bforce = debug_switch("DPC_CONV3H_BDIRECT", -1);
    const int bdirect = bforce >= 0 ? bforce : (wide ? 1 : 0);
    if (wide) {
        const long long grid = tiles * (p.Npad / 128);
        DPC_REQUIRE(grid < (1ll << 31), "conv3h: grid too large");
        if (bdirect) {
            hipLaunchKernelGGL((conv3h_kernel<128, true>), dim3((unsigned)grid), dim3(256), NHALO * AST * sizeof(float), s, p);
        } else {
            const size_t lds = (NHALO * AST + 2 * 128 * AST) * sizeof(float);
            hipLaunchKernelGGL((conv3h_kernel<128, false>), dim3((unsigned)grid), dim3(256), lds, s, p);
        }
    } else {
        DPC_REQUIRE(p.Npad % 64 == 0, "conv3h: Npad must be a multiple of 64");
        const long long grid = tiles * (p.Npad / 64);
        DPC_REQUIRE(grid < (1ll << 31), "conv3h: grid too large");
        if (bdirect) {
            hipLaunchKernelGGL((conv3h_kernel<64, true>), dim3((unsigned)grid), dim3(256), NHALO * AST * sizeof(float), s, p);
        } else {
            const size_t lds = (NHALO * AST + 2 * 64 * AST) * sizeof(float);
            hipLaunchKernelGGL((conv3h_kernel<64, false>), dim3((unsigned)grid), dim3(256), lds, s, p);
        }
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
