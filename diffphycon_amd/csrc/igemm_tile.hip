// Pixel-tile form of the f16x3 implicit GEMM for stride-1 few-tap operators inside one image plane: the four 2 x 2-tap output-parity
// classes of ConvTranspose3d (1,4,4)/(1,2,2) (video_diffusion_pytorch_conv3d.py:159-160; igemm_panel.hip ran them as im2col panels).
//
// An im2col panel loads every input pixel once PER TAP (4 x 64 rows for 64 output points) and re-reads the K x N weights once
// per 64 rows: at 64 -> 64 channels a launch moved 1 GB from L2 into the CUs for 268 MB of HBM traffic (7.8 TB/s on the L2 -> L1
// side, r03 microbench: 137 us against a 55 us HBM floor).  Here a workgroup stages the UNIQUE input pixels of its BM output points
// once -- the flattened pixel range [m0 + min tap offset, m0 + BM - 1 + max tap offset], split into fp16 planes, one LDS row per
// pixel, plus one zero row -- and a tap is an LDS row offset of the A fragment (a lane whose (y + dh, x + dw) leaves the image reads
// the zero row).  BM = 128 rows at 64 channels (weights re-read per 128 rows, 40 KB of LDS -> three workgroups per CU), BM = 64 at
// 128 channels.  K loop, weight ring, arithmetic and epilogue are igemm_panel.hip's.
#include <algorithm>

#include "common.h"
#include "f16x3.h"
#include "igemm_epilogue.h"

namespace dpc {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

namespace gt {
constexpr int RH = 80;                      // at most this many halo rows (max tap offset - min tap offset): Wi <= 78 for the 2 x 2 classes
}

// KCH: 32-channel chunks per tap (K = 32 KCH exactly); NTAPS taps; wave grid WM x WN, wave tile MT x NT blocks of 32 x 32
template <int KCH, int NTAPS, int MT, int NT, int WM, int WN, int BM>
__global__ __launch_bounds__(256, 2) void igemm3t_kernel(IgemmParams p, const unsigned char* __restrict__ wp6) {
    h3::hw_sat_enable();                               // (f16x3.h: operand conversions saturate in hardware)
    static_assert(WM * WN == 4 && WM * MT * 32 == BM, "four waves cover the tile");
    constexpr int KP = KCH * 32;                                  // channels per pixel
    constexpr int PITCH = KP * 4 + 16;                            // bytes per LDS row: plane 0 (KP fp16) | plane 1 | pad (conflict-free b128 reads)
    constexpr int GPR = KP / 8;                                   // 8-channel groups per pixel
    constexpr int JMAX = ((BM + gt::RH) * GPR + 255) / 256;       // load rounds per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, hh = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;
    const int HW = p.Hi * p.Wi;
    // tap offsets in flattened pixels (wave-uniform)
    int toff[NTAPS], tdh[NTAPS], tdw[NTAPS];
    int lo = 1 << 30, hi = -(1 << 30);
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        tdh[t] = p.tdh[t]; tdw[t] = p.tdw[t];
        toff[t] = tdh[t] * p.Wi + tdw[t];
        lo = min(lo, toff[t]); hi = max(hi, toff[t]);
    }
    const int NR = BM + hi - lo;                                  // staged pixel rows; row NR is the zero row
    const long long P0 = m0 + lo;                                 // flattened pixel of LDS row 0 (may be < 0)
    const long long PT = (long long)p.BF * HW;                    // pixels in the tensor

    // ---- 1. the unique pixels: group g = tid + 256 j -> LDS row g / GPR, channels (g % GPR) * 8 .. + 7; all loads issued first
    f32x4 v[JMAX][2];
    const int ngroups = NR * GPR;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int g = tid + 256 * j;
        const int row = g / GPR, c = (g % GPR) * 8;
        const long long px = P0 + row;
        const bool ok = g < ngroups && px >= 0 && px < PT;
        const long long pp = ok ? px : 0;
        const float* src = c < p.C0 ? p.a0 + pp * p.C0 + c : p.a1 + pp * p.C1 + (c - p.C0);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[j][0] = *reinterpret_cast<const f32x4*>(src);
        v[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
        if (!ok) { v[j][0] = z; v[j][1] = z; }
    }
    // weight fragments: [chunk][Npad / 32][k-step][plane][half][n 32][16 B]; chunk = tap * KCH + kc as packed
    const unsigned char* wlane = wp6 + (long long)(wn * NT) * 4096 + hh * 512 + l31 * 16;
    const long long wchunk = (long long)p.Npad * 128;
    f16x8_t w[4][NT][2];
    auto ldw = [&](int step, f16x8_t (&ws)[NT][2]) {
        const unsigned char* src = wlane + (step >> 1) * wchunk + (step & 1) * 2048;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) ws[nt][pl] = *reinterpret_cast<const f16x8_t*>(src + nt * 4096 + pl * 1024);
    };
    ldw(0, w[0]);
    ldw(1, w[1]);
    ldw(2, w[2]);
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int g = tid + 256 * j;
        if (g < ngroups) {
            const int row = g / GPR, c = (g % GPR) * 8;
            const f32x4 a = v[j][0] * p.act_scale, b = v[j][1] * p.act_scale;
            h3::f16x8 pl[2];
            h3::split8(h3::sat16h(a.x), h3::sat16h(a.y), h3::sat16h(a.z), h3::sat16h(a.w), h3::sat16h(b.x), h3::sat16h(b.y), h3::sat16h(b.z),
                       h3::sat16h(b.w), pl);
            unsigned char* dst = smem_t + row * PITCH + c * 2;
            *reinterpret_cast<h3::f16x8*>(dst) = pl[0];
            *reinterpret_cast<h3::f16x8*>(dst + KP * 2) = pl[1];
        }
    }
    for (int i = tid; i < PITCH / 16; i += 256) *reinterpret_cast<uint4*>(smem_t + NR * PITCH + i * 16) = uint4{0, 0, 0, 0};      // the zero row
    // this lane's output points (one per 32-row block) and, per tap, the LDS row of its input pixel
    int arow[NTAPS][MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = wm * MT * 32 + mt * 32 + l31;
        const long long m = m0 + r;
        const long long mm = m < p.M ? m : 0;
        const int hw = (int)(mm % HW);
        const int y = hw / p.Wi, x = hw - y * p.Wi;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const bool ok = m < p.M && (unsigned)(y + tdh[t]) < (unsigned)p.Hi && (unsigned)(x + tdw[t]) < (unsigned)p.Wi;
            arow[t][mt] = (ok ? r + toff[t] - lo : NR) * PITCH + hh * 16;
        }
    }
    __syncthreads();

    // ---- 2. K loop from LDS: step = (tap, chunk, k-step); 16 channels per step
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    f16x8_t fa[2][MT][2];
    auto lda = [&](int step, f16x8_t (&a)[MT][2]) {
        const int t = step / (2 * KCH), within = step % (2 * KCH);          // (compile-time after unrolling)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) a[mt][pl] = *reinterpret_cast<const f16x8_t*>(smem_t + arow[t][mt] + pl * KP * 2 + within * 32);
    };
    auto mma = [&](const f16x8_t (&a)[MT][2], const f16x8_t (&ws)[NT][2]) {
        constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};          // small terms first (as igemm3)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][PA[term]], ws[nt][PB[term]], acc[mt][nt], 0, 0, 0);
    };
    constexpr int NS = NTAPS * 2 * KCH;
    static_assert(NS % 4 == 0, "the weight ring is walked four k-steps at a time");
    // spent weight sets (the MFMAs' B operand) keep their registers for >= 4 younger MFMAs (common.h: mfma_keep), as in igemm_panel.hip
    constexpr int BACK = 3 * MT * NT >= 4 ? 1 : 2;
    f16x8_t spent[3][NT][2];
    lda(0, fa[0]);
#pragma unroll
    for (int s = 0; s < NS; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = s + u;
            if (st + 1 < NS) lda(st + 1, fa[(u + 1) & 1]);
            mma(fa[u & 1], w[u]);
            __builtin_amdgcn_sched_barrier(0);                    // (the re-load of set u+3 = u-1 stays behind this step's MFMAs: DESIGN.md 6.2)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) spent[st % 3][nt][pl] = w[u][nt][pl];
            if (st >= BACK) mfma_keep_set<MT, NT>(acc, spent[(st - BACK) % 3]);
            if (st + 3 < NS) ldw(st + 3, w[(u + 3) & 3]);
        }
    }
#pragma unroll
    for (int b = 0; b < BACK; ++b) mfma_keep_set<MT, NT>(acc, spent[(NS - 1 - b) % 3]);

    // ---- 3. epilogue (igemm_epilogue.h)
    const int q3 = l31 & 3;
    auto mrow = [&](int mt, int g) { return m0 + wm * MT * 32 + mt * 32 + 8 * g + 4 * hh + q3; };
    auto orow = [&](int mt, int g) -> long long {
        const long long m = mrow(mt, g);
        if (p.out_mode == 0) return m * p.N;
        const long long mm = m < p.M ? m : 0;                      // ConvTranspose parity scatter into [BF][2 H][2 W][N]
        const long long bf = mm / HW;
        const int hw = (int)(mm - bf * HW), y = hw / p.Wi, x = hw - y * p.Wi;
        return ((bf * (2 * p.Hi) + 2 * y + p.par_a) * (long long)(2 * p.Wi) + 2 * x + p.par_b) * p.N;
    };
    auto ncol = [&](int nt) { return wn * NT * 32 + nt * 32 + (l31 & ~3); };
    igemm_epilogue_vec<MT, NT>(p, acc, lane, 0, mrow, orow, ncol);
}

// shape-only rule (never the batch): four taps inside one plane, stride 1, K = 64 -> N = 64 or K = 128 -> N = 128
bool igemm3t_supported(const IgemmParams& p) {
    static const int on = debug_switch("DPC_IGEMM_TILE", 1);
    const int K = p.C0 + p.C1;
    if (!on || p.ntaps != 4 || p.sh != 1 || p.sw != 1 || p.Hi != p.Ho || p.Wi != p.Wo || p.a0_stride || p.ln_stats || p.gn_raw) return false;
    if (!(p.out_mode == 0 || p.out_mode == 2) || p.N != p.Npad || p.C0 % 8 || p.C1 % 8) return false;
    if (!((K == 64 && p.N == 64) || (K == 128 && p.N == 128))) return false;
    int lo = 1 << 30, hi = -(1 << 30);
    for (int t = 0; t < 4; ++t) {
        if (p.tdf[t] != 0) return false;
        const int o = p.tdh[t] * p.Wi + p.tdw[t];
        lo = std::min(lo, o); hi = std::max(hi, o);
    }
    return hi - lo <= gt::RH;
}

template <int KCH, int MT, int NT, int BM>
static int launch_t(const IgemmParams& p, const void* wp6, hipStream_t s) {
    constexpr int LDS = (BM + gt::RH + 1) * (KCH * 128 + 16);
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)igemm3t_kernel<KCH, 4, MT, NT, 2, 2, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        once = true;
    }
    int lo = 1 << 30, hi = -(1 << 30);
    for (int t = 0; t < 4; ++t) {
        const int o = p.tdh[t] * p.Wi + p.tdw[t];
        lo = std::min(lo, o); hi = std::max(hi, o);
    }
    const size_t lds = (size_t)(BM + hi - lo + 1) * (KCH * 128 + 16);            // what this launch needs (occupancy follows the image width)
    const unsigned nwg = (unsigned)((p.M + BM - 1) / BM);
    hipLaunchKernelGGL((igemm3t_kernel<KCH, 4, MT, NT, 2, 2, BM>), dim3(nwg), dim3(256), lds, s, p, (const unsigned char*)wp6);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_igemm3t(const IgemmParams& p, const void* wp6, hipStream_t s) {
    if (p.C0 + p.C1 == 64) return launch_t<2, 2, 1, 128>(p, wp6, s);       // 64 -> 64: 128 rows, waves 2 x 2, 64 x 32 each
    return launch_t<4, 1, 2, 64>(p, wp6, s);                               // 128 -> 128: 64 rows, waves 2 x 2, 32 x 64 each
}

}  // namespace dpc
