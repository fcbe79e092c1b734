// Vector epilogue of the f16x3 implicit-GEMM kernels (igemm6.hip, igemm_wide.hip, igemm_panel.hip), out_mode 0 / 2 with N % 4 == 0.
// Registers 4g .. 4g+3 of a lane's 32 x 32 accumulator are 4 consecutive rows of ONE channel; a 4 x 4 transpose inside each lane
// quad (two DPP exchange stages, no LDS) turns them into 4 consecutive CHANNELS of one row, so output, residual and the fused
// GroupNorm input move as dwordx4.
// r03: every residual / GroupNorm-input load of the tile is ISSUED before the first one is used.  The earlier form loaded inside
// the (mt, g, nt) loop behind `if (p.resid)` and bounds branches: hipcc waited vmcnt(0) per element -- 8-16 dependent round trips
// to HBM per wave (16-32 us of a 45 us workgroup on the C = 256 projections).  Now the loads are branch-free (rows past M re-read
// row M-1, columns past N re-read column 0; the results are dropped) and there is one exposed latency per 32-row slab.
#pragma once
#include "common.h"

namespace dpc {

// orow(mt, g): output row offset (elements) of this lane's row after the transpose, m(mt, g) its GEMM row (bounds, residual and
// GroupNorm rows are indexed by m * N); ncol(nt): this lane's first column.  bsmp: sample index of the tile (fused GroupNorm).
template <int MT, int NT, class RowFn, class ORowFn, class ColFn>
__device__ __forceinline__ void igemm_epilogue_vec(const IgemmParams& p, f32x16 (&acc)[MT][NT], int lane, long long bsmp, RowFn mrow,
                                                   ORowFn orow, ColFn ncol) {
    constexpr int GB = NT >= 3 ? 2 : 4;
    f32x4 cf[NT][3], bv[NT];
    const bool has_r = p.resid != nullptr, has_g = p.gn_raw != nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = ncol(nt) < p.N ? ncol(nt) : 0;
        bv[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_g) {
            const f32x4* c = reinterpret_cast<const f32x4*>(p.gn_coef) + (bsmp * (p.N >> 2) + (n >> 2)) * 5;
            cf[nt][0] = c[0]; cf[nt][1] = c[1]; cf[nt][2] = c[2];
        }
    }
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)              // one 32-row slab (half a slab for NT >= 3) at a time: its dwordx4 loads in flight, then
#pragma unroll                                   // its math and stores
    for (int gb = 0; gb < 4; gb += GB) {
        f32x4 rr[4][NT], gr[4][NT];
        if (has_r || has_g) {
#pragma unroll
            for (int g = gb; g < gb + GB; ++g) {
                const long long m = mrow(mt, g) < p.M ? mrow(mt, g) : p.M - 1;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = ncol(nt) < p.N ? ncol(nt) : 0;
                    // (non-temporal loads / stores: residual, GroupNorm input and output are touched once per launch -- the implicit-GEMM
                    // classes 27.5 -> 26.7 ms per S64 step, r03_bt)
                    if (has_r) rr[g][nt] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.resid + m * p.N + n));
                    if (has_g) gr[g][nt] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.gn_raw + m * p.N + n));
                }
            }
        }
#pragma unroll
        for (int g = gb; g < gb + GB; ++g) {
            const bool mok = mrow(mt, g) < p.M;
            const long long ro = orow(mt, g);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) x[e] = acc[mt][nt][4 * g + e];
                {   // 4 x 4 transpose across the lane quad: x[j] of lane L  <-  x[L] of lane j
#define DPC_QUAD_XCHG(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
                    const float r0 = DPC_QUAD_XCHG(b0 ? x[0] : x[1], 0xB1), r1 = DPC_QUAD_XCHG(b0 ? x[2] : x[3], 0xB1);   // quad_perm [1,0,3,2]
                    const float y0 = b0 ? r0 : x[0], y1 = b0 ? x[1] : r0, y2 = b0 ? r1 : x[2], y3 = b0 ? x[3] : r1;
                    const float s0 = DPC_QUAD_XCHG(b1 ? y0 : y2, 0x4E), s1 = DPC_QUAD_XCHG(b1 ? y1 : y3, 0x4E);           // quad_perm [2,3,0,1]
#undef DPC_QUAD_XCHG
                    x[0] = b1 ? s0 : y0; x[2] = b1 ? y2 : s0; x[1] = b1 ? s1 : y1; x[3] = b1 ? y3 : s1;
                }
                f32x4 v = f32x4{x[0], x[1], x[2], x[3]} * p.descale;
                if (p.bias) v += bv[nt];
                if (has_r) v += rr[g][nt];
                if (has_g) {
                    f32x4 y = (gr[g][nt] - cf[nt][0]) * cf[nt][1] + cf[nt][2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.0f + expf(-y[e]));
                    v += y;
                }
                if (mok && ncol(nt) < p.N) {
                    overflow_note4(p.oflag, v);          // f16x3 activation-range sentinel (common.h): this output may be split next
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p.out + ro + ncol(nt)));
                }
            }
        }
    }
}

}  // namespace dpc
