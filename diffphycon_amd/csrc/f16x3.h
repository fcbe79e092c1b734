// Split-operand fp16 helpers for register-resident operands (see conv3f3.hip for the arithmetic): a value x (pre-scaled
// by a power of two into fp16's range) is written as h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1) -- 22 significant bits --
// and a product is  a1 b1 + (a1 b2 + a2 b1)  on v_mfma_f32_32x32x16_f16 with fp32 accumulation.
#pragma once
#include "common.h"

namespace dpc {
namespace h3 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sat16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f); }

// Hardware saturation (r04): with MODE.FP16_OVFL set (hwreg MODE bit 23) an fp16 RESULT that overflows is clamped to +-65504 instead of
// becoming inf, so the v_med3 in front of every operand conversion (one of the ~four VALU instructions a split element costs next to
// the MFMAs: DESIGN.md 6.1c) is not needed.  A kernel calls hw_sat_enable() once at its top and then uses sat16h / split_acc_h, which
// are the identity / unsaturated forms under DPC_FP16_OVFL (default) and the software clamp otherwise (A/B: -DDPC_FP16_OVFL=0).
// In range the results are bit-identical; beyond it (outside the f16x3 contract, watched by the range sentinel) hi = +-65504 either
// way and the remainder plane holds f16(x - hi) instead of 0.  dpc_selftest_fp16_clamp (api.hip) checks the mode bit on the device.
constexpr bool HW_SAT = DPC_FP16_OVFL != 0;                 // (common.h: fp16_ovfl_enable / sat16x, shared with the conv / stem loaders)
__device__ __forceinline__ void hw_sat_enable() { fp16_ovfl_enable(); }
__device__ __forceinline__ float sat16h(float x) { return HW_SAT ? x : sat16(x); }

// 8 floats (already scaled and inside +-65504) -> two f16x8 planes (plane 0 = leading term)
__device__ __forceinline__ void split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                       f16x8 (&o)[2]) {
#ifdef DPC_DBG_NO_SPLIT       // attribution builds only (tools/attn_ceiling.sh): the operand conversion is skipped, results are INVALID
    asm volatile("" :: "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7));
    o[0] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    o[1] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    asm volatile("" : "+v"(o[0]), "+v"(o[1]));
    return;
#endif
    uint4 a, b;
    a.x = cvt_pk(v0, v1); a.y = cvt_pk(v2, v3); a.z = cvt_pk(v4, v5); a.w = cvt_pk(v6, v7);
    // remainder plane: one v_fma_mix per element (common.h: f16_sub_pk)
    b.x = f16_sub_pk(v0, v1, a.x); b.y = f16_sub_pk(v2, v3, a.y); b.z = f16_sub_pk(v4, v5, a.z); b.w = f16_sub_pk(v6, v7, a.w);
    o[0] = __builtin_bit_cast(f16x8, a);
    o[1] = __builtin_bit_cast(f16x8, b);
}

// the two k-steps of a 32x32 accumulator used as an MFMA operand: k-step s = registers 8s .. 8s+7 of every lane
// (rows (i&3) + 8(i>>2) + 4hh + 16s, i = 0..7); any operand contracted against it must use the same k order.
// `mul` folds the descale of the producing GEMM and the pre-scale of this operand; SAT clamps into fp16's range.
template <bool SAT>
__device__ __forceinline__ void split_acc(const f32x16& v, float mul, f16x8 (&o)[2][2]) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = SAT ? sat16(v[r] * mul) : v[r] * mul;
    split8(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], o[0]);
    split8(t[8], t[9], t[10], t[11], t[12], t[13], t[14], t[15], o[1]);
}

template <bool SAT>
__device__ __forceinline__ void split_acc_h(const f32x16& v, float mul, f16x8 (&o)[2][2]) { split_acc<SAT && !HW_SAT>(v, mul, o); }

__device__ __forceinline__ void mfma3(f32x16& acc, const f16x8 (&a)[2], const f16x8 (&b)[2]) {
#ifdef DPC_DBG_NO_MFMA        // attribution builds only (tools/attn_ceiling.sh): the three matrix instructions are skipped, results are INVALID
    asm volatile("" : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
    return;
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);      // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
}

// LDS image of a pre-split 32-row weight block: [kstep][plane 2][row 32][khalf 2][8 fp16] = 1 KB per (kstep, plane);
// lane (row l31, khalf hh) reads 16 B at l31*32 + hh*16: a wave covers the KB contiguously (conflict-free ds_read_b128).
__device__ __forceinline__ void load_w2(const unsigned char* base, int kstep, int loff, f16x8 (&w)[2]) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) w[pl] = *reinterpret_cast<const f16x8*>(base + (kstep * 2 + pl) * 1024 + loff);
}

}  // namespace h3
}  // namespace dpc
