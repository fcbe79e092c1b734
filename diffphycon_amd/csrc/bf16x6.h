// fp32-on-bf16-matrix-core helpers shared by the fused attention kernels: exact 3-way bf16 split of fp32 values held
// in registers, and the 6-term product  a*b ~= a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)  on v_mfma_f32_32x32x16_bf16
// (fp32 accumulate; the dropped terms are <= 2^-24 |a||b|).
#pragma once
#include "common.h"

namespace dpc {
namespace b6 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float lo_f32(unsigned pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float hi_f32(unsigned pk) { return __uint_as_float(pk & 0xffff0000u); }

// 8 floats -> three bf16x8 planes (plane 0 = leading term)
__device__ __forceinline__ void split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7,
                                       bf16x8 (&o)[3]) {
    uint4 a, b, c;
    a.x = cvt_pk(v0, v1); a.y = cvt_pk(v2, v3); a.z = cvt_pk(v4, v5); a.w = cvt_pk(v6, v7);
    const float r0 = v0 - lo_f32(a.x), r1 = v1 - hi_f32(a.x), r2 = v2 - lo_f32(a.y), r3 = v3 - hi_f32(a.y);
    const float r4 = v4 - lo_f32(a.z), r5 = v5 - hi_f32(a.z), r6 = v6 - lo_f32(a.w), r7 = v7 - hi_f32(a.w);
    b.x = cvt_pk(r0, r1); b.y = cvt_pk(r2, r3); b.z = cvt_pk(r4, r5); b.w = cvt_pk(r6, r7);
    c.x = cvt_pk(r0 - lo_f32(b.x), r1 - hi_f32(b.x));
    c.y = cvt_pk(r2 - lo_f32(b.y), r3 - hi_f32(b.y));
    c.z = cvt_pk(r4 - lo_f32(b.z), r5 - hi_f32(b.z));
    c.w = cvt_pk(r6 - lo_f32(b.w), r7 - hi_f32(b.w));
    o[0] = __builtin_bit_cast(bf16x8, a);
    o[1] = __builtin_bit_cast(bf16x8, b);
    o[2] = __builtin_bit_cast(bf16x8, c);
}

// the two k-steps of a 32x32 accumulator used as an MFMA operand: k-step s = registers 8s .. 8s+7 of every lane
// (rows (i&3) + 8(i>>2) + 4hh + 16s, i = 0..7) -- any operand contracted against it must use the same k order.
__device__ __forceinline__ void split_acc(const f32x16& v, bf16x8 (&o)[2][3]) {
    split8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], o[0]);
    split8(v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15], o[1]);
}

__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8 (&a)[3], const bf16x8 (&b)[3]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);      // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

// LDS / global image of a pre-split 32-row weight block: [kstep][plane][row 32][khalf 2][8 bf16] = 1 KB per (kstep, plane);
// lane (row l31, khalf hh) reads 16 B at l31*32 + hh*16: a wave covers the KB contiguously (conflict-free ds_read_b128).
__device__ __forceinline__ void load_w3(const unsigned char* base, int kstep, int loff, bf16x8 (&w)[3]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w[pl] = *reinterpret_cast<const bf16x8*>(base + (kstep * 3 + pl) * 1024 + loff);
}

}  // namespace b6
}  // namespace dpc
