// Helpers shared by the loader-wave halo kernels (conv3f3c.hip: direct 3x3x3 / (1,3,3); conv3w.hip: Winograd F(2,3) along frames):
// fp16 operand split, the swizzled 64-byte halo point layout, barriers.
#pragma once
#include "common.h"

namespace dpc {
namespace f3c {
constexpr int KC = 16, WROW = 64;
constexpr int HBS = 65536;                  // byte stride between the two halo buffers (power of two: the toggle is an XOR)
constexpr float SA = 16.0f, SW = 4096.0f;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lane_hw(int i, int& h, int& w) {      // 32 points of a 4 x 8 slab; see conv3h.hip
    if (i < 4) { h = 0; w = i; }
    else if (i < 12) { h = 1; w = i - 4; }
    else if (i < 16) { h = 0; w = i - 8; }
    else if (i < 20) { h = 3; w = i - 16; }
    else if (i < 28) { h = 2; w = i - 20; }
    else { h = 3; w = i - 24; }
}
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sat16(float x) { return __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f); }
__device__ __forceinline__ void split2(const f32x4 v, uint2& p1, uint2& p2) {
    const float x0 = sat16x(v.x), x1 = sat16x(v.y), x2 = sat16x(v.z), x3 = sat16x(v.w);
    p1.x = cvt_pk_f16(x0, x1);
    p1.y = cvt_pk_f16(x2, x3);
    p2.x = f16_sub_pk(x0, x1, p1.x);
    p2.y = f16_sub_pk(x2, x3, p1.y);
}
// byte offset of the 16-byte slot (plane 0, k-half kh) of halo point (pf, ph, pw) inside a buffer; plane 1 = offset ^ 32
__device__ __forceinline__ int slot0(int pf, int ph, int pw, int kh) {
    const int idx = (pf * 10 + ph) * 10 + pw;
    return idx * 64 + (((kh ^ (pw >> 2)) & 1) << 4) + ((((ph >> 1) & 1)) << 5);
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_done_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace f3c
}  // namespace dpc
