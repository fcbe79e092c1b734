// Host-side building blocks shared by the two U-Net orchestrators (unet3d.hip, unet2d.hip): owned device buffers,
// packed GEMM operands and the stack arena that carves the caller's workspace.
#pragma once
#include <string>

#include "common.h"

namespace dpc {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (p && bytes == n) return DPC_OK;          // re-pack of the same shape (training: the weights change every step)
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = n;
        DPC_HIP(hipMalloc(&p, n ? n : 4));
        return DPC_OK;
    }
    float* f() const { return reinterpret_cast<float*>(p); }
};

struct PackedConv {
    DevBuf wp;
    int N = 0, Npad = 0, K = 0, kchunks = 0, ntaps = 0;
    int sh = 1, sw = 1;
    bool flat3 = false;     // (1,3,3) stride-1 pad-(0,1,1) conv: wp3 also holds the 9-tap f16x3 pack for the big-tile kernel (conv3f3.hip)
    bool halo = false;      // 3x3x3 stride-1 conv: packed with bk = 16 for the LDS halo-tile kernel (conv3h.hip)
    DevBuf wp6;             // ... and pre-split into 3 bf16 planes for the bf16x6 kernel (conv3x6.hip)
    DevBuf wp3;             // ... or into 2 fp16 planes for the f16x3 kernel (conv3f3.hip)
    DevBuf wpw;             // f16x3 3x3x3 convs: the Winograd-over-frames pack as well (F(4,3): conv3w4.hip; F(2,3): conv3w.hip)
    DevBuf wp6g;            // every other op: pre-split planes for the bf16x6 implicit GEMM (igemm6.hip)
    signed char tdf[32], tdh[32], tdw[32];
};

struct Arena {
    char* base = nullptr;
    size_t off = 0, peak = 0, cap = 0;
    bool dry = true;
    bool overflow = false;
    void* alloc(size_t bytes) {
        off = align_up(off, 256);
        void* p = dry ? nullptr : (void*)(base + off);
        off += bytes;
        if (off > peak) peak = off;
        if (!dry && off > cap) overflow = true;
        return p;
    }
    float* allocf(long long n) { return reinterpret_cast<float*>(alloc((size_t)n * sizeof(float))); }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

// Build a PackedConv from a reference-layout conv weight [N][K][kd][kh][kw] / run it (unet3d.hip).
int pack_conv3d(PackedConv& pc, const float* w, int N, int K, int kd, int kh, int kw, int sh, int sw, int pd, int ph,
                int pw, hipStream_t s);
int run_conv(const PackedConv& pc, const float* a0, const float* a1, int C0, int C1, const float* bias,
             const float* resid, float* out, int BF, int F, int Hi, int Wi, int Ho, int Wo, const float* ln_stats,
             const float* ln_gamma, int out_mode, int par_a, int par_b, hipStream_t s, float* gn_part = nullptr,
             const float* in_coef = nullptr, const float* gn_raw = nullptr, const float* gn_coef = nullptr, float act_scale = 0.f, int a0_stride = 0);
// true when run_conv can take (gn_raw, gn_coef): the f16x3 implicit GEMM with whole 128-row tiles per sample
bool conv_can_fuse_gn_residual(const PackedConv& pc, long long rows_per_sample);
// (1,3,3) convolution of a 2-D net with N output channels on H x W images: does it run on the loader-wave halo kernel, whose r05 form
// emits per-IMAGE GroupNorm partial sums and applies a per-image GroupNorm + SiLU to its input?  Shape-only (never the batch).
bool conv2d_gn_fusable(int N, int Npad, int H, int W);

}  // namespace dpc
