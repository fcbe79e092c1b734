// Training-step kernels of the smoke denoiser: what the backward pass of Unet3D_with_Conv3D needs beyond the
// forward kernels and the input-gradient operators of surr.hip --
//   * convolution WEIGHT gradients for every conv shape of the net (3x3x3, 1x1x1, (1,4,4)/s2, ConvTranspose (1,4,4)/s2, 7x7x7 stem),
//   * column reductions (bias / LayerNorm-gamma gradients),
//   * the backward of the temporal attention core (rotary, relative-position bias, bias gradient),
//   * q_sample + conditioning, the mse loss and its gradient, the global gradient norm, fused Adam + EMA.
// Reference: /root/reference/diffusion/diffusion_2d_smoke.py q_sample :791-797, p_losses :809-831, Trainer.train :998-1054
// (clip_grad_norm_ :1027, Adam :912, EMA :920); model/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py :276-352.
// All reductions run in a fixed order (two-stage, no float atomics): a training step is bit-reproducible run to run.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"

namespace dpc {

// ------------------------------------------------------------------------------------ convolution weight gradient
// dW[n][c][df][dh][dw] = sum over output points p = (b, f, ho, wo) of dy[p][n] * x[b][f + df - pf][ho sh + dh - ph][wo sw + dw - pw][c]
// (zero outside the tensor), x channels-last [B, F, Hi, Wi, C], dy channels-last [B, F, Ho, Wo, N].
//
// GEMM view: the reduction axis is the POINT axis, which is the slow axis of both channels-last operands -- an MFMA operand that
// wants several reduction indices per lane would need a transpose.  The fp32 MFMA 32x32x2 takes ONE value per lane per operand
// (A: row = lane % 32, k = lane / 32), so both operands are read straight from global memory in their natural layout: 32 lanes
// = 32 consecutive floats of a point (128-byte segments), the two half-waves = two consecutive points.  No LDS, no transpose.
// A "row fragment" is 32 consecutive floats of the run of kw * C floats that starts at x[.., wo sw - pw][0] for a (df, dh) tap
// row: for C % 32 == 0 it is (dw, 32-channel block), for the 8-channel stem it is 4 taps x 8 channels.  Zero padding and every
// edge come from the range check of raw buffer loads (per-row descriptors; offsets outside [0, Wi C) read 0), never from branches.
// A wave owns RA row fragments x RB column fragments (32 output channels each) = RA RB accumulator tiles, walks its slab of
// (b, f, ho) rows two points at a time with a one-step register prefetch, and stores its partial sums; a second kernel adds the
// slabs in fixed order and scatters into the reference weight layout.  Exact fp32 products, fp32 accumulation.
// r04: (a) the loads of step q + 4 are issued while step q is multiplied (a ring of four static register stages: a step is 8 MFMAs =
// 0.2 us of matrix-pipe time, far less than a global-memory round trip, and a 128-accumulator wave has one or two neighbours on its
// SIMD to hide it: the one-step prefetch of r03 ran at 0.27 of the fp32 roof); (b) a wave's 8 accumulator tiles are RA x RB = 4 x 2, or
// 2 x 4 when the operand has only two row fragments (the 64-channel 1x1 projections: half of the 4 x 2 form's MFMAs multiplied
// fragments that do not exist).  Per-accumulator summation order is unchanged.
constexpr int WG_DEPTH = 4;

struct WgradParams {
    const float* x;
    const float* dy;
    float* part;               // [nslab][nrf][32][ncf * 32]
    int B, F, Hi, Wi, C, Ho, Wo, N;
    int kf, kh, kw, sh, sw, pf, ph, pw;
    int fpr, nrf, ncf, nru, ncu, nslab;
    long long rows;            // B * F * Ho
    unsigned dy_limit_bits;    // != 0: an operand element with |v| above this (bit pattern of a positive float; Inf / NaN too) raises
    int* oflag;                //       the gradient-range sentinel (wgrad3.hip): bit 1 for the GRADIENT operand -- it is also the input of
                               //       this layer's f16x3 backward-DATA convolution, which clamps there -- and bit 0 for the ACTIVATION
                               //       operand (not a loss-scale matter: include/dpc.h)
    int x_is_grad;             // ConvTranspose call form: the gradient is passed as x, the activation as dy
};

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

template <int WG_RA, int WG_RB>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int unit = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (unit >= p.nru * p.ncu) return;
    const int ru = unit / p.ncu, cu = unit % p.ncu;
    const int slab = blockIdx.y;
    const long long rows_per = (p.rows + p.nslab - 1) / p.nslab;
    const long long row_begin = slab * rows_per, row_end = min(p.rows, row_begin + rows_per);

    // per-fragment constants
    int df[WG_RA], dh[WG_RA];
    unsigned ea[WG_RA];                      // byte offset of this lane inside the tap row's run (incl. the second point of a pair)
    bool rf_ok[WG_RA];
#pragma unroll
    for (int j = 0; j < WG_RA; ++j) {
        const int rf = ru * WG_RA + j;
        rf_ok[j] = rf < p.nrf;
        const int run = rf_ok[j] ? rf / p.fpr : 0, jj = rf_ok[j] ? rf % p.fpr : 0;
        df[j] = run / p.kh;
        dh[j] = run % p.kh;
        ea[j] = (unsigned)((jj * 32 + l31 + (hh * p.sw - p.pw) * p.C) * 4);
    }
    unsigned eb[WG_RB];
#pragma unroll
    for (int i = 0; i < WG_RB; ++i) {
        const int n = (cu * WG_RB + i) * 32 + l31;
        eb[i] = n < p.N ? (unsigned)((hh * p.N + n) * 4) : 0x80000000u;
    }
    const unsigned xrow_bytes = (unsigned)(p.Wi * p.C * 4), yrow_bytes = (unsigned)(p.Wo * p.N * 4);
    const unsigned astep = (unsigned)(2 * p.sw * p.C * 4), bstep = (unsigned)(2 * p.N * 4);
    const int npair = (p.Wo + 1) >> 1;

    f32x16 acc[WG_RA][WG_RB];
#pragma unroll
    for (int j = 0; j < WG_RA; ++j)
#pragma unroll
        for (int i = 0; i < WG_RB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    __amdgpu_buffer_rsrc_t ra[WG_RA], rb;
    // request cursor: row = (b, f, ho) as counters + the two row pointers, advanced by constant strides (a 64-bit division and a
    // multiply chain per row advance, inlined once per ring stage, cost more code than the rest of the kernel)
    int c_ho = (int)(row_begin % p.Ho), c_f = (int)((row_begin / p.Ho) % p.F);
    const long long xrow = (long long)p.Wi * p.C, yrow = (long long)p.Wo * p.N;
    const float* c_x = p.x + (((row_begin / p.Ho / p.F) * p.F + c_f) * p.Hi + (long long)c_ho * p.sh) * xrow;     // row (b, f, ho * sh)
    const float* c_y = p.dy + row_begin * yrow;
    long long dj[WG_RA];                     // fragment j reads the row (df - pf) frames and (dh - ph) rows away
#pragma unroll
    for (int j = 0; j < WG_RA; ++j) dj[j] = ((long long)(df[j] - p.pf) * p.Hi + (dh[j] - p.ph)) * xrow;
    const long long x_next_f = (long long)(p.Hi - (p.Ho - 1) * p.sh) * xrow;     // from the last row of a frame to row 0 of the next
    auto setup_row = [&]() {
#pragma unroll
        for (int j = 0; j < WG_RA; ++j) {
            const int fs = c_f + df[j] - p.pf, hs = c_ho * p.sh + dh[j] - p.ph;
            const bool ok = rf_ok[j] && (unsigned)fs < (unsigned)p.F && (unsigned)hs < (unsigned)p.Hi;
            ra[j] = make_rsrc(ok ? c_x + dj[j] : p.x, ok ? xrow_bytes : 0u);
        }
        rb = make_rsrc(c_y, yrow_bytes);
    };
    auto next_row = [&]() {
        c_y += yrow;
        if (++c_ho == p.Ho) {
            c_ho = 0;
            c_x += x_next_f;                 // (frames of consecutive samples are contiguous: f wraps without a jump)
            if (++c_f == p.F) c_f = 0;
        } else {
            c_x += (long long)p.sh * xrow;
        }
        setup_row();
    };
    float a_st[WG_DEPTH][WG_RA], b_st[WG_DEPTH][WG_RB];
    auto issue = [&](int pr, float (&a)[WG_RA], float (&b)[WG_RB]) {
#pragma unroll
        for (int j = 0; j < WG_RA; ++j) a[j] = bload(ra[j], ea[j] + (unsigned)pr * astep);
#pragma unroll
        for (int i = 0; i < WG_RB; ++i) {
            // the second point of the last pair of an odd row lies beyond the row: (2 pr + hh) >= Wo -> its bytes are past the descriptor
            b[i] = bload(rb, eb[i] + (unsigned)pr * bstep);
        }
    };
    unsigned amax = 0, bmax = 0;             // largest |operand| bit pattern this lane multiplied, per side (a ConvTranspose passes its
                                             // output gradient as x: p.x_is_grad says which side is the gradient)
    if (row_begin < row_end) {
        // request cursor (row, pr) of step `req`; it runs WG_DEPTH steps ahead of the multiplications and stops at the last step
        // (steps past the end re-request the last one: harmless, never multiplied)
        int pr = 0;
        const long long total = (row_end - row_begin) * npair;
        long long req = 0;
        auto advance = [&]() {
            if (req + 1 < total) {
                ++req;
                if (++pr == npair) { pr = 0; next_row(); }
            }
        };
        setup_row();
#pragma unroll
        for (int d = 0; d < WG_DEPTH; ++d) {
            if (d) advance();
            issue(pr, a_st[d], b_st[d]);
        }
        for (long long q = 0; q < total; q += WG_DEPTH) {
#pragma unroll
            for (int d = 0; d < WG_DEPTH; ++d) {
                if (q + d < total) {
#pragma unroll
                    for (int j = 0; j < WG_RA; ++j)
#pragma unroll
                        for (int i = 0; i < WG_RB; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_st[d][j], b_st[d][i], acc[j][i], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < WG_RB; ++i) bmax = max(bmax, __builtin_bit_cast(unsigned, b_st[d][i]) & 0x7fffffffu);
#pragma unroll
                    for (int j = 0; j < WG_RA; ++j) amax = max(amax, __builtin_bit_cast(unsigned, a_st[d][j]) & 0x7fffffffu);
                }
                advance();
                issue(pr, a_st[d], b_st[d]);             // step q + d + WG_DEPTH into the stage just consumed
            }
        }
    }
    if (p.dy_limit_bits) {
        const unsigned gmax = p.x_is_grad ? amax : bmax, xmax = p.x_is_grad ? bmax : amax;
        const int bits = (gmax > p.dy_limit_bits ? 2 : 0) | (xmax > p.dy_limit_bits ? 1 : 0);
        if (bits) atomicOr(p.oflag, bits);
    }
    const int ldn = p.ncf * 32;
#pragma unroll
    for (int j = 0; j < WG_RA; ++j) {
        const int rf = ru * WG_RA + j;
        if (rf >= p.nrf) continue;
#pragma unroll
        for (int i = 0; i < WG_RB; ++i) {
            const int cf = cu * WG_RB + i;
            if (cf >= p.ncf) continue;
            float* dst = p.part + (((long long)slab * p.nrf + rf) * 32) * ldn + cf * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // (read out of the accumulator file element by element: left to itself the register allocator copies all 128
                // accumulators into VGPRs at the loop exit -- 138 + 128 registers, one wave per SIMD instead of three)
                float v;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[j][i][r]));
                dst[(long long)((r & 3) + 8 * (r >> 2) + 4 * hh) * ldn] = v;
            }
        }
    }
}

// dW (reference layout [N][ctot][kf][kh][kw], channel slice [coff, coff + C)) = scale * sum over slabs, fixed order.
// r04: 64 outputs x 4 slab quarters per workgroup, eight loads in flight per thread (one thread walking all <= 512 slabs of an output
// ran at 1 TB/s: 6 ms of the training step, r04_t trace); quarter sums are added in the order 0, 1, 2, 3: deterministic.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nslab, int nrf, int ncf,
                                                          int fpr, int C, int c_valid, int N, int kf, int kh, int kw, int ctot, int coff,
                                                          float scale, int accumulate) {
    __shared__ float quarter[4][64];
    const long long total = (long long)nrf * 32 * ncf * 32;
    const int ldn = ncf * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int per = (nslab + 3) >> 2, k0 = ty * per, k1 = min(nslab, k0 + per);
    for (long long base = (long long)blockIdx.x * 64; base < total; base += (long long)gridDim.x * 64) {
        const long long idx = base + tx;
        float s = 0.f;
        if (idx < total) {
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += part[(long long)(k + u) * total + idx];
            }
            for (; k < k1; ++k) a[0] += part[(long long)k * total + idx];
            s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
        quarter[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && idx < total) {
            s = ((quarter[0][tx] + quarter[1][tx]) + quarter[2][tx]) + quarter[3][tx];
            const int n = (int)(idx % ldn);
            const long long rr = idx / ldn;
            const int r = (int)(rr & 31), rf = (int)(rr >> 5);
            const int run = rf / fpr, jj = rf % fpr;
            const int e = jj * 32 + r, dwi = e / C, c = e % C;
            if (n < N && dwi < kw && c < c_valid) {
                const int df = run / kh, dh = run % kh;
                float* dst = dw + ((((long long)n * ctot + coff + c) * kf + df) * kh + dh) * kw + dwi;
                *dst = accumulate ? *dst + s * scale : s * scale;
            }
        }
        __syncthreads();
    }
}

// accumulator tiles of a wave: 2 x 4 when the row operand has two fragments, else 4 x 2 (three waves per SIMD).  A 4 x 4 form (256
// accumulators, one wave per SIMD, twice the FLOP per loaded byte) was measured and lost: 265.9 vs 263.5 ms per training step (r04_m)
static void wgrad_shape(int nrf, int ncf, int& ra, int& rb) {
    (void)ncf;
    if (nrf <= 2) { ra = 2; rb = 4; }
    else { ra = 4; rb = 2; }
}
static int wgrad_slabs(int nrf, int ncf, long long rows) {
    int WG_RA, WG_RB;
    wgrad_shape(nrf, ncf, WG_RA, WG_RB);
    const int units = ((nrf + WG_RA - 1) / WG_RA) * ((ncf + WG_RB - 1) / WG_RB), wgs = (units + 3) / 4;
    int nslab = std::max(1, (1024 + wgs - 1) / wgs);          // ~4 workgroups per CU
    const long long min_rows = 4;                             // at least a few rows per slab
    nslab = (int)std::min<long long>(nslab, std::max<long long>(1, rows / min_rows));
    return std::min(nslab, 512);
}

// ------------------------------------------------------------------------------------ column reductions
// out[c] = sum_r dy[r][c]                               (conv / linear bias gradients)
// out[c] = sum_r dy[r][c] * (x[r][c] - mean_r) * rstd_r (channel-LayerNorm gamma gradient, stats [rows][2])
// dy, x [rows][C], C % 4 == 0 and (C / 4) | 256; partials [nblk][C] in fp64, fixed-order finalize.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ stats, double* __restrict__ part, long long rows,
                                                            int C, int nblk) {
    __shared__ double red[256][4];
    const int tid = threadIdx.x, tpr = C >> 2, rpp = 256 / tpr;
    const int c4 = tid % tpr, rsub = tid / tpr;
    const long long rpb = (rows + nblk - 1) / nblk;
    const long long r_begin = blockIdx.x * rpb, r_end = min(rows, r_begin + rpb);
    double a[4] = {0, 0, 0, 0};
    for (long long r = r_begin + rsub; r < r_end; r += rpp) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + r * C + c4 * 4);
        if (x) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * C + c4 * 4);
            const float mean = stats[2 * r], rstd = stats[2 * r + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] += (double)(d[i] * ((v[i] - mean) * rstd));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] += (double)d[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[tid][i] = a[i];
    __syncthreads();
    if (rsub == 0) {
        double t[4] = {0, 0, 0, 0};
        for (int k = 0; k < rpp; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] += red[k * tpr + c4][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) part[(long long)blockIdx.x * C + c4 * 4 + i] = t[i];
    }
}

// 16 channels x 16 partial lanes per block: lane j adds partials j, j + 16, ... in order, then the 16 lane sums are added in order
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, float* __restrict__ out, int C, int nblk,
                                                          float scale, int accumulate) {
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0;
    if (c < C)
        for (int k = j; k < nblk; k += 16) s += part[(long long)k * C + c];
    red[j][cl] = s;
    __syncthreads();
    if (j == 0 && c < C) {
        double t = 0;
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        const float v = (float)(t * (double)scale);
        out[c] = accumulate ? out[c] + v : v;
    }
}

// GroupNorm affine gradients from the per-(sample, chunk, channel) partial sums (A = sum dz, Bs = sum dz xhat) that
// gn_bwd_partial_kernel (norm.hip) leaves in the workspace: dgamma[c] = sum_b (scale_b,c + 1) Bs, dbeta[c] = sum_b (scale_b,c + 1) A
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const double* __restrict__ part, const float* __restrict__ scale_shift,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int nchunk) {
    // 16 channels x 16 lanes per block; lane j takes the (sample, chunk) pairs j, j + 16, ... in order; fixed-order finish
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double dg = 0, db = 0;
    if (c < C)
        for (int idx = j; idx < B * nchunk; idx += 16) {
            const int b = idx / nchunk;
            const double* src = part + ((long long)idx * C + c) * 2;
            const double sc = scale_shift ? (double)scale_shift[(long long)b * 2 * C + c] + 1.0 : 1.0;
            db += sc * src[0];
            dg += sc * src[1];
        }
    red[0][j][cl] = dg;
    red[1][j][cl] = db;
    __syncthreads();
    if (j == 0 && c < C) {
        double tg = 0, tb = 0;
        for (int k = 0; k < 16; ++k) { tg += red[0][k][cl]; tb += red[1][k][cl]; }
        dgamma[c] = (float)tg;
        dbeta[c] = (float)tb;
    }
}

int launch_gn_param_grad(const void* ws, const float* scale_shift, float* dgamma, float* dbeta, int B, int C, int nchunk, hipStream_t s) {
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3((C + 15) / 16), dim3(256), 0, s, reinterpret_cast<const double*>(ws), scale_shift,
                       dgamma, dbeta, B, C, nchunk);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------------------------ temporal attention backward
// Attention.forward (...conv3d.py:293-352) on sequences addressed like launch_attention: q' = rot(q scale), k' = rot(k),
// S = q' k'^T + bias[head], P = softmax_j S, O = P v.  Given dO: dv_j = sum_i P_ij dO_i; dP_ij = dO_i . v_j;
// dS_ij = P_ij (dP_ij - sum_j P_ij dP_ij); dq' = dS k', dk' = dS^T q'; dq = scale rot^T(dq'), dk = rot^T(dk');
// dbias[head][i][j] = sum over sequences of dS_ij.   One workgroup = 256 / Lp sequences of one head (Lp = 32 | 64), thread =
// (sequence, token); persistent over sequence groups so that the bias-gradient partial sums stay in registers; per-workgroup
// partials [nwg][heads][L][L] are added in fixed order by tattn_dbias_final_kernel.
struct TattnBwdParams {
    const float* qkv;      // [rows][3 * heads * 32]
    const float* dout;     // [rows][heads * 32]
    float* dqkv;           // [rows][3 * heads * 32]
    float* dbias_part;     // [nwg][heads][L][L] or null
    int heads, L;
    long long n_seq, seq_inner, seq_outer_stride, seq_inner_stride, token_stride;
    const float* rot_cos;  // [L][32] or null
    const float* rot_sin;
    const float* bias;     // [heads][L][L] or null
};

template <int LP>
__global__ __launch_bounds__(256) void tattn_bwd_kernel(const TattnBwdParams p) {
    constexpr int G = 256 / LP;                 // sequences per workgroup
    extern __shared__ float s_tb[];
    const int tid = threadIdx.x, sq_i = tid / LP, tok = tid % LP;
    const int head = blockIdx.y;
    const int L = p.L, ld = 3 * p.heads * 32, HD = p.heads * 32;
    const float scale = 0.17677669529663687f;
    float* base = s_tb + sq_i * (4 * LP * 33 + 3 * LP);
    float* sq = base;                  // [LP][33] q' (scaled, rotated)
    float* sk = sq + LP * 33;          // k' (rotated)
    float* sv = sk + LP * 33;
    float* sd = sv + LP * 33;          // dO
    float* sm = sd + LP * 33;          // row max, row sum, t_i
    float* sl = sm + LP;
    float* st = sl + LP;
    float db[LP];                      // bias-gradient partial of (query ii, key = this thread's token), summed over this thread's sequences
#pragma unroll
    for (int i = 0; i < LP; ++i) db[i] = 0.f;
    const long long ngroups = (p.n_seq + G - 1) / G;
    const bool act_tok = tok < L;
    float* s_rc = s_tb + G * (4 * LP * 33 + 3 * LP);      // rotary tables [LP][32] x 2 behind the sequence areas
    float* s_rs = s_rc + LP * 32;
    for (int idx = tid; idx < LP * 32; idx += 256) {
        const bool in = p.rot_cos && idx < L * 32;
        s_rc[idx] = in ? p.rot_cos[idx] : 1.f;
        s_rs[idx] = in ? p.rot_sin[idx] : 0.f;
    }
    const float* rc = s_rc + tok * 32;
    const float* rs = s_rs + tok * 32;
    for (long long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long long seq = grp * G + sq_i;
        const bool act = seq < p.n_seq && act_tok;
        const long long base_row = act ? (seq / p.seq_inner) * p.seq_outer_stride + (seq % p.seq_inner) * p.seq_inner_stride +
                                             (long long)tok * p.token_stride : 0;
        __syncthreads();               // (previous group's readers are done)
        {
            float q[32], k[32];
            const float* row = p.qkv + base_row * ld + head * 32;
#pragma unroll
            for (int d4 = 0; d4 < 8; ++d4) {
                const f32x4 a = act ? *reinterpret_cast<const f32x4*>(row + 4 * d4) : f32x4{0, 0, 0, 0};
                const f32x4 b = act ? *reinterpret_cast<const f32x4*>(row + HD + 4 * d4) : f32x4{0, 0, 0, 0};
                const f32x4 c = act ? *reinterpret_cast<const f32x4*>(row + 2 * HD + 4 * d4) : f32x4{0, 0, 0, 0};
                const f32x4 e = act ? *reinterpret_cast<const f32x4*>(p.dout + base_row * HD + head * 32 + 4 * d4) : f32x4{0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    q[4 * d4 + u] = a[u] * scale;
                    k[4 * d4 + u] = b[u];
                    sv[tok * 33 + 4 * d4 + u] = c[u];
                    sd[tok * 33 + 4 * d4 + u] = e[u];
                }
            }
            // t cos + rotate_half(t) sin, rotate_half(t)[2m] = -t[2m+1], [2m+1] = t[2m]
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                sq[tok * 33 + 2 * m] = q[2 * m] * rc[2 * m] - q[2 * m + 1] * rs[2 * m];
                sq[tok * 33 + 2 * m + 1] = q[2 * m + 1] * rc[2 * m + 1] + q[2 * m] * rs[2 * m + 1];
                sk[tok * 33 + 2 * m] = k[2 * m] * rc[2 * m] - k[2 * m + 1] * rs[2 * m];
                sk[tok * 33 + 2 * m + 1] = k[2 * m + 1] * rc[2 * m + 1] + k[2 * m] * rs[2 * m + 1];
            }
        }
        __syncthreads();
        // ---- pass Q: thread = query i
        {
            const int i = tok;
            float q[32], dO[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) { q[d] = sq[i * 33 + d]; dO[d] = sd[i * 33 + d]; }
            const float* brow = p.bias ? p.bias + ((long long)head * L + (act_tok ? i : 0)) * L : nullptr;
            float m = -INFINITY;
            for (int j = 0; j < L; ++j) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) s += q[d] * sk[j * 33 + d];
                if (brow) s += brow[j];
                m = fmaxf(m, s);
            }
            float l = 0.f, t = 0.f;
            for (int j = 0; j < L; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) { s += q[d] * sk[j * 33 + d]; dp += dO[d] * sv[j * 33 + d]; }
                if (brow) s += brow[j];
                const float e = expf(s - m);
                l += e;
                t += e * dp;
            }
            t /= l;
            sm[i] = m; sl[i] = l; st[i] = t;
            float dq[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) dq[d] = 0.f;
            for (int j = 0; j < L; ++j) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) { s += q[d] * sk[j * 33 + d]; dp += dO[d] * sv[j * 33 + d]; }
                if (brow) s += brow[j];
                const float ds = (expf(s - m) / l) * (dp - t);
#pragma unroll
                for (int d = 0; d < 32; ++d) dq[d] += ds * sk[j * 33 + d];
            }
            if (act) {
                float* o = p.dqkv + base_row * ld + head * 32;
                // rot^T: dt[2m] = c[2m] g[2m] + s[2m+1] g[2m+1];  dt[2m+1] = c[2m+1] g[2m+1] - s[2m] g[2m]
#pragma unroll
                for (int m2 = 0; m2 < 16; ++m2) {
                    o[2 * m2] = (rc[2 * m2] * dq[2 * m2] + rs[2 * m2 + 1] * dq[2 * m2 + 1]) * scale;
                    o[2 * m2 + 1] = (rc[2 * m2 + 1] * dq[2 * m2 + 1] - rs[2 * m2] * dq[2 * m2]) * scale;
                }
            }
        }
        __syncthreads();
        // ---- pass K: thread = key j
        {
            const int j = tok;
            float k[32], v[32], dk[32], dv[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) { k[d] = sk[j * 33 + d]; v[d] = sv[j * 33 + d]; dk[d] = 0.f; dv[d] = 0.f; }
#pragma unroll 1
            for (int ii = 0; ii < L; ++ii) {
                float s = 0.f, dp = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) { s += sq[ii * 33 + d] * k[d]; dp += sd[ii * 33 + d] * v[d]; }
                if (p.bias && act_tok) s += p.bias[((long long)head * L + ii) * L + j];
                const float pij = expf(s - sm[ii]) / sl[ii];
                const float ds = act ? pij * (dp - st[ii]) : 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) { dk[d] += ds * sq[ii * 33 + d]; dv[d] += pij * sd[ii * 33 + d]; }
                // (static register index: db is fully unrolled below through the switch-free trick of a rotating copy)
#pragma unroll
                for (int u = 0; u < LP; ++u) db[u] += (u == ii) ? ds : 0.f;
            }
            if (act) {
                float* o = p.dqkv + base_row * ld + head * 32;
#pragma unroll
                for (int m2 = 0; m2 < 16; ++m2) {
                    o[HD + 2 * m2] = rc[2 * m2] * dk[2 * m2] + rs[2 * m2 + 1] * dk[2 * m2 + 1];
                    o[HD + 2 * m2 + 1] = rc[2 * m2 + 1] * dk[2 * m2 + 1] - rs[2 * m2] * dk[2 * m2];
                }
#pragma unroll
                for (int d = 0; d < 32; ++d) o[2 * HD + d] = dv[d];
            }
        }
    }
    if (p.dbias_part) {
        // reduce db over the G sequences of the workgroup (fixed order) and store this workgroup's partial
        __syncthreads();
        float* red = s_tb;             // [G][LP][LP + 1]
#pragma unroll
        for (int ii = 0; ii < LP; ++ii) red[(sq_i * LP + ii) * (LP + 1) + tok] = db[ii];
        __syncthreads();
        for (int idx = tid; idx < L * L; idx += 256) {
            const int ii = idx / L, j = idx % L;
            float s = 0.f;
            for (int g = 0; g < G; ++g) s += red[(g * LP + ii) * (LP + 1) + j];
            p.dbias_part[(((long long)blockIdx.x * p.heads + head) * L + ii) * L + j] = s;
        }
    }
}

// The same backward for L <= 32 tokens on the fp32 matrix cores: one WAVE per (sequence, head), persistent, wave-private LDS
// (q', k', v | P^T, dO, dS^T as [32][33] fp32); every product is a 32 x 32 x 32 fp32 GEMM = 16 v_mfma_f32_32x32x2_f32:
//   S^T = K' Q'^T (+ bias^T)  -> softmax over the key axis = over a lane's 16 registers + its partner lane (l ^ 32)
//   dP^T = V dO^T;  dS^T = P^T (dP^T - t_i);  dV = P^T dO;  dK' = dS^T Q';  dQ' = dS K'
// Scores are kept TRANSPOSED (lane = query) so that the softmax statistics are lane-local; P^T / dS^T pass through LDS once to
// become A operands.  A wave keeps ONE head (bias and rotary rows in registers) and accumulates that head's bias gradient in the
// accumulator layout across its sequences; partials [wave][L][L] are reduced in fixed order by tattn_dbias_final_kernel.
struct TattnBwdMParams {
    TattnBwdParams b;
    int waves_per_head;    // total waves / heads
};

__global__ __launch_bounds__(256) void tattn_bwd_mfma_kernel(const TattnBwdMParams pp) {
    const TattnBwdParams& p = pp.b;
    extern __shared__ float s_tm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int gw = blockIdx.x * 4 + wave;
    const int head = gw % p.heads, wslot = gw / p.heads;
    const int L = p.L, ld = 3 * p.heads * 32, HD = p.heads * 32;
    const float scale = 0.17677669529663687f;
    float* sq = s_tm + wave * (5 * 32 * 33);
    float* sk = sq + 32 * 33;
    float* sv = sk + 32 * 33;          // v, later P^T
    float* sd = sv + 32 * 33;          // dO
    float* st = sd + 32 * 33;          // dS^T
    auto rowof = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * hh; };
    // per-lane constants: bias^T column (query i = l31, keys = this lane's rows), rotary rows for the output tokens
    float bias_r[16], dbias_r[16], crow[16], srow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = rowof(r);
        bias_r[r] = (j < L && l31 < L) ? (p.bias ? p.bias[((long long)head * L + l31) * L + j] : 0.f) : -INFINITY;
        if (j < L && l31 >= L) bias_r[r] = 0.f;                    // inactive query columns: finite scores, zero gradients
        dbias_r[r] = 0.f;
        const int tok = j;                                         // output tiles: rows = tokens, lane = head dim d = l31
        const bool in = p.rot_cos && tok < L;
        crow[r] = in ? p.rot_cos[tok * 32 + l31] : 1.f;
        // rot^T: out[d] = c[d] g[d] + s[d ^ 1] g[d ^ 1] (d even) | c[d] g[d] - s[d ^ 1] g[d ^ 1] (d odd)
        srow[r] = in ? ((l31 & 1) ? -p.rot_sin[tok * 32 + (l31 ^ 1)] : p.rot_sin[tok * 32 + (l31 ^ 1)]) : 0.f;
    }
    for (long long seq = wslot; seq < p.n_seq; seq += pp.waves_per_head) {
        const long long base_row = (seq / p.seq_inner) * p.seq_outer_stride + (seq % p.seq_inner) * p.seq_inner_stride;
        // ---- stage q' (scaled, rotated), k' (rotated), v, dO: lane -> (token = idx / 8, 4 dims)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = lane + 64 * u, tok = idx >> 3, c4 = (idx & 7) * 4;
            const bool in = tok < L;
            const long long row = base_row + (long long)tok * p.token_stride;
            const float* src = p.qkv + row * ld + head * 32 + c4;
            f32x4 q = in ? *reinterpret_cast<const f32x4*>(src) : f32x4{0, 0, 0, 0};
            f32x4 k = in ? *reinterpret_cast<const f32x4*>(src + HD) : f32x4{0, 0, 0, 0};
            const f32x4 v = in ? *reinterpret_cast<const f32x4*>(src + 2 * HD) : f32x4{0, 0, 0, 0};
            const f32x4 d = in ? *reinterpret_cast<const f32x4*>(p.dout + row * HD + head * 32 + c4) : f32x4{0, 0, 0, 0};
            q = q * scale;
            if (p.rot_cos && in) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.rot_cos + tok * 32 + c4);
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.rot_sin + tok * 32 + c4);
                q = f32x4{q[0] * c[0] - q[1] * s[0], q[1] * c[1] + q[0] * s[1], q[2] * c[2] - q[3] * s[2], q[3] * c[3] + q[2] * s[3]};
                k = f32x4{k[0] * c[0] - k[1] * s[0], k[1] * c[1] + k[0] * s[1], k[2] * c[2] - k[3] * s[2], k[3] * c[3] + k[2] * s[3]};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sq[tok * 33 + c4 + e] = q[e];
                sk[tok * 33 + c4 + e] = k[e];
                sv[tok * 33 + c4 + e] = v[e];
                sd[tok * 33 + c4 + e] = d[e];
            }
        }
        // (wave-private LDS: DS operations of a wave complete in order, no barrier needed)
        f32x16 sT, dpT;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sT[r] = bias_r[r]; dpT[r] = 0.f; }
        // (B operands come straight from LDS, one register per MFMA: they are read up front and each keeps its register until four
        //  younger MFMAs were issued -- common.h: mfma_keep_a / mfma_order_point)
        {
            float bq1[16], bd1[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                bq1[s2] = sq[l31 * 33 + 2 * s2 + hh];
                bd1[s2] = sd[l31 * 33 + 2 * s2 + hh];
            }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int kk = 2 * s2 + hh;
                sT = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[l31 * 33 + kk], bq1[s2], sT, 0, 0, 0);
                if (s2 >= 2) mfma_keep_a(sT, bq1[s2 - 2]);
                dpT = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[l31 * 33 + kk], bd1[s2], dpT, 0, 0, 0);
                if (s2 >= 2) mfma_keep_a(dpT, bd1[s2 - 2]);
                mfma_order_point();
            }
            mfma_drain(sT);                                      // (consumed by the softmax right below)
            mfma_drain(dpT);
        }
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, sT[r]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sT[r] = expf(sT[r] - m); l += sT[r]; }
        l += __shfl_xor(l, 32, 64);
        const float il = 1.0f / l;
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sT[r] *= il; t += sT[r] * dpT[r]; }
        t += __shfl_xor(t, 32, 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dpT[r] = sT[r] * (dpT[r] - t);                       // dS^T
            dbias_r[r] += dpT[r];
            sv[rowof(r) * 33 + l31] = sT[r];                      // P^T [j][i] over v (its last MFMA read is issued above)
            st[rowof(r) * 33 + l31] = dpT[r];
        }
        f32x16 dv, dk, dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dv[r] = 0.f; dk[r] = 0.f; dq[r] = 0.f; }
        {
            float bd2[16], bq2[16], bk2[16];
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {
                const int kk = 2 * s2 + hh;
                bd2[s2] = sd[kk * 33 + l31]; bq2[s2] = sq[kk * 33 + l31]; bk2[s2] = sk[kk * 33 + l31];
            }
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) {                    // three chains: a B register is released two iterations (6 MFMAs) later
                const int kk = 2 * s2 + hh;
                dv = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[l31 * 33 + kk], bd2[s2], dv, 0, 0, 0);       // P^T[j][i] dO[i][d]
                if (s2 >= 2) mfma_keep_a(dv, bd2[s2 - 2]);
                dk = __builtin_amdgcn_mfma_f32_32x32x2f32(st[l31 * 33 + kk], bq2[s2], dk, 0, 0, 0);       // dS^T[j][i] q'[i][d]
                if (s2 >= 2) mfma_keep_a(dk, bq2[s2 - 2]);
                dq = __builtin_amdgcn_mfma_f32_32x32x2f32(st[kk * 33 + l31], bk2[s2], dq, 0, 0, 0);       // dS[i][j] k'[j][d]
                if (s2 >= 2) mfma_keep_a(dq, bk2[s2 - 2]);
                mfma_order_point();
            }
            mfma_drain(dv);
            mfma_drain(dk);
            mfma_drain(dq);
        }
        // ---- rot^T (pairs are neighbouring lanes), q scale, stores: rows = tokens, 32 lanes = one 128-byte row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tok = rowof(r);
            const float gq = dq[r], gk = dk[r];
            const float pq = __shfl_xor(gq, 1, 64), pk = __shfl_xor(gk, 1, 64);
            if (tok < L) {
                float* o = p.dqkv + (base_row + (long long)tok * p.token_stride) * ld + head * 32 + l31;
                o[0] = (crow[r] * gq + srow[r] * pq) * scale;
                o[HD] = crow[r] * gk + srow[r] * pk;
                o[2 * HD] = dv[r];
            }
        }
    }
    if (p.dbias_part) {
        float* dst = p.dbias_part + ((long long)wslot * p.heads + head) * L * L;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = rowof(r);
            if (j < L && l31 < L) dst[l31 * L + j] = dbias_r[r];
        }
    }
}

__global__ __launch_bounds__(256) void tattn_dbias_final_kernel(const float* __restrict__ part, float* __restrict__ dbias, int nwg, int n,
                                                               int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0;
    for (int k = 0; k < nwg; ++k) s += (double)part[(long long)k * n + i];
    dbias[i] = accumulate ? dbias[i] + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------ q_sample / loss / optimizer
// p_losses :809-816: state = a[t_b] x0 + b[t_b] noise; state[:, 0, 0] = x0[:, 0, 0]; target = noise with [:, 0, 0] = 0.
// x0 is a channel slice [coff, coff + C) of a [B][F][Ctot][H][W] tensor (the w model trains on state[:, :, 3:5], Trainer :1018-1019).
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                      const long long* __restrict__ t, const float* __restrict__ sqrt_ac,
                                                      const float* __restrict__ sqrt_1mac, float* __restrict__ state,
                                                      float* __restrict__ target, long long total, int F, int C, int Ctot, int coff,
                                                      int HW) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        long long r = i / HW;
        const int c = (int)(r % C);
        r /= C;
        const int f = (int)(r % F);
        const long long b = r / F;
        const float x = x0[((b * F + f) * Ctot + coff + c) * (long long)HW + hw];
        const float n = noise[i];
        const long long tb = t[b];
        const bool cond = f == 0 && c == 0;
        const float a = sqrt_ac[tb] * x, bn = sqrt_1mac[tb] * n;
        state[i] = cond ? x : a + bn;
        target[i] = cond ? 0.f : n;
    }
}

// mse (reduction 'mean'): part[blk] = sum (out - target)^2 over the block's elements (fp64), dout = gscale * 2 (out - target) / n
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ out, const float* __restrict__ target, float* __restrict__ dout,
                                                      double* __restrict__ part, long long n, float gmul) {
    __shared__ double red[256];
    double s = 0;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float d = out[i] - target[i];
        s += (double)d * (double)d;
        if (dout) dout[i] = d * gmul;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, double* __restrict__ part, long long n) {
    __shared__ double red[256];
    double s = 0;
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) s += (double)x[i] * (double)x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// out[0] = (float)(mul * sum_k part[k]) (sqrt_it: sqrt of the sum first) -- one thread, fixed order
__global__ void reduce_final_kernel(const double* __restrict__ part, int n, double mul, int sqrt_it, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0;
    for (int k = 0; k < n; ++k) s += part[k];
    if (sqrt_it) s = sqrt(s);
    out[0] = (float)(s * mul);
}

// Trainer.train :1027-1043 on flat parameter buffers: clip_grad_norm_ (coefficient from the device-resident total norm),
// torch.optim.Adam's update (lerp / addcmul / addcdiv association, eps 1e-8, bias corrections passed by value), then the EMA of
// ema-pytorch 0.7.3 (ema_mode 0 skip | 1 copy | 2 lerp | 3 copy then lerp, weight ema_w).  g is read as g * ginv (the inverse of
// the loss scale, a power of two) before clipping.
// The scalars arrive as torch forms them: 1 - beta, lr / (1 - beta1^step) and sqrt(1 - beta2^step) are evaluated in DOUBLE on the host
// (Python floats in torch.optim.adam._single_tensor_adam) and rounded to fp32 once, when they meet the fp32 tensors.
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, float* __restrict__ ema, long long n,
                                                      const float* __restrict__ total_norm, float max_norm, float ginv, float step_size,
                                                      float omb1, float beta2, float omb2, float eps, float sqrt_bc2, int ema_mode,
                                                      float ema_w) {
    float coef = 1.f;
    if (total_norm && max_norm > 0.f) coef = fminf(max_norm / (total_norm[0] + 1e-6f), 1.0f);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = (g[i] * ginv) * coef;
        float mi = m[i], vi = v[i], wi = w[i];
        mi = mi + omb1 * (gi - mi);                                // exp_avg.lerp_(grad, 1 - beta1)         (weight < 0.5 branch of lerp)
        vi = vi * beta2 + (omb2 * gi) * gi;                        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2):  a + alpha * b * c
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;            // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        wi = wi + (-step_size * mi) / denom;                       // param.addcdiv_(exp_avg, denom, value = -step_size): a + alpha * b / c,
                                                                   // the CPU kernel's association (the fixtures' reference run)
        m[i] = mi; v[i] = vi; w[i] = wi;
        if (ema_mode) {
            float e = ema[i];
            if (ema_mode & 1) e = wi;
            if (ema_mode & 2) e = e + ema_w * (wi - e);            // lerp_(current, 1 - decay)
            ema[i] = e;
        }
    }
}

// Backward of dpc_small_linear (out = bias + act(in) W^T; time_mlp ...conv3d.py:404-409, ResnetBlock.mlp :209-212), B <= a few
// hundred rows: one thread per output element, the batch / feature loop inside (fixed order).
//   dW[n][k] = sum_b dy[b][n] act(x[b][k]);  db[n] = sum_b dy[b][n];  dx[b][k] (+)= act'(x[b][k]) sum_n dy[b][n] W[n][k]
__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == 1) return v / (1.0f + expf(-v));
    if (act == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    return v;
}
__device__ __forceinline__ float act_grad(float v, int act) {
    if (act == 1) {
        const float sg = 1.0f / (1.0f + expf(-v));
        return sg * (1.0f + v * (1.0f - sg));
    }
    if (act == 2) return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * expf(-0.5f * v * v) * 0.3989422804014327f;
    return 1.0f;
}
__global__ __launch_bounds__(256) void small_linear_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ W, float* __restrict__ dx, float* __restrict__ dW,
                                                              float* __restrict__ db, int B, int K, int N, int in_act, int accum_dx) {
    const long long nW = (long long)N * K, nX = dx ? (long long)B * K : 0, nB = db ? N : 0;
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < nW) {
        const int n = (int)(i / K), k = (int)(i % K);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(long long)b * N + n] * act_fwd(x[(long long)b * K + k], in_act);
        dW[i] = s;
    } else if (i < nW + nX) {
        const long long j = i - nW;
        const int b = (int)(j / K), k = (int)(j % K);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dy[(long long)b * N + n] * W[(long long)n * K + k];
        s *= act_grad(x[j], in_act);
        dx[j] = accum_dx ? dx[j] + s : s;
    } else if (i < nW + nX + nB) {
        const int n = (int)(i - nW - nX);
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(long long)b * N + n];
        db[n] = s;
    }
}

static inline unsigned grid1d(long long total) { return (unsigned)std::min<long long>((total + 255) / 256, 256 * 8); }

}  // namespace dpc

using namespace dpc;

extern "C" {

int dpc_train_range_status(int reset, dpc_stream_t stream) { return f16x3_grad_overflow_status(reset, (hipStream_t)stream); }
int dpc_train_range_poison(float* g, dpc_stream_t stream) { return f16x3_grad_overflow_poison(g, (hipStream_t)stream); }

size_t dpc_conv_wgrad_workspace_bytes(int C, int N, int kf, int kh, int kw, int64_t rows) {
    const int fpr = (kw * C + 31) / 32;
    const int nrf = kf * kh * fpr, ncf = (N + 31) / 32;
    size_t need = (size_t)wgrad_slabs(nrf, ncf, rows) * nrf * 32 * ncf * 32 * sizeof(float) + 512;
    if (kf == 3 && kh == 3 && kw == 3 && C % 32 == 0 && N % 64 == 0) need = std::max(need, wgrad3_workspace_bytes(C, N, rows));
    return need;
}

int dpc_conv_wgrad_cl(const float* x, const float* dy, float* dw, int B, int F, int Hi, int Wi, int C, int Ho, int Wo, int N, int kf,
                      int kh, int kw, int sh, int sw, int pf, int ph, int pw, int c_valid, int dw_ctot, int dw_coff, float scale,
                      float f16_dy_scale, float dy_abs_limit, int accumulate, void* ws, size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(x && dy && dw && ws, "conv_wgrad: null argument");
    DPC_REQUIRE((accumulate & ~3) == 0, "conv_wgrad: accumulate is a bit set (1: add to dw, 2: x is the gradient operand)");
    const int x_is_grad = (accumulate >> 1) & 1;
    accumulate &= 1;
    DPC_REQUIRE(dy_abs_limit >= 0.f, "conv_wgrad: dy_abs_limit must be >= 0 (0 = no check)");
    DPC_REQUIRE(B >= 1 && F >= 1 && C >= 1 && N >= 1 && kf >= 1 && kh >= 1 && kw >= 1 && sh >= 1 && sw >= 1, "conv_wgrad: bad shape");
    DPC_REQUIRE(C % 32 == 0 || 32 % C == 0, "conv_wgrad: input channels must divide or be a multiple of 32 (pad on the host)");
    DPC_REQUIRE((long long)Wi * C * 4 < (1ll << 31) && (long long)Wo * N * 4 < (1ll << 31), "conv_wgrad: row too long");
    if (c_valid <= 0) c_valid = C;
    DPC_REQUIRE(c_valid <= C && dw_coff >= 0 && dw_coff + c_valid <= dw_ctot, "conv_wgrad: bad weight channel slice");
    hipStream_t s = (hipStream_t)stream;
    if (f16_dy_scale != 0.f) {
        int e = 0;
        DPC_REQUIRE(f16_dy_scale > 0.f && std::frexp(f16_dy_scale, &e) == 0.5f, "conv_wgrad: f16_dy_scale must be a power of two (or 0)");
        if (kf == 3 && kh == 3 && kw == 3 && sh == 1 && sw == 1 && pf == 1 && ph == 1 && pw == 1 && Hi == Ho && Wi == Wo && c_valid == C &&
            wgrad3_supported(Wi, Hi, C, N)) {
            DPC_REQUIRE(ws_bytes >= wgrad3_workspace_bytes(C, N, (long long)B * F), "conv_wgrad: workspace too small");
            return launch_wgrad3(x, dy, dw, B, F, Hi, Wi, C, N, dw_ctot, dw_coff, 16.0f, f16_dy_scale, scale, accumulate, ws, s);
        }
    }
    WgradParams p{};
    p.x = x; p.dy = dy;
    p.B = B; p.F = F; p.Hi = Hi; p.Wi = Wi; p.C = C; p.Ho = Ho; p.Wo = Wo; p.N = N;
    p.kf = kf; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.pf = pf; p.ph = ph; p.pw = pw;
    p.fpr = (kw * C + 31) / 32;
    p.nrf = kf * kh * p.fpr;
    p.ncf = (N + 31) / 32;
    int WG_RA, WG_RB;
    wgrad_shape(p.nrf, p.ncf, WG_RA, WG_RB);
    p.nru = (p.nrf + WG_RA - 1) / WG_RA;
    p.ncu = (p.ncf + WG_RB - 1) / WG_RB;
    p.rows = (long long)B * F * Ho;
    p.dy_limit_bits = dy_abs_limit > 0.f ? __builtin_bit_cast(unsigned, dy_abs_limit) : 0u;
    p.x_is_grad = x_is_grad;
    p.oflag = p.dy_limit_bits ? f16x3_grad_overflow_flag() : nullptr;
    DPC_REQUIRE(!p.dy_limit_bits || p.oflag, "conv_wgrad: cannot allocate the gradient-range sentinel word");
    p.nslab = wgrad_slabs(p.nrf, p.ncf, p.rows);
    const size_t tile_floats = (size_t)p.nrf * 32 * p.ncf * 32;
    DPC_REQUIRE(ws_bytes >= (size_t)p.nslab * tile_floats * sizeof(float) + 256, "conv_wgrad: workspace too small");
    p.part = reinterpret_cast<float*>(align_up((size_t)ws, 256));
    {
        ProfScope prof(PROF_WGRAD, 2.0 * (double)p.rows * Wo * N * kf * kh * kw * C, 0, s);
        const dim3 grid((p.nru * p.ncu + 3) / 4, p.nslab);
        if (WG_RA == 2) hipLaunchKernelGGL((wgrad_kernel<2, 4>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wgrad_kernel<4, 2>), grid, dim3(256), 0, s, p);
        DPC_LAUNCH_CHECK();
    }
    ProfScope prof(PROF_TRAIN_MISC, 0, (double)p.nslab * tile_floats * 4, s);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)std::min<long long>(((long long)tile_floats + 63) / 64, 4096)), dim3(256), 0, s, p.part, dw, p.nslab, p.nrf, p.ncf, p.fpr,
                       C, c_valid, N, kf, kh, kw, dw_ctot, dw_coff, scale, accumulate);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

size_t dpc_colsum_workspace_bytes(int C) { return (size_t)1024 * C * sizeof(double) + 256; }

int dpc_colsum(const float* dy, const float* x, const float* ln_stats, float* out, int64_t rows, int C, float scale, int accumulate,
               void* ws, size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(dy && out && ws, "colsum: null argument");
    DPC_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0, "colsum: C / 4 must divide 256");
    DPC_REQUIRE(!x == !ln_stats, "colsum: x and ln_stats go together");
    DPC_REQUIRE(ws_bytes >= dpc_colsum_workspace_bytes(C), "colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int rpp = 256 / (C / 4);
    int nblk = (int)std::min<long long>(1024, std::max<long long>(1, rows / ((long long)rpp * 4)));
    double* part = reinterpret_cast<double*>(align_up((size_t)ws, 256));
    ProfScope prof(PROF_TRAIN_MISC, 0, 4.0 * (double)rows * C * (x ? 2 : 1), s);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, s, dy, x, ln_stats, part, (long long)rows, C, nblk);
    DPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, s, part, out, C, nblk, scale, accumulate);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

size_t dpc_attention_bwd_seq_workspace_bytes(int heads, int L) { return (size_t)512 * heads * L * L * sizeof(float) + 256; }

int dpc_attention_bwd_seq(const float* qkv, const float* dout, float* dqkv, float* dbias, int heads, int L, int64_t n_seq,
                          int64_t seq_inner, int64_t seq_outer_stride_rows, int64_t seq_inner_stride_rows, int64_t token_stride_rows,
                          const float* rot_cos, const float* rot_sin, const float* bias, int accumulate_dbias, void* ws, size_t ws_bytes,
                          dpc_stream_t stream) {
    DPC_REQUIRE(qkv && dout && dqkv && heads >= 1 && L >= 1 && L <= 64, "attention_bwd_seq: bad argument (1 <= L <= 64 tokens)");
    DPC_REQUIRE(!rot_cos == !rot_sin, "attention_bwd_seq: rotary tables go together");
    DPC_REQUIRE(!dbias || (ws && ws_bytes >= dpc_attention_bwd_seq_workspace_bytes(heads, L)), "attention_bwd_seq: workspace too small");
    if (n_seq == 0) return DPC_OK;
    hipStream_t s = (hipStream_t)stream;
    static const int use_mfma = debug_switch("DPC_TATTN_BWD_MFMA", 1);
    if (L <= 32 && use_mfma) {
        // one wave per (sequence, head); waves-per-head partial slots for the bias gradient (<= 512 as the workspace is sized)
        long long waves = std::min<long long>((n_seq * heads + 3) / 4 * 4, 256 * 4);
        waves = std::min<long long>(waves, 512ll * heads);                               // (the workspace holds 512 slots per head)
        waves = std::max<long long>(heads * 4, waves / (heads * 4) * (heads * 4));       // multiple of 4 (workgroup) and of heads
        TattnBwdMParams q{};
        q.b.qkv = qkv; q.b.dout = dout; q.b.dqkv = dqkv; q.b.heads = heads; q.b.L = L;
        q.b.n_seq = n_seq; q.b.seq_inner = seq_inner; q.b.seq_outer_stride = seq_outer_stride_rows;
        q.b.seq_inner_stride = seq_inner_stride_rows; q.b.token_stride = token_stride_rows;
        q.b.rot_cos = rot_cos; q.b.rot_sin = rot_sin; q.b.bias = bias;
        q.b.dbias_part = dbias ? reinterpret_cast<float*>(align_up((size_t)ws, 256)) : nullptr;
        q.waves_per_head = (int)(waves / heads);
        DPC_REQUIRE(q.waves_per_head <= 512, "attention_bwd_seq: too many heads for the bias-gradient workspace");
        const size_t lds = (size_t)4 * 5 * 32 * 33 * sizeof(float);
        static DeviceOnce once_m;
        if (!once_m) {
            DPC_HIP(hipFuncSetAttribute((const void*)tattn_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            once_m = true;
        }
        {
            ProfScope prof(PROF_ATTN_BWD, 10.0 * (double)n_seq * heads * L * L * 32, 0, s);
            hipLaunchKernelGGL(tattn_bwd_mfma_kernel, dim3((unsigned)(waves / 4)), dim3(256), lds, s, q);
            DPC_LAUNCH_CHECK();
        }
        if (dbias) {
            const int n = heads * L * L;
            hipLaunchKernelGGL(tattn_dbias_final_kernel, dim3((n + 255) / 256), dim3(256), 0, s, q.b.dbias_part, dbias, q.waves_per_head, n,
                               accumulate_dbias);
            DPC_LAUNCH_CHECK();
        }
        return DPC_OK;
    }
    const int LP = L <= 32 ? 32 : 64, G = 256 / LP;
    const long long ngroups = (n_seq + G - 1) / G;
    const int nwg = (int)std::min<long long>(ngroups, 512);
    TattnBwdParams p{};
    p.qkv = qkv; p.dout = dout; p.dqkv = dqkv; p.heads = heads; p.L = L;
    p.n_seq = n_seq; p.seq_inner = seq_inner; p.seq_outer_stride = seq_outer_stride_rows; p.seq_inner_stride = seq_inner_stride_rows;
    p.token_stride = token_stride_rows; p.rot_cos = rot_cos; p.rot_sin = rot_sin; p.bias = bias;
    p.dbias_part = dbias ? reinterpret_cast<float*>(align_up((size_t)ws, 256)) : nullptr;
    const size_t lds = ((size_t)G * (4 * LP * 33 + 3 * LP) + 2 * LP * 32) * sizeof(float);     // (>= the [G][LP][LP + 1] reduction area)
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)tattn_bwd_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DPC_HIP(hipFuncSetAttribute((const void*)tattn_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once = true;
    }
    {
        ProfScope prof(PROF_ATTN_BWD, 10.0 * (double)n_seq * heads * L * L * 32, 0, s);
        if (LP == 32) hipLaunchKernelGGL(tattn_bwd_kernel<32>, dim3(nwg, heads), dim3(256), lds, s, p);
        else hipLaunchKernelGGL(tattn_bwd_kernel<64>, dim3(nwg, heads), dim3(256), lds, s, p);
        DPC_LAUNCH_CHECK();
    }
    if (dbias) {
        const int n = heads * L * L;
        hipLaunchKernelGGL(tattn_dbias_final_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p.dbias_part, dbias, nwg, n, accumulate_dbias);
        DPC_LAUNCH_CHECK();
    }
    return DPC_OK;
}

int dpc_q_sample_smoke(const float* x0, int x_channels_total, int x_channel_offset, const float* noise, const int64_t* t,
                       const float* sqrt_alphas_cumprod, const float* sqrt_one_minus_alphas_cumprod, float* state, float* target, int B,
                       int F, int C, int H, int W, dpc_stream_t stream) {
    DPC_REQUIRE(x0 && noise && t && sqrt_alphas_cumprod && sqrt_one_minus_alphas_cumprod && state && target, "q_sample: null argument");
    DPC_REQUIRE(x_channel_offset >= 0 && x_channel_offset + C <= x_channels_total, "q_sample: bad channel slice");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)B * F * C * H * W;
    if (total == 0) return DPC_OK;
    ProfScope prof(PROF_TRAIN_MISC, 0, 16.0 * (double)total, s);
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid1d(total)), dim3(256), 0, s, x0, noise, reinterpret_cast<const long long*>(t),
                       sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod, state, target, total, F, C, x_channels_total, x_channel_offset,
                       H * W);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

size_t dpc_reduce_workspace_bytes(void) { return 2048 * sizeof(double) + 256; }

int dpc_mse_loss_grad(const float* out, const float* target, float* dout, float* loss, int64_t n, float grad_scale, void* ws,
                      size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(out && target && loss && ws && n >= 1, "mse_loss_grad: null argument");
    DPC_REQUIRE(ws_bytes >= dpc_reduce_workspace_bytes(), "mse_loss_grad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(align_up((size_t)ws, 256));
    const int nblk = (int)std::min<long long>(2048, (n + 1023) / 1024);
    ProfScope prof(PROF_TRAIN_MISC, 0, 12.0 * (double)n, s);
    hipLaunchKernelGGL(mse_grad_kernel, dim3(nblk), dim3(256), 0, s, out, target, dout, part, (long long)n, grad_scale * 2.0f / (float)n);
    DPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, s, part, nblk, 1.0 / (double)n, 0, loss);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_l2_norm(const float* x, int64_t n, float scale, float* out, void* ws, size_t ws_bytes, dpc_stream_t stream) {
    DPC_REQUIRE(x && out && ws && n >= 1, "l2_norm: null argument");
    DPC_REQUIRE(ws_bytes >= dpc_reduce_workspace_bytes(), "l2_norm: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(align_up((size_t)ws, 256));
    const int nblk = (int)std::min<long long>(2048, (n + 1023) / 1024);
    ProfScope prof(PROF_TRAIN_MISC, 0, 4.0 * (double)n, s);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, s, x, part, (long long)n);
    DPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(64), 0, s, part, nblk, (double)scale, 1, out);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_small_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int B, int K, int N,
                         int in_act, int accumulate_dx, dpc_stream_t stream) {
    DPC_REQUIRE(dy && x && W && dW && B >= 1 && K >= 1 && N >= 1 && in_act >= 0 && in_act <= 2, "small_linear_bwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)N * K + (dx ? (long long)B * K : 0) + (db ? N : 0);
    ProfScope prof(PROF_TRAIN_MISC, 0, 0, s);
    hipLaunchKernelGGL(small_linear_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dy, x, W, dx, dW, db, B, K, N, in_act,
                       accumulate_dx);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_adam_ema_step(float* w, const float* g, float* m, float* v, float* ema, int64_t n, const float* total_norm, float max_norm,
                      float grad_inv_scale, double lr, double beta1, double beta2, double eps, int step, int ema_mode, float ema_weight,
                      dpc_stream_t stream) {
    DPC_REQUIRE(w && g && m && v && n >= 1 && step >= 1, "adam_ema_step: bad argument");
    DPC_REQUIRE(ema_mode >= 0 && ema_mode <= 3 && (ema_mode == 0 || ema), "adam_ema_step: bad ema mode");
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - std::pow(beta1, step), bc2 = 1.0 - std::pow(beta2, step);
    ProfScope prof(PROF_TRAIN_MISC, 0, (ema_mode ? 36.0 : 28.0) * (double)n, s);
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid1d(n)), dim3(256), 0, s, w, g, m, v, ema, (long long)n, total_norm, max_norm, grad_inv_scale,
                       (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)std::sqrt(bc2), ema_mode,
                       ema_weight);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // extern "C"
