// Weight gradient of the 3x3x3 stride-1 pad-1 convolutions (Block.proj, ...conv3d.py:189-204) on the fp16 matrix cores:
//   dW[n][c][df][dh][dw] = sum over points p = (b, f, h, w) of dy[p][n] * x[b][f + df - 1][h + dh - 1][w + dw - 1][c]
// -- 95 % of the weight-gradient FLOP of the smoke denoiser's training step.
//
// GEMM view per tap: dW_tap [ci][co] = X_tap^T [ci][points] * dY [points][co]: the reduction axis is the POINT axis, the slow axis of
// both channels-last operands, while a 16-bit MFMA operand wants 8 consecutive reduction indices per lane.  gfx950's LDS transpose
// read (ds_read_b64_tr_b16) delivers exactly that from a plain channels-last fp16 image: a 16-lane group reads a [4 points][16
// channels] block and each lane receives one channel's 4 points.  So the tiles are staged in LDS as [point][32 channels] fp16 (64
// bytes per point: conflict-free for the transpose read), as TWO planes -- x 2^4 = h1 + h2, h1 = fp16(x), h2 = fp16(x - h1), 22
// significant bits, the f16x3 arithmetic of the forward convolutions -- and every product costs 3 MFMAs (h1 h1 + h1 h2 + h2 h1),
// fp32 accumulation.  The gradient operand is pre-scaled by a power of two chosen by the caller (loss scale) and saturated.
//
// Work decomposition: a workgroup owns a UNIT = (32 input channels) x (64 output channels) x all 27 taps = 54 accumulator tiles
// of 32 x 32, held in registers by its 4 waves (wave w: taps 7w .. 7w + 6, both column tiles: 14 tiles = 224 accumulator
// registers), and streams a slab of (b, f) planes through LDS: per iteration 64 output points (64 / W rows) = 4 MFMA k-blocks.
// The input rows live in three rolling rings (frame offsets -1, 0, +1), 4 row groups deep: group it + 2 is fetched from HBM while
// group it is multiplied; one barrier per iteration.  Taps that fall outside the tensor read a zero row (no branches in the loop).
// Partial sums per (slab, unit) are added in fixed order by wgrad3_reduce_kernel, which also undoes the operand scales and writes
// the reference layout.  Workgroups of one slab sit on the same XCD so that its L2 serves their shared reads.
#include <mutex>
#include <algorithm>

#include "common.h"
#include "f16x3.h"

namespace dpc {

typedef short v4s __attribute__((__vector_size__(8)));
typedef short v8s __attribute__((__vector_size__(16)));
typedef __attribute__((address_space(3))) v4s* lds_v4s_ptr;

struct Wgrad3Params {
    const float* x;            // [B][F][H][W][Cx], this source's channels
    const float* dy;           // [B][F][H][W][N]
    float* part;               // [nslab][nunits][27][32][64]
    int Cx, N, F, H;
    int n_ci, n_co;            // units: Cx / 32 input-channel tiles x N / 64 output-channel blocks
    int nslab;
    long long planes;          // B * F
    float x_scale, dy_scale;
    int* oflag;                // gradient-range sentinel (f16x3_grad_overflow_flag): bit 0 = an activation, bit 1 = an output gradient left the
                               // fp16 window after its pre-scale (it was clamped), or was not finite
};

constexpr unsigned F16_MAX_BITS = 0x477FE000u;          // 65504.0f: |bits| above it (Inf / NaN included) do not fit an fp16 operand

template <int W>
struct Wg3Cfg {
    static constexpr int RPI = 64 / W;                   // output rows per iteration
    static constexpr int NS = 4 * RPI;                   // ring slots (rows) per frame offset: 4 groups
    static constexpr int ROWP = W + 2;                   // halo points per row
    static constexpr int PLANE = ROWP * 64;              // bytes of one fp16 plane of a row [ROWP][32]
    static constexpr int SLOT = 2 * PLANE;
    static constexpr int DFS = NS * SLOT;                // bytes per frame-offset ring
    static constexpr int XBYTES = 3 * DFS;
    static constexpr int ZERO_OFF = XBYTES;              // a zero row (both planes read the same zeros)
    static constexpr int Y_OFF = ZERO_OFF + PLANE;
    static constexpr int YPLANE = 64 * 64;               // [64 points][32 co] fp16
    static constexpr int YBUF = 4 * YPLANE;              // [co half 2][plane 2]
    static constexpr int TOTAL = Y_OFF + 2 * YBUF;
};

__device__ __forceinline__ h3::f16x8 tr_frag(lds_v4s_ptr p0, lds_v4s_ptr p1) {
    const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
    const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
    return __builtin_bit_cast(h3::f16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int W>
__global__ __launch_bounds__(256, 1) void wgrad3_kernel(const Wgrad3Params p) {
    using Cf = Wg3Cfg<W>;
    constexpr int RPI = Cf::RPI, NS = Cf::NS, KPR = W / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- which (slab, unit): the units of a slab are neighbours on one XCD (block id % 8 = XCD)
    const int nunits = p.n_ci * p.n_co;
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int unit = j % nunits, slab = (j / nunits) * 8 + xcd;
    if (slab >= p.nslab) return;
    const int ci_tile = unit / p.n_co, co_blk = unit % p.n_co;
    const long long planes_per = (p.planes + p.nslab - 1) / p.nslab;
    const long long plane0 = slab * planes_per;
    const long long nplanes = min(planes_per, p.planes - plane0);
    const int H = p.H, F = p.F;
    const long long V = nplanes > 0 ? nplanes * H : 0;                   // virtual rows of this slab
    const int ngroups = (int)((V + RPI - 1) / RPI);

    for (int i = tid * 16; i < Cf::TOTAL; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = uint4{0, 0, 0, 0};
    __syncthreads();

    // ---- loader: group G = virtual rows [G RPI, G RPI + RPI) = 64 points; per thread 2 float4 per frame offset + 4 of dY
    f32x4 xr[3][2], yr[4];
    bool xok[3];
    auto issue_x = [&](int G) {
        const long long v0 = (long long)G * RPI;
        const bool in = v0 < V;
        const long long pl = in ? plane0 + v0 / H : 0;
        const int h0 = in ? (int)(v0 % H) : 0;
        const int f = (int)(pl % F);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            xok[d] = in && (unsigned)(f + d - 1) < (unsigned)F;
            if (xok[d]) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = tid + 256 * u, pt = q >> 3, c4 = q & 7;
                    const int r = pt / W, w = pt % W;
                    const float* src = p.x + (((pl + d - 1) * H + h0 + r) * W + w) * (long long)p.Cx + ci_tile * 32 + c4 * 4;
                    xr[d][u] = *reinterpret_cast<const f32x4*>(src);
                }
            }
        }
    };
    auto store_x = [&](int G) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (!xok[d]) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = tid + 256 * u, pt = q >> 3, c4 = q & 7;
                const int r = pt / W, w = pt % W;
                const int slot = (G * RPI + r) & (NS - 1);
                f32x4 v = xr[d][u] * p.x_scale;
                if (max(max(abs_bits(v[0]), abs_bits(v[1])), max(abs_bits(v[2]), abs_bits(v[3]))) > F16_MAX_BITS) atomicOr(p.oflag, 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = h3::sat16(v[e]);
                uint2 hi, lo;
                hi.x = h3::cvt_pk(v[0], v[1]); hi.y = h3::cvt_pk(v[2], v[3]);
                lo.x = f16_sub_pk(v[0], v[1], hi.x); lo.y = f16_sub_pk(v[2], v[3], hi.y);
                unsigned char* dst = smem + d * Cf::DFS + slot * Cf::SLOT + (w + 1) * 64 + c4 * 8;
                *reinterpret_cast<uint2*>(dst) = hi;
                *reinterpret_cast<uint2*>(dst + Cf::PLANE) = lo;
            }
        }
    };
    bool yok = false;
    auto issue_y = [&](int G) {
        const long long v0 = (long long)G * RPI;
        yok = v0 < V;
        if (yok) {
            const long long row0 = plane0 * H + v0;                        // rows are contiguous over (plane, h)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = tid + 256 * u, pt = q >> 4, c4 = q & 15;
                yr[u] = *reinterpret_cast<const f32x4*>(p.dy + (row0 * W + pt) * (long long)p.N + co_blk * 64 + c4 * 4);
            }
        }
    };
    auto store_y = [&](int G) {
        if (!yok) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tid + 256 * u, pt = q >> 4, c4 = q & 15;
            f32x4 v = yr[u] * p.dy_scale;
            if (max(max(abs_bits(v[0]), abs_bits(v[1])), max(abs_bits(v[2]), abs_bits(v[3]))) > F16_MAX_BITS) atomicOr(p.oflag, 2);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = h3::sat16(v[e]);
            uint2 hi, lo;
            hi.x = h3::cvt_pk(v[0], v[1]); hi.y = h3::cvt_pk(v[2], v[3]);
            lo.x = f16_sub_pk(v[0], v[1], hi.x); lo.y = f16_sub_pk(v[2], v[3], hi.y);
            unsigned char* dst = smem + Cf::Y_OFF + (G & 1) * Cf::YBUF + (c4 >> 3) * 2 * Cf::YPLANE + pt * 64 + (c4 & 7) * 8;
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + Cf::YPLANE) = lo;
        }
    };

    // ---- MFMA side: lane offset of the transpose reads inside a [point][32 ch] plane (see the header)
    const int li = lane & 15;
    const unsigned lane_off = (unsigned)((8 * (lane >> 5) + (li >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2);
    const int tap0 = wave * 7, ntap = min(7, 27 - tap0);
    f32x16 acc[7][2];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][n][r] = 0.f;
    const lds_v4s_ptr lbase = (lds_v4s_ptr)(smem);
    auto ldsp = [&](unsigned byte_off) { return (lds_v4s_ptr)((__attribute__((address_space(3))) unsigned char*)lbase + byte_off); };

    // prologue: groups 0 and 1 of x, group 0 of dy
    issue_x(0); issue_y(0);
    store_x(0); store_y(0);
    issue_x(1);
    store_x(1);
    __syncthreads();

    // r04: the 28 (k-block, tap) steps of an iteration are one static sequence with the fragments software-pipelined -- a wave is alone
    // on its SIMD (224 accumulator registers), so every LDS round trip that is not covered by its own MFMAs is idle matrix-pipe time: the
    // earlier form read a tap's fragments right in front of its six MFMAs (ISA: 4 ds_read, s_waitcnt, 6 MFMAs, 28 times per iteration:
    // 0.33 of the roof).  Now the x fragments of step s + 1 are requested before the MFMAs of step s into a ring of FOUR static register
    // sets (28 % 4 == 0: the slot of a step is the same in every iteration, no loop-carried copies), the dY fragments of k-block kb + 1
    // during tap 3 of k-block kb into the other of TWO sets.  A set is re-loaded two (x) / at least three (dY) steps after its last
    // reader: >= 6 younger MFMAs even on the wave whose seventh tap does not exist (common.h: the >= 4 rule).
    // Across the loop's back edge the wave drains its MFMAs before the barrier (mfma_drain: one accumulator read per chain).
    for (int it = 0; it < ngroups; ++it) {
        issue_x(it + 2);
        issue_y(it + 1);
        const long long v0 = (long long)it * RPI;
        const long long pl = plane0 + v0 / H;
        const int h0 = (int)(v0 % H), f = (int)(pl % F);
        const unsigned ybuf = Cf::Y_OFF + (it & 1) * Cf::YBUF;
        h3::f16x8 bf[2][2][2], af[4][2];
        auto load_b = [&](int kb, h3::f16x8 (&dst)[2][2]) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int pln = 0; pln < 2; ++pln) {
                    const unsigned o = ybuf + n * 2 * Cf::YPLANE + pln * Cf::YPLANE + kb * 16 * 64 + lane_off;
                    dst[n][pln] = tr_frag(ldsp(o), ldsp(o + 256));
                }
        };
        auto step_live = [&](int st) { const int t = st % 7; return t < 6 || t < ntap; };      // (every wave owns >= 6 taps)
        auto load_a = [&](int st, h3::f16x8 (&dst)[2]) {
            if (!step_live(st)) return;
            const int kb = st / 7, t = st % 7;
            const int r = kb / KPR, w0 = 16 * (kb % KPR);
            const int h = h0 + r, vrow = it * RPI + r;
            const int tap = tap0 + t, d = tap / 9, dh = (tap % 9) / 3, dw = tap % 3;
            const bool ok = (unsigned)(f + d - 1) < (unsigned)F && (unsigned)(h + dh - 1) < (unsigned)H;
            const unsigned row_off = ok ? (unsigned)(d * Cf::DFS + ((vrow + dh - 1) & (NS - 1)) * Cf::SLOT + (w0 + dw) * 64)
                                        : (unsigned)Cf::ZERO_OFF;
            const unsigned pl_off = ok ? (unsigned)Cf::PLANE : 0u;
            const unsigned o = row_off + lane_off;
            dst[0] = tr_frag(ldsp(o), ldsp(o + 256));
            dst[1] = tr_frag(ldsp(o + pl_off), ldsp(o + pl_off + 256));
        };
        load_b(0, bf[0]);
        load_a(0, af[0]);
#pragma unroll
        for (int st = 0; st < 28; ++st) {
            const int kb = st / 7, t = st % 7;
            if (st + 1 < 28) load_a(st + 1, af[(st + 1) & 3]);
            if (t == 3 && kb + 1 < 4) load_b(kb + 1, bf[(kb + 1) & 1]);
            if (step_live(st)) {
#pragma unroll
                for (int n = 0; n < 2; ++n) h3::mfma3(acc[t][n], af[st & 3], bf[kb & 1][n]);
                // the previous step's x set and (first tap of a k-block) the previous k-block's dY set stay LIVE behind this step's six
                // MFMAs: the register allocator may not hand them to the reads that follow (common.h: mfma_keep_a)
                if (st > 0) mfma_keep_a(acc[t][1], af[(st - 1) & 3][0], af[(st - 1) & 3][1]);
                if (st > 1) mfma_keep_a(acc[t][0], af[(st - 2) & 3][0], af[(st - 2) & 3][1]);      // (the step before may have been the missing seventh tap)
                if (t == 0 && kb > 0) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) mfma_keep_a(acc[t][n], bf[(kb - 1) & 1][n][0], bf[(kb - 1) & 1][n][1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                   // (one step at a time: left alone the scheduler hoists all 28 steps' reads)
        }
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int n = 0; n < 2; ++n) mfma_drain(acc[t][n]);
        store_x(it + 2);
        store_y(it + 1);
        __syncthreads();
    }

    // ---- partial sums: [slab][unit][tap][ci 32][co 64]
    float* dst = p.part + ((long long)slab * nunits + unit) * (27 * 32 * 64);
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        if (t >= ntap) continue;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dst[((tap0 + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 64 + n * 32 + l31] = acc[t][n][r];
    }
}

// dW [N][ctot][3][3][3] (channel slice at coff) = scale * sum over slabs (fixed order)
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nslab, int n_ci, int n_co,
                                                           int ctot, int coff, float scale, int accumulate) {
    const int nunits = n_ci * n_co;
    const long long per_unit = 27 * 32 * 64, total = (long long)nunits * per_unit;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += part[(long long)k * total + idx];
        const int unit = (int)(idx / per_unit);
        const int e = (int)(idx % per_unit);
        const int co = e & 63, ci = (e >> 6) & 31, tap = e >> 11;
        const int n = (unit % n_co) * 64 + co, c = (unit / n_co) * 32 + ci;
        float* dst = dw + ((long long)n * ctot + coff + c) * 27 + tap;
        *dst = accumulate ? *dst + s * scale : s * scale;
    }
}

static int wgrad3_slabs(int nunits, long long planes) {
    int nslab = std::max(8, (256 / std::max(nunits, 1)) / 8 * 8);
    while (nslab > 8 && nslab > planes) nslab -= 8;
    return nslab;
}

bool wgrad3_supported(int W, int H, int C, int N) {
    return (W == 64 || W == 32 || W == 16) && H % (64 / W) == 0 && C % 32 == 0 && N % 64 == 0;
}

size_t wgrad3_workspace_bytes(int C, int N, long long planes) {
    const int nunits = (C / 32) * (N / 64);
    return (size_t)wgrad3_slabs(nunits, planes) * nunits * 27 * 32 * 64 * sizeof(float) + 512;
}

template <int W>
static int launch_w(const Wgrad3Params& p, int nblocks, hipStream_t s) {
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)wgrad3_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, Wg3Cfg<W>::TOTAL));
        once = true;
    }
    hipLaunchKernelGGL(wgrad3_kernel<W>, dim3(nblocks), dim3(256), Wg3Cfg<W>::TOTAL, s, p);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// One sentinel word PER DEVICE (the word lives in that device's memory: a process that drives two GPUs must not share -- or fault on --
// one; keyed like DeviceOnce, by the current device).
static std::mutex g_grad_flag_mutex;
static int* g_grad_flag[64] = {};
int* f16x3_grad_overflow_flag() {
    const int d = DeviceOnce::dev();
    std::lock_guard<std::mutex> lock(g_grad_flag_mutex);
    if (!g_grad_flag[d]) {
        int* w = nullptr;
        if (hipMalloc(&w, sizeof(int)) != hipSuccess) return nullptr;
        (void)hipMemset(w, 0, sizeof(int));
        g_grad_flag[d] = w;
    }
    return g_grad_flag[d];
}
int f16x3_grad_overflow_status(int reset, hipStream_t s) {
    int* word;
    {
        std::lock_guard<std::mutex> lock(g_grad_flag_mutex);
        word = g_grad_flag[DeviceOnce::dev()];
    }
    if (!word) return DPC_OK;
    int v = 0;
    DPC_HIP(hipMemcpyAsync(&v, word, sizeof(int), hipMemcpyDeviceToHost, s));
    DPC_HIP(hipStreamSynchronize(s));
    if (!v) return DPC_OK;
    if (reset) DPC_HIP(hipMemsetAsync(word, 0, sizeof(int), s));
    return fail(DPC_ERR_STATE, std::string("f16x3 weight gradient: ") + ((v & 1) ? "an activation exceeded |x| = 4094" : "") +
                               ((v & 3) == 3 ? " and " : "") + ((v & 2) ? "an output gradient exceeded 65504 / f16_dy_scale" : "") +
                               " (or was not finite) since the last check: the operand was clamped, the step's gradients are not exact; "
                               "lower the loss scale or use wgrad_mode='f32'");
}

// Dynamic loss scaling without a per-step flag read or an extra collective: if an output gradient left the fp16 window since the last
// call (bit 1), the first element of the flat gradient buffer becomes +inf -- the all-reduce that follows spreads it to every rank, the
// gradient norm every rank computes anyway is then not finite, and all ranks skip the step together.  Clears bit 1.
__global__ void grad_poison_kernel(int* flag, float* g) {
    const int v = *flag;
    if (v & 2) {
        g[0] = INFINITY;
        *flag = v & ~2;
    }
}
int f16x3_grad_overflow_poison(float* g, hipStream_t s) {
    int* flag = f16x3_grad_overflow_flag();
    DPC_REQUIRE(flag && g, "train_range_poison: null argument");
    hipLaunchKernelGGL(grad_poison_kernel, dim3(1), dim3(1), 0, s, flag, g);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_wgrad3(const float* x, const float* dy, float* dw, int B, int F, int H, int W, int C, int N, int ctot, int coff, float x_scale,
                  float dy_scale, float out_scale, int accumulate, void* ws, hipStream_t s) {
    Wgrad3Params p{};
    p.x = x; p.dy = dy; p.Cx = C; p.N = N; p.F = F; p.H = H;
    p.n_ci = C / 32; p.n_co = N / 64;
    p.planes = (long long)B * F;
    const int nunits = p.n_ci * p.n_co;
    p.nslab = wgrad3_slabs(nunits, p.planes);
    p.part = reinterpret_cast<float*>(align_up((size_t)ws, 256));
    p.x_scale = x_scale; p.dy_scale = dy_scale;
    p.oflag = f16x3_grad_overflow_flag();
    DPC_REQUIRE(p.oflag, "wgrad3: cannot allocate the gradient-range sentinel word");
    const int nblocks = nunits * p.nslab;                                  // nslab % 8 == 0: 8 XCD lanes of nunits * nslab / 8 blocks
    {
        ProfScope prof(PROF_WGRAD3, 2.0 * (double)p.planes * H * W * 27 * C * N, 0, s);
        int rc = W == 64 ? launch_w<64>(p, nblocks, s) : (W == 32 ? launch_w<32>(p, nblocks, s) : launch_w<16>(p, nblocks, s));
        if (rc) return rc;
    }
    ProfScope prof(PROF_TRAIN_MISC, 0, (double)p.nslab * nunits * 27 * 32 * 64 * 4, s);
    const long long total = (long long)nunits * 27 * 32 * 64;
    hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, p.part, dw, p.nslab,
                       p.n_ci, p.n_co, ctot, coff, out_scale / (x_scale * dy_scale), accumulate);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
