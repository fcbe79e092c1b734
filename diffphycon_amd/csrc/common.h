// Internal helpers shared by the libdpc translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/dpc.h"

namespace dpc {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define DPC_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::dpc::fail(DPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define DPC_LAUNCH_CHECK()                                                                               \
    do {                                                                                                 \
        hipError_t _e = hipGetLastError();                                                               \
        if (_e != hipSuccess)                                                                            \
            return ::dpc::fail(DPC_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(_e) +     \
                                                " at " + __FILE__ + ":" + std::to_string(__LINE__));     \
    } while (0)

#define DPC_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) return ::dpc::fail(DPC_ERR_ARG, (msg));     \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// One-time, PER-DEVICE set-up guard (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: a
// process that touches a second GPU must set it there too).  Drop-in for `static bool once`: `if (!once) { ...; once = true; }`.
// Racing host threads may both run the block -- it is idempotent -- but neither can skip it on a device where it has not run.
struct DeviceOnce {
    std::atomic<unsigned long long> mask{0};
    static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
    bool operator!() const { return !((mask.load(std::memory_order_acquire) >> dev()) & 1ull); }
    DeviceOnce& operator=(bool v) { if (v) mask.fetch_or(1ull << dev(), std::memory_order_release); return *this; }
};

// Remainder plane of the 2-way fp16 operand split: the packed pair (f16(x0 - h.lo), f16(x1 - h.hi)) for a packed f16 pair h, as ONE
// v_fma_mix per element (f32 * 1.0 - f16 -> f16).  x - h is exact in fp32, so this rounds exactly as convert -> subtract -> convert
// does, in a third of the VALU instructions.
// Hardware saturation of fp16 conversions (r04; f16x3.h has the long form of this note): with MODE.FP16_OVFL set an fp16 RESULT that
// overflows becomes +-65504 instead of inf, so the v_med3 in front of every operand conversion of the f16x3 loaders is not needed.
// A kernel calls fp16_ovfl_enable() once at its top and converts sat16x(x) instead of a clamped x.  -DDPC_FP16_OVFL=0: software clamp.
#ifndef DPC_FP16_OVFL
#define DPC_FP16_OVFL 1
#endif
__device__ __forceinline__ void fp16_ovfl_enable() {
    if (DPC_FP16_OVFL) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);       // hwreg(HW_REG_MODE, offset 23, width 1) = 1
}
__device__ __forceinline__ float sat16x(float x) {
    return DPC_FP16_OVFL ? x : __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f);
}

__device__ __forceinline__ unsigned f16_sub_pk(float x0, float x1, unsigned h) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(r) : "v"(x0), "v"(x1), "v"(h));
    return r;
}

// ---------------------------------------------------------------- arithmetic modes (api.hip)
// How fp32 products are evaluated, per op family: 0 = native fp32 MFMA, 1 = bf16x6 (exact 3-way bf16 split, 6 MFMAs),
// 2 = f16x3 (2-way fp16 split, 3 MFMAs; default).  The process-wide setting (initialised from DPC_{CONV,IGEMM,ATTN,STEM}_MODE,
// changed by dpc_set_mode) is CAPTURED by a U-Net handle when it is created; every later call on that handle runs under a
// ModeScope with the captured values, so two handles with different modes can live side by side and an inherited
// environment cannot change a handle after the fact.  dpc_get_mode / dpc_unet*_modes report what is active.
struct Modes { int conv, igemm, attn, stem; };
Modes modes_global();
const Modes& modes_current();              // the innermost ModeScope of this thread, else the process-wide setting
struct ModeScope {
    explicit ModeScope(const Modes& m);
    ~ModeScope();
    const Modes* prev_;
    Modes cur_;
};
// Opt-in activation range check of the f16x3 mode (dpc_unet*_set_range_check).  Range contract per kernel family: every f16x3
// kernel represents |x| <= 4094 with 22 significant bits; beyond it the direct convolutions / implicit GEMMs / stem (conv3f3.hip /
// igemm6.hip / stem7x6.hip: pre-scale SA = 16) CLAMP at 65504 / 2^4 = 4094, while the Winograd convolutions (conv3w4.hip, F(4,3):
// pre-scale 2 / log2 e inside the operand split; conv3w.hip, F(2,3): 2^3 / 4 log2 e) stay exact up to 4678 / 6486 (F(2,3): 4094 / 5676; plain / fused GroupNorm input) and then produce inf -> NaN.  With the check on, every f16x3 conv / implicit
// GEMM / stem launch of a forward is preceded by a streaming pass over its input (after the fused GroupNorm+SiLU where the
// halo staging applies one) that records the first op whose input leaves the range; the forward then FAILS instead of
// returning a result computed from clamped activations.  Off by default (costs one extra read of every conv input).
struct RangeCheck {
    bool on = false;
    int* flag = nullptr;                    // device word: 1-based index of the first offending op, INT_MAX = clean
    std::vector<std::string> names;         // op names in launch order (index = position)
    std::string cur;                        // name of the op about to be launched (set by the orchestrators)
};
RangeCheck* range_check_current();          // innermost RangeCheckScope of this thread or null
struct RangeCheckScope {
    explicit RangeCheckScope(RangeCheck* rc);
    ~RangeCheckScope();
    RangeCheck* prev_;
};
// input [rows][ctot] fp32 (channels [coff, coff + C) checked), optional GroupNorm coefficient table in_coef [B][C/4][5][4] with
// rows_per_sample rows per sample; atomicMin(flag, id) when any |value| > limit or is not finite
int launch_range_check(const float* x, long long rows, int C, const float* in_coef, long long rows_per_sample, float limit,
                       int* flag, int id, hipStream_t s);
int range_check_note(const float* a0, long long rows0, int C0, const float* a1, long long rows1, int C1, const float* in_coef,
                     long long rows_per_sample, hipStream_t s);
// ALWAYS-ON activation-range sentinel of the f16x3 mode (cheap; complements the opt-in streaming RangeCheck above).  The tensors a
// later f16x3 kernel splits WITHOUT a normalisation in front are the residual-stream tensors; each is checked where it is produced,
// inside kernels that are not matrix-bound: the GroupNorm apply (+ residual) pass, the implicit-GEMM vector epilogue (res_conv with
// the fused GroupNorm residual, attention output projections, down / up-sampling convolutions, split-K reduce) and the fused temporal
// attention's store.  A value with |v| > 4094 (the documented contract of every f16x3 consumer: 65504 / 2^4) or a non-finite one ORs
// bit 0 into this device word; dpc_unet3d_range_status reads it once per sample() -- no per-forward sync, no extra pass.
constexpr unsigned F16X3_ACT_LIMIT_BITS = 0x457FE000u;      // 4094.0f
int* overflow_flag_current();               // device word of the innermost OverflowScope of this thread, or null
struct OverflowScope {
    explicit OverflowScope(int* flag);
    ~OverflowScope();
    int* prev_;
};
// A/B and attribution switches of the kernel selection (DPC_CONV3W=0, DPC_IGEMM_SPLITK=0, DPC_UNFUSED_ATTN=1, ...) are
// DEVELOPMENT aids: they are read only when the process also sets DPC_DEBUG=1 -- a stray variable in a production environment
// cannot change which kernels run (the arithmetic modes DPC_*_MODE are the documented, per-handle-captured setting and are not
// gated).  Switches that make a kernel skip work (DPC_CONV_DBG) exist only in builds with -DDPC_ENABLE_CONV_DBG.
int debug_switch(const char* name, int dflt);
// attribution bits added in r05 (Conv3hParams::dbg 8 / 64 / 128 / 256) exist in -DDPC_ENABLE_CONV_DBG builds only: the product kernels
// do not even test them
#ifdef DPC_ENABLE_CONV_DBG
constexpr bool CONV_DBG_BUILD = true;
#else
constexpr bool CONV_DBG_BUILD = false;
#endif
int cu_budget(int ncu);           // api.hip: min(ncu, the value of dpc_set_cu_budget rounded down to a multiple of 8)
// library-internal scratch buffer of at least `bytes` for (slot, current device, stream) -- api.hip; slots:
enum { SCRATCH_SMALL_ACT = 0, SCRATCH_SPLITK = 1 };
int stream_scratch(int slot, hipStream_t s, size_t bytes, float** out);
// A U-Net forward lends a region of its caller-provided workspace for the duration of the traversal: requests that fit are served
// from it (per handle / per call, nothing shared between handles or streams, and no allocation under a HIP-graph capture); the
// consumers of one region are launches of ONE stream whose uses do not overlap (the activated time embedding is consumed by the
// launch right behind it, split-K partials by their reduce kernel).  Larger requests and the operator-level C-ABI entry points fall
// back to the (slot, device, stream) table.
struct ScratchScope {
    ScratchScope(void* base, size_t bytes);
    ~ScratchScope();
    void* prev_base_;
    size_t prev_bytes_;
};
const char* mode_name(int mode);
std::string modes_string(const Modes& m);
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- opt-in event timing (profile.hip)
enum ProfClass {
    PROF_IGEMM64 = 0, PROF_IGEMM128, PROF_STEM, PROF_GN, PROF_LN, PROF_ATTN, PROF_LINATTN, PROF_UPDATE, PROF_SMALL,
    PROF_BURGERS, PROF_PHILOX, PROF_SMOKE_EVAL, PROF_CONV3H64, PROF_CONV3H128, PROF_TATTN_FUSED, PROF_LATTN_FUSED, PROF_CONV3X6_64,
    PROF_CONV3X6_128, PROF_WGRAD, PROF_ATTN_BWD, PROF_TRAIN_MISC, PROF_WGRAD3, PROF_NCLASS
};
struct ProfScope {
    ProfScope(int cls, double flops, double bytes, hipStream_t s);
    ~ProfScope();
    long long idx_;
    hipStream_t s_;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// |bits| compare as unsigned: Inf / NaN patterns are larger than every finite one, so they trip the sentinel too
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ void overflow_note4(int* flag, const f32x4& v) {
    const unsigned m = max(max(abs_bits(v[0]), abs_bits(v[1])), max(abs_bits(v[2]), abs_bits(v[3])));
    if (flag && m > F16X3_ACT_LIMIT_BITS) atomicOr(flag, 1);
}

// MFMA operand keep-alive (third hazard).  The matrix pipe reads an MFMA's B operand while the instruction executes; an
// LDS read that is issued right behind the MFMA into the SAME registers can overwrite them first when the MFMA is held up (a foreign
// wave's MFMAs on the SIMD).  hipcc's register allocator creates exactly that pattern on its own: a B fragment that is dead after its
// last MFMA hands its registers to the next ds_read.  mfma_keep(acc, frag...) emits NO instruction: it is an empty asm that reads the
// fragment(s) and is tied into the accumulator's dependence chain ("+v"), so it sits between the MFMAs that produced `acc` and the next
// MFMA on it, and the fragments' registers stay allocated up to that point -- no load issued before it can be given them.  Called once
// per accumulator of a k-step AFTER the step that follows the fragments' last reader, it keeps a spent B operand's registers for all
// MFMAs of that step (>= 4).  Accumulators must live in VGPRs (every kernel that uses this stays within 256 registers).
// tools/mfma_war_audit.py measures the result in the ISA; tests/test_mfma_war_audit.py pins >= 4 MFMAs for every MFMA kernel of the
// default path.
template <class A, class T>
__device__ __forceinline__ void mfma_keep(A& acc, const T& f0) { asm volatile("" : "+v"(acc) : "v"(f0)); }
template <class A, class T>
__device__ __forceinline__ void mfma_keep(A& acc, const T& f0, const T& f1) { asm volatile("" : "+v"(acc) : "v"(f0), "v"(f1)); }
template <class A, class T>
__device__ __forceinline__ void mfma_keep(A& acc, const T& f0, const T& f1, const T& f2) {
    asm volatile("" : "+v"(acc) : "v"(f0), "v"(f1), "v"(f2));
}
template <class A, class T>
__device__ __forceinline__ void mfma_keep(A& acc, const T& f0, const T& f1, const T& f2, const T& f3) {
    asm volatile("" : "+v"(acc) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
}
template <class A, class T>
__device__ __forceinline__ void mfma_keep(A& acc, const T& f0, const T& f1, const T& f2, const T& f3, const T& f4, const T& f5) {
    asm volatile("" : "+v"(acc) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5));
}
// the same for accumulators that live in AccVGPRs (kernels whose accumulators fill the 256 AGPRs of a 512-register wave)
template <class A, class T>
__device__ __forceinline__ void mfma_keep_a(A& acc, const T& f0, const T& f1) { asm volatile("" : "+a"(acc) : "v"(f0), "v"(f1)); }
template <class A, class T>
__device__ __forceinline__ void mfma_keep_a(A& acc, const T& f0) { asm volatile("" : "+a"(acc) : "v"(f0)); }
// Where a loop's back edge (or a workgroup barrier) separates the last MFMAs from the next LDS reads, the wave waits for its MFMAs
// instead: one VALU read of an element of every accumulator chain completes only when that chain's last MFMA has written its result,
// i.e. has long finished reading its operands (the reads are ordered before the following barrier / loads by being volatile).
template <class A>
__device__ __forceinline__ void mfma_drain(const A& acc) {
    float t;
    asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(acc[0]) : "memory");      // ("memory": no later load is hoisted above it)
}
// MFMAs may not be scheduled across this point (everything else may: sched_barrier mask VALU 2 | SALU 4 | VMEM 0x70 | DS 0x380 |
// transcendentals 0x400): where independent accumulator chains follow each other, it makes "the MFMAs of the next group come after the
// MFMAs of this one" a property of the schedule instead of a habit of the scheduler
__device__ __forceinline__ void mfma_order_point() { __builtin_amdgcn_sched_barrier(0x7f6); }
// a set of NT x 2 plane fragments against every accumulator of an MT x NT wave tile
template <int MT, int NT, class A, class T>
__device__ __forceinline__ void mfma_keep_set(A (&acc)[MT][NT], const T (&w)[NT][2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (NT == 1) mfma_keep(acc[mt][nt], w[0][0], w[0][1]);
            else if constexpr (NT == 2) mfma_keep(acc[mt][nt], w[0][0], w[0][1], w[1][0], w[1][1]);
            else mfma_keep(acc[mt][nt], w[0][0], w[0][1], w[1][0], w[1][1], w[2][0], w[2][1]);
        }
}

// ---------------------------------------------------------------- implicit GEMM (igemm.hip)
// out[m][n] = bias[n] + resid[m][n] + sum_{tap,c} A(m,tap,c) * W(tap,c,n)
// A rows are output points (b,f,ho,wo) of a channels-last activation; see DESIGN.md.
struct IgemmParams {
    float act_scale;        // f16x3 kernels: power-of-two scale applied to the activation operand before the fp16 split (0: 2^4),
                            // undone in the epilogue; lets a caller place a tensor of any magnitude inside the fp16 window
    float descale;          // (filled by the launcher: 1 / (act_scale * 2^12))
    int cs0;                // elements between consecutive pixels of source 0 (filled by the launcher: C0 unless a0_stride is set)
    int a0_stride;          // 0, or a pixel stride < C0: the C0 "channels" of a pixel are a contiguous window over the following
                            // pixels (row-window convolution of the 7x7 stems, surr.hip)
    const float* a0;        // source 0, channels-last [B*F, Hi, Wi, C0]
    const float* a1;        // source 1 (virtual concat along channels) or null
    int C0, C1;
    const float* wp;        // packed weights [ntaps][kchunks][Npad][32]
    const float* bias;      // [N] or null
    const float* resid;     // [M][N] or null (added after bias)
    float* out;
    const float* ln_stats;  // [rows][2] (mean, inv_std) or null: channel LayerNorm fused on the A operand
    const float* ln_gamma;  // [K]
    int BF, F;              // BF = B*F frames in total
    int Hi, Wi, Ho, Wo;
    int sh, sw;             // spatial stride of the input sampling
    int ntaps;
    int N, Npad, kchunks;
    int out_mode;           // 0: [M][N]; 1: per-frame channels-first [BF][N][Ho*Wo]; 2: parity scatter into [BF][2Ho][2Wo][N]
    int par_a, par_b;
    signed char tdf[32], tdh[32], tdw[32];
    long long M;
    // split-K (igemm3 only, set by its launcher): blockIdx.y = slice of the (tap, channel-chunk) iterations; raw accumulators go
    // to part[slice][M][N] and a second kernel adds the slices in fixed order, rescales and applies bias / residual
    int ksplit;
    float* part;
    // fused GroupNorm-apply residual (igemm3, out_mode 0): v += silu((gn_raw[m][n] - mu) * ga + be) with the per-(sample,
    // channel) coefficients of launch_gn_finalize_fused ([B][N/4][5][4]); gn_rows = rows per sample (multiple of 128).
    // This is ResnetBlock's  block2(h) + res_conv(x)  without materialising block2's activated output.
    const float* gn_raw;
    const float* gn_coef;
    long long gn_rows;
    int dbg;                // debugging only (env DPC_IGEMM_DBG): 1 = scalar epilogue
    int* oflag;             // f16x3 activation-range sentinel (filled by the launcher from overflow_flag_current(); out_mode 0 / 2 outputs)
};
// f16x3 weight packers raise this device flag when a weight leaves the fp16 range after its 2^12 pre-scale (|w| > 15.99);
// dpc_unet*_finalize reads it (a host sync, at load time only) and fails loudly instead of computing with a clamped weight.
int* f16x3_weight_overflow_flag();          // device pointer
int f16x3_weight_overflow_check(const char* who);   // DPC_OK, or DPC_ERR_STATE (and resets the flag)
// f16x3 weight-gradient kernel (wgrad3.hip): device word raised when an operand leaves the fp16 window after its pre-scale (the value
// is clamped); dpc_train_range_status reads it (one host sync; the Trainer asks when it logs the loss, never per step)
int* f16x3_grad_overflow_flag();
int f16x3_grad_overflow_status(int reset, hipStream_t s);
int f16x3_grad_overflow_poison(float* g, hipStream_t s);
int igemm_npad(int N);
int igemm_kchunks(int K);
int launch_igemm(const IgemmParams& p, hipStream_t s);
// bf16x6 variant of the same op family (igemm6.hip): weights pre-split into [it][Npad][3 planes][32] bf16
int igemm_mode_default();          // 0: native fp32 MFMA (env DPC_IGEMM_MODE=f32), 1: bf16x6 (default)
size_t igemm6_packed_bytes(int Npad, int K, int ntaps);
int launch_igemm6(const IgemmParams& p, const void* wp6, hipStream_t s);
// 256 x 128 tile with both operands staged through LDS (igemm_wide.hip): long reductions into >= 128 columns; p as prepared by
// launch_igemm6 (nsl > 1: split-K slices, raw partials into p.part)
// 64-row x all-N panels for 1-tap ops with K <= 256 (igemm_panel.hip); p as prepared by launch_igemm6
bool igemm3p_supported(const IgemmParams& p);
// pixel tiles for the stride-1 2 x 2-tap classes of ConvTranspose (igemm_tile.hip)
bool igemm3t_supported(const IgemmParams& p);
int launch_igemm3t(const IgemmParams& p, const void* wp6, hipStream_t s);
int launch_igemm3p(const IgemmParams& p, const void* wp6, hipStream_t s);
// pixel tiles with a K-block loop for 3 x 3 convolutions on small images with long reductions (igemm_img.hip); nsl as igemm3w
bool igemm3i_supported(const IgemmParams& p);
int launch_igemm3i(const IgemmParams& p, const void* wp6, int nsl, hipStream_t s);
bool igemm3w_supported(const IgemmParams& p);
int igemm3w_slices(const IgemmParams& p);      // split-K slices by shape (N, reduction length), never by the batch
int launch_igemm3w(const IgemmParams& p, const void* wp6, int nsl, hipStream_t s);
int launch_pack_weights_g6(const float* w, void* wp6, int N, int Npad, int K, int ntaps, long long stride_n,
                           long long stride_c, const int* tap_off_host, hipStream_t s);
// generic weight re-pack: wp[tap][kc][n][kk] = w[n*stride_n + (kc*32+kk)*stride_c + tap_off[tap]]
int launch_pack_weights(const float* w, float* wp, int N, int Npad, int K, int ntaps, long long stride_n,
                        long long stride_c, const int* tap_off_host, hipStream_t s, int bk = 32);

// Conv3d 3x3x3 stride 1 pad 1 with the LDS-staged halo tile (conv3h.hip); weights packed with bk = 16
struct Conv3hParams {
    float act_scale;        // as IgemmParams::act_scale (conv3f3 / conv3f3b; 0: 2^4)
    const float* a0;        // channels-last [B,F,H,W,C0]
    const float* a1;        // virtual concat source [B,F,H,W,C1] or null
    int C0, C1;
    const float* wp;        // [27][kchunks][Npad][16]
    const void* wpw;        // f16x3 only: Winograd-transformed pack -- F(4,3), default: [6][9][kchunks][Npad/32][2][2][32][8] fp16 (conv3w4.hip);
                            // F(2,3): [4][9][kchunks][Npad][2][16] (conv3w.hip); launch_pack_weights_w3 writes the form launch_conv3w runs -- or null
    const float* bias;
    float* out;             // channels-last [B,F,H,W,N]
    int B, F, H, W;
    int N, Npad, kchunks;   // kchunks = ceil((C0+C1)/16)
    int dbg;                // perf attribution only (env DPC_CONV_DBG): 1 skip output stores, 2 skip halo loads, 4 skip weight loads
    int kd;                 // 0 / 3: 3x3x3 kernel; 1: (1,3,3) kernel over independent frames (conv3f3 big-tile kernel only)
    int total_wg;           // conv3f3b persistent form: number of tile workgroups walked by a one-per-CU grid (0 = one tile per workgroup)
    // GroupNorm fusion (conv3x6 only):
    float* gn_part;         // out: per-(sample, tile, frame-pair) channel sums of the conv output [B][tiles][2][N][2] (sum, sum sq)
    const float* in_coef;   // in: GroupNorm+scale/shift coefficients of the INPUT [B][K/4][5][4] (mu, rstd*gamma, beta, scale+1,
                            //     shift), then the folded table [B][K/4][2][4] = (A, B) log2(e) (launch_gn_finalize_fused; conv3w),
                            //     shift): the halo load applies GN -> (scale, shift) -> SiLU on the fly (requires C1 == 0)
};
// tiles per sample of the conv3x6 output tiling (4 x 4 x 8)
long long conv3x6_tiles_per_sample(int F, int H, int W);
// stats [B][groups][2] (mean, rstd) and, when coef != null, the per-channel coefficient table consumed by Conv3hParams::in_coef
int launch_gn_finalize_fused(const float* part, int B, long long tiles, int C, int groups, long long rows_per_sample,
                             const float* gamma, const float* beta, const float* scale_shift, float* stats, float* coef,
                             hipStream_t s, long long entries = 0 /* partial-sum entries per sample; 0: tiles * 2 */);
// GroupNorm apply (+ scale/shift, SiLU, residual) from finished statistics
int launch_gn_apply(const float* x, float* out, const float* resid, const float* stats, const float* gamma,
                    const float* beta, const float* scale_shift, int B, long long rows_per_sample, int C, int groups,
                    hipStream_t s);
int launch_conv3h(const Conv3hParams& p, hipStream_t s);
// same op on the bf16 matrix cores with an exact 3-way bf16 split of both operands (conv3x6.hip); p.wp then points to
// the pre-split weights [27][kchunks][Npad][3][16] bf16 made by launch_pack_weights_x6
int launch_conv3x6(const Conv3hParams& p, hipStream_t s);
int launch_pack_weights_x6(const float* w, void* wp, int N, int Npad, int K, hipStream_t s);
// same op with a 2-way fp16 split (22-bit operands, 3 MFMAs per product; conv3f3.hip); p.wp = [27][kchunks][Npad][2][16] fp16
int launch_conv3f3(const Conv3hParams& p, hipStream_t s);
int launch_pack_weights_f3(const float* w, void* wp, int N, int Npad, int K, hipStream_t s, int ntaps = 27);   // 9: (1,3,3) kernel
long long conv3f3_gn_entries(int F, int H, int W, int N, int Npad);
// loader-wave / persistent form of the big-tile kernel (conv3f3c.hip); same GroupNorm partial-sum layout
bool conv3f3c_supported(const Conv3hParams& p);
bool conv3f3c_flat_gn_ok(int N, int Npad, int H, int W);      // the (1,3,3) form with per-image GroupNorm hooks takes this shape
long long conv3f3c_flat_gn_entries(int H, int W);             // its partial-sum entries per image
int launch_conv3f3c(const Conv3hParams& p, hipStream_t s);     // GroupNorm partial-sum entries per (sample, channel)
// Winograd F(2,3)-over-frames form of the f16x3 3x3x3 convolution (conv3w.hip): 2/3 of the matrix products of the direct kernels.
// Taken by launch_conv3f3 when Conv3hParams::wpw is set and the shape qualifies (shape-only rule).
bool conv3w_shape_ok(int F, int H, int W, int N, int Npad);
bool conv3w_supported(const Conv3hParams& p);
long long conv3w_gn_entries(int F, int H, int W);
int launch_conv3w(const Conv3hParams& p, hipStream_t s);
size_t conv3w_packed_bytes(int Npad, int K);
// Winograd F(4,3)-over-frames form (conv3w4.hip, r06): 3/4 of conv3w's matrix products.  Default; launch_conv3w / launch_pack_weights_w3 /
// conv3w_packed_bytes forward to it (same shape rule, same GroupNorm partial-sum entry count), DPC_DEBUG=1 DPC_CONV3W_F43=0 keeps F(2,3).
bool conv3w_f43_enabled();
int launch_conv3w4(const Conv3hParams& p, hipStream_t s);
size_t conv3w4_packed_bytes(int Npad, int K);
int launch_pack_weights_w4(const float* w, void* wp, int N, int Npad, int K, hipStream_t s);
int launch_pack_weights_w3(const float* w, void* wp, int N, int Npad, int K, hipStream_t s);
// 0: native fp32 MFMA (conv3h), 1: bf16x6 split (conv3x6), 2: f16x3 split (conv3f3); env DPC_CONV_MODE=f32|x6|f16x3, default f16x3
int conv_mode_default();

// Fused Residual(PreNorm(temporal Attention)) (tattn_fused.hip); weights in the reference layout
struct TattnParams {
    const float* x;         // channels-last [B,F,HW,C]
    float* out;             // may alias x
    const float* gamma;     // [C]
    const float* wqkv;      // to_qkv.weight [384][C]
    const float* wout;      // to_out.weight [C][128]
    const float* rot_cos;   // [F][32]
    const float* rot_sin;
    const float* bias;      // [4][F][F]
    const float* bias32;    // the same table zero-padded to [4][32][32] (bf16x6 kernel: 16-byte row loads); F <= 32 only
    const float* brel;      // Toeplitz form [4][128]: bias[h][i][j] = brel[h][j - i + 63] (tattn3, 32 < F <= 64); null if the
                            // table is not a function of j - i
    long long npix;         // B*HW sequences
    long long HW;
    int F;
    int* oflag;             // tattn3 only: f16x3 activation-range sentinel for the block output (filled by its launcher)
};
// Fused Residual(PreNorm(SpatialLinearAttention)) (lattn_fused.hip); weights in the reference layout
struct LattnParams {
    const float* x;         // channels-last [images, N, C]
    float* out;             // may alias x
    const float* gamma;     // [C]
    const float* wqkv;      // to_qkv.weight [384][C] (Conv2d 1x1)
    const float* wout;      // to_out.weight [C][128]
    const float* bout;      // to_out.bias [C]
    float* ctx;             // workspace [images][4][32][32]
    long long images;
    int N;
    const float* gamma_out = nullptr;   // lattn3 only: gain of a channel LayerNorm behind to_out (2-D U-Net), nullptr = none
};
bool lattn_fused_supported(int C, int heads);
size_t lattn_fused_workspace_bytes(long long images);
int launch_lattn_fused(const LattnParams& p, int C, hipStream_t s);
bool tattn_fused_supported(int C, int F, int heads);
int launch_tattn_fused(const TattnParams& p, int C, hipStream_t s);
// bf16x6 versions (tattn6.hip / lattn6.hip): weights pre-split at load time into per-head LDS images
size_t attn6_qkv_bytes(int C);
size_t attn6_out_bytes(int C);
int launch_pack_attn6(const float* w, unsigned char* dst, int C, bool is_out, hipStream_t s);
int launch_tattn6(const TattnParams& p, const unsigned char* wq6, const unsigned char* wo6, int C, hipStream_t s);
// persistent weight-stationary f16x3 version for C = 64 (tattn3.hip)
bool tattn3_supported(int C, int F, int heads);
size_t tattn3_qkv_bytes(int C);
size_t tattn3_out_bytes(int C);
size_t tattn3_workspace_bytes(int C, long long rows);      // C = 128: partial to_out sums of the first head pair
int launch_pack_tattn3(const float* w, unsigned char* dst, int C, bool is_out, hipStream_t s);
int launch_tattn3(const TattnParams& p, const unsigned char* wq3, const unsigned char* wo3, int C, void* workspace, hipStream_t s);
// single-launch weight-stationary f16x3 linear attention for C = 64 (lattn3.hip); same weight images as tattn3; p.ctx unused
bool lattn3_supported(int C, int heads);
int launch_lattn3(const LattnParams& p, const unsigned char* wq3, const unsigned char* wo3, int C, hipStream_t s);

// "gather" variant for the 7x7x7 stem on the reference-layout input [BF, C, H, W] (K = taps*C flattened)
struct StemParams {
    const float* x;         // [BF][C][H][W]
    const float* wp;        // [kchunks][Npad][32]
    const int* ktab;        // [kchunks*32] packed (df+64)<<24 | (dh+64)<<16 | (dw+64)<<8 | c ; -1 = padding
    const float* bias;
    float* out;             // channels-last [M][N]
    int BF, F, C, H, W;
    int Ctot, c_off;        // x is a channel slice [c_off, c_off+C) of a [BF][Ctot][H][W] tensor
    int N, Npad, kchunks;
    long long M;
};
int launch_stem(const StemParams& p, hipStream_t s);
// bf16x6 LDS-halo variant for the 7x7x7 kernel with <= 8 input channels (stem7x6.hip)
bool stem7x6_supported(int C, int k);
size_t stem7x6_packed_bytes(int Npad);
int launch_pack_stem7x6(const float* w, void* wp6, int N, int Npad, int C, hipStream_t s);
int launch_stem7x6(const StemParams& p, const void* wp6, hipStream_t s);
int launch_pack_stem(const float* w, float* wp, int* ktab, int N, int Npad, int C, int k, hipStream_t s, int kd = 0);

// ---------------------------------------------------------------- norms (norm.hip)
int launch_ln_stats(const float* x, float* stats, long long rows, int C, hipStream_t s);
// out[r][c] = resid[r][c] + (x[r][c] - mean_r) * rstd_r * gamma[c]   (channel LayerNorm + residual; out may alias resid)
int launch_ln_apply(const float* x, const float* stats, const float* gamma, const float* resid, float* out,
                    long long rows, int C, hipStream_t s);
size_t gn_workspace_bytes(int B, int C);
// statistics only: stats [B][groups][2] = (mean, rstd) of x [B][R][C]
int launch_gn_stats(const float* x, float* stats, int B, long long R, int C, int groups, void* ws, hipStream_t s);
// backward of out = SiLU(GN(x) * (scale + 1) + shift): dx [B][R][C] and, when dss != null, d(scale | shift) [B][2C]
size_t gn_bwd_workspace_bytes(int B, int C);
int launch_gn_silu_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                       const float* scale_shift, float* dx, float* dss, int B, long long R, int C, int groups, void* ws,
                       hipStream_t s, float* dgamma = nullptr, float* dbeta = nullptr);
// d gamma / d beta [C] of the same GroupNorm from the partial sums launch_gn_silu_bwd's first pass leaves in ws (train.hip)
int launch_gn_param_grad(const void* ws, const float* scale_shift, float* dgamma, float* dbeta, int B, int C, int nchunk, hipStream_t s);
// 3x3x3 weight gradient on the fp16 matrix cores (wgrad3.hip): f16x3 operands, x pre-scaled by x_scale, dy by dy_scale (powers of two)
bool wgrad3_supported(int W, int H, int C, int N);
size_t wgrad3_workspace_bytes(int C, int N, long long planes);
int launch_wgrad3(const float* x, const float* dy, float* dw, int B, int F, int H, int W, int C, int N, int ctot, int coff, float x_scale,
                  float dy_scale, float out_scale, int accumulate, void* ws, hipStream_t s);
// backward of the channel LayerNorm y = (x - mean) * rstd * g: dx (= or +=) per row
int launch_ln_bwd(const float* x, const float* stats, const float* g, const float* dy, float* dx, long long rows, int C, int accum,
                  hipStream_t s);
// out = SiLU(GN(x)*(scale+1)+shift) (+ resid); out may alias x or resid (same-element in-place)
int launch_groupnorm_silu(const float* x, float* out, const float* resid, const float* gamma, const float* beta,
                          const float* scale_shift, int B, long long rows_per_sample, int C, int groups, void* ws,
                          hipStream_t s);

// ---------------------------------------------------------------- attention (attn.hip)
struct AttnParams {
    const float* qkv;      // [rows][3*heads*32]
    float* out;            // [rows][heads*32]
    int heads, L;
    long long n_seq, seq_inner, seq_outer_stride, seq_inner_stride, token_stride;   // in rows
    const float* rot_cos;  // [L][32] or null
    const float* rot_sin;
    const float* bias;     // [heads][L][L] or null
};
int launch_attention(const AttnParams& p, hipStream_t s);
size_t linattn_workspace_bytes(long long images, int heads);
// ws: [images * heads][ctx_stride] floats; save != 0 also stores kmax / Z of the k softmax behind each context (+1024, +1056)
int launch_linear_attention(const float* qkv, float* out, int heads, long long images, int N, void* ws,
                            hipStream_t s, int ctx_stride = 1024, int save = 0);

// ---------------------------------------------------------------- small ops (small.hip)
// out[b][n] = out_act( bias[n] + sum_k in_act(in[b][k]) * W[n][k] ),  act: 0 none, 1 SiLU, 2 GELU(erf)
int launch_small_linear(const float* in, const float* W, const float* bias, float* out, int B, int K, int N,
                        int in_act, int out_act, hipStream_t s);
// the same for up to 32 (W, bias, out, N) sets sharing the input, one launch (small.hip)
struct SmallLinearBatch { const float* W[32]; const float* bias[32]; float* out[32]; int N[32]; int count; };
int launch_small_linear_multi(const float* in, const SmallLinearBatch& d, int B, int K, int in_act, int out_act, hipStream_t s);
int launch_sinusoidal(const int64_t* t, const float* freqs, float* out, int B, int half, hipStream_t s);
int launch_cl_to_cf(const float* x_cl, float* x_cf, int BF, int C, long long HW, int F, hipStream_t s);  // debug taps
// 1x1x1 convolution K = 64 -> N <= 8 channels, rows [M][K] -> per-frame channels-first out[M / HW][N][HW] (small.hip)
bool conv1x1_rows_supported(int K, int N);
int launch_conv1x1_rows(const float* x, const float* W, const float* bias, float* out, long long M, long long HW, int K, int N,
                        hipStream_t s);
int launch_cf_to_cl(const float* x_cf, float* x_cl, int BF, int C, long long HW, hipStream_t s);
// nearest-neighbour x2 up-sampling of a channels-last image batch [BF][H][W][C] -> [BF][2H][2W][C]
int launch_upsample2x_cl(const float* x, float* y, int BF, int H, int W, int C, hipStream_t s);

// ---------------------------------------------------------------- sampler update (update.hip)
int launch_ddpm_update_smoke(const float* x, const float* eps_j, const float* eps_w, const float* z,
                             const float* init, const float* rescaler, float* x_next, float* x0_out,
                             const dpc_step_coef& c, int B, int F, int C, int H, int W, hipStream_t s);
int launch_philox_normal(float* out, int B, long long per_traj, uint64_t seed, long long traj0, long long draw,
                         hipStream_t s);

// Burgers sampler (update.hip)
int launch_burgers_prepare(float* img, float* x_w, const float* u0, const float* uT, int B, int nt, int nx, int cond_idx,
                           int set_zero, hipStream_t s);
int launch_ddpm_update_burgers(const float* x, const float* eps_uw, const float* eps_w, const float* z,
                               const float* u_target, float* x_next, float* x0_out, float* eps_out,
                               const dpc_burgers_coef& c, int B, int nt, int nx, hipStream_t s);

int launch_ddpm_update_jelly(const float* x, const float* eps, const float* eps_g, const float* z, float* pred,
                             float* x0_out, const dpc_jelly_coef& c, int B, int F, int Cx, int ns, int H, int W,
                             hipStream_t s);
int launch_jelly_guidance(float* io, const float* g, const float* eps_w, float eta_J, float eta_w, int pad_w, float sign,
                          int B, int F, int Cd, int H, int W, hipStream_t s);

// ---------------------------------------------------------------- PDE evaluators
int launch_burgers_fd(const float* u0, const float* f, float* traj, int N, int nx, int num_t, double visc, double T,
                      double dt, hipStream_t s);

}  // namespace dpc
