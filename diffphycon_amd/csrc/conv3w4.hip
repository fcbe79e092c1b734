// Conv3d 3x3x3 / stride 1 / pad 1, channels-last, f16x3 arithmetic -- Winograd F(4,3) along the FRAME axis (r06), on the loader-wave /
// persistent skeleton of conv3w.hip (F(2,3)), which it replaces as the default (DPC_DEBUG=1 DPC_CONV3W_F43=0 selects the old form).
//
// Why: the F(2,3) kernel runs its MFMA stream at the part's power-managed rate, so the only lever left is fewer matrix
// products.  F(4,3) computes FOUR output frames from six transformed input frames with 6 instead of 8 (F(2,3)) or 12 (direct) frame-tap
// products: 54 (h, w, frame-component) tap products per four output frames instead of 72 -- 3/4 of conv3w's MFMAs, 1/2 of the direct form's.
// Interpolation points (0, 1, -1, 1/2, -2, inf) -- not Lavin's (0, +-1, +-2, inf): on the f16x3 products their worst-case error at the
// S64 layer shapes is 0.5-0.6 of the standard set's and 2-3 x F(2,3)'s (tools/winograd_f43_error.py, profiles/r06_a_winograd_f43_error.log):
//     V = B^T d (input frames d0..d5 = f0-1 .. f0+4)                     U = G g (frame taps g0..g2 of one (dh, dw, cin, cout))
//     V0 =  d0 - 1.5 d1 - 2 d2 + 1.5 d3 + d4                             U0 = g0
//     V1 =     -     d1 + .5 d2 + 2.5 d3 + d4                            U1 = ( g0 + g1 + g2) / 3
//     V2 =           d1 - 2.5 d2 + .5 d3 + d4                            U2 = (-g0 + g1 - g2) / 3
//     V3 =      -  2 d1 -    d2 +  2 d3 + d4                             U3 = -(16 g0 + 8 g1 + 4 g2) / 15
//     V4 =       .5 d1 -    d2 - .5 d3 + d4                              U4 = (g0 - 2 g1 + 4 g2) / 15
//     V5 =           d1 - 1.5 d2 - 2 d3 + 1.5 d4 + d5                    U5 = g2
//     m_k = sum over (dh, dw, cin) U_k V_k
//     out(f0)   = m0 + m1 + m2 + m3 + m4              out(f0+1) = m1 - m2 + m3 / 2 - 2 m4
//     out(f0+2) = m1 + m2 + m3 / 4 + 4 m4             out(f0+3) = m1 - m2 + m3 / 8 - 8 m4 + m5
// (tests/test_winograd_identity.py restates exactly this and checks it against the direct convolution).
//   * Output tile 4 x 8 x 8, 64 output channels; 6 input frames x 10 x 10 halo per 16-channel chunk -- the SAME halo loads as conv3w.
//   * Loader waves 4-7: as conv3w (raw buffer loads one chunk ahead, hand-counted vmcnt, fused GroupNorm + SiLU), then B^T in fp32
//     (16 multiply-adds per element), the fp16 split and 6 (not 8) transformed frames to the double-buffered swizzled halo (600 points x
//     64 B per buffer).  |V| <= 7 |d|: operand pre-scale 2 (plain input; fp16 range ends at |x| = 4678) or log2(e) (fused activation;
//     it falls out of the SiLU evaluation: 6486) -- beyond it the operand is inf and the output NaN, as in conv3w.
//   * MFMA waves 0-3: six components do not divide over four waves, and the register file (256 per wave with a loader wave on the
//     same SIMD) holds 6 accumulator sets of 64 points x 64 channels per CU but not 12.  Wave w owns component w (64 points x 64
//     channels) AND one 32-channel half (w & 1) of component 4 + (w >> 1): 6 accumulator tiles (96 registers), 18 MFMAs per tap, 162
//     per chunk (conv3w: 216).  Price: each weight fragment now serves 64 points instead of 128 -- 2 x the L2 -> L1 weight bytes per MFMA
//     (measured on conv3w with doubled weight requests: + 10 %, profiles/r06_a_wtraffic_ab.log).  Transformed weights
//     [6][9 taps][chunk][n / 32][2 planes][2 k-halves][32 n][8] fp16 (launch_pack_weights_w4: fragment order, a load is 1 KB
//     contiguous), two running pointers, 3-deep register ring.
//     Fragment re-loads keep conv3w's rule (a set is re-loaded >= 4 MFMAs after its last reader was issued and 8 MFMAs before its
//     next reader: DESIGN.md 6.2, third hazard): three sets -- component A rows 0-3, rows 4-7, component B -- rotate through the
//     three 6-MFMA groups of a tap.  (Measured and dropped, profiles/r06_e_term_major_ab.log: all four sets double-buffered and the
//     18 MFMAs of a tap in term-major order over the six tiles -- an accumulator revisited after 6 instead of 2 MFMAs -- is 11 % SLOWER;
//     profiles/r06_j_spread_ab.log: the six weight requests of a tap spread over the tap, one behind every second MFMA, instead of six in
//     a row: no difference (+0.7 %).)  What bounds the tap phase is the CU's vector-memory path: 216 KB of weight fragments + 38 KB of halo
//     per chunk through 64 B / clk = 3970 of the chunk's 5184 MFMA-issue cycles (profiles/r06_i_stamps_dbg.log: without the weight
//     stream the tap phase is 0.64 of its length; r06_h_stamps.log: taps 7.7 k cycles per chunk, the loader waves idle 45 % of theirs).
//   * Epilogue: ONE exchange per tile (conv3w: one per frame pair).  The four MFMA waves park all six components in LDS (96 KB: halo
//     buffer 1, which every tile leaves last, and the otherwise unused tail), the loader waves read them between two barriers --
//     wave l takes plane rows 2l, 2l+1 of all four output frames and both channel halves: 6 reads per 4 output values -- combine,
//     add bias, emit the GroupNorm partial sums (entry l of the tile: 4 entries per tile as conv3w, the finalize pass sums all
//     entries) and store full 128-byte lines; frames f0+2, f0+3 drain behind the next tile's halo production.
// Rounding: U in fp32 from the fp32 weights before the split, V in fp32 after the fused activation; the same 22-bit f16x3 products
// with fp32 accumulation.  tests/test_gpu_ops.py bounds the per-convolution error (see there for the measured values).
// Perf attribution (DPC_ENABLE_CONV_DBG builds, env DPC_CONV_DBG; results INVALID): 2 the loader skips its global loads, 32 the
// loader does nothing but the barriers, 8 no epilogue, 4 every MFMA wave streams component 0's weights for both of its streams,
// 16 weight fragments are loaded once (three taps' worth, then re-used), 64 activation fragments are read from LDS during the first chunk only.
// Compile-time attribution (tools/conv_stamps_w4.py with -DDPC_CONV_STAMPS; no branch enters the instruction stream, unlike the bits):
// -DDPC_W4_ATTR_NOW no weight loads after the first three taps, _NOA no activation-fragment LDS reads after the first chunk, _NOL the loader
// requests its halo but does not process it, _NOH the loader requests nothing.  Measured (profiles/r06_l_stamps_attr.log, 256 -> 256 at
// 16 x 16, shader cycles per chunk of the MFMA waves' tap phase; 5184 = back-to-back MFMA issue): product 7765, NOW 6079, NOA 7356,
// NOL 7584, NOH 6740, NOW+NOL 5625, NOW+NOA+NOL 5376.  I.e. the tap phase runs at 1.50 x its issue time and the excess is the CU's
// vector-memory path: the MFMA waves' own weight requests cost them 1.7 k cycles per chunk and the loader waves' 42 halo requests another
// 1.0 k; the loader's arithmetic costs 0.2 k, the LDS fragment reads 0.4 k.
// Reference op: nn.Conv3d(dim, dim_out, (3,3,3), padding=(1,1,1)) in Block (video_diffusion_pytorch_conv3d.py:189-204).
#include "common.h"
#include "f3c.h"

namespace dpc {

namespace w4 {
constexpr float SAW = 2.0f;                 // activation pre-scale of the un-normalised path (an inline constant of v_fma_mix)
constexpr int TFO = 4;                      // output frames per tile
constexpr int HFI = 6;                      // input halo frames = transformed frames
constexpr int NCOMP = 6;
constexpr int ITEMS = 400;                  // loader work items: 100 (h, w) x 4 channel quads
constexpr int FPL = 6400;                   // bytes of one transformed halo frame: 100 points x 64 B
constexpr int HB1 = 40960;                  // halo buffer 1 (buffer 0 at 0; 38400 bytes used of each)
constexpr int XCH = HB1;                    // epilogue exchange area [6 components][2 channel halves][64 points][128 B] = 96 KB
constexpr int LDS_BYTES = XCH + NCOMP * 16384;
static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");
}  // namespace w4

// GN: the loader applies the producer's GroupNorm + (scale, shift) + SiLU (Conv3hParams::in_coef)
// NC: 64 / 128 = the profile class of the layer (N == 64 / N > 64); the code does not depend on it
template <bool GN, int NC>
__global__ __launch_bounds__(512, 2) void conv3w4_kernel(Conv3hParams p) {
    using namespace f3c;
    using namespace w4;
    constexpr int NTAPS = 9;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w4[];
    unsigned char* halo = smem_w4;                      // two buffers at 0 and HB1

    // operand pre-scales undone in the epilogue: activations SAW = 2 (plain input) or log2(e) (fused GroupNorm + SiLU, see the loader)
    const float descale = GN ? (float)(1.0 / (1.4426950408889634 * 4096.0)) : 1.0f / (SAW * SW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / 64;
    const int ntf = (p.F + TFO - 1) / TFO, nth = p.H / 8, ntw = p.W / 8;
    const int K = p.C0 + p.C1, kchunks = p.kchunks;
    const int nb = p.total_wg;
    const int ntiles = nb > (int)blockIdx.x ? (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const long long nsteps = (long long)ntiles * kchunks;
    // XCD-aware order, as conv3f3c: consecutive tile indices (shared halo planes, same weights) stay on one XCD
    auto decode = [&](int j, int& n0, int& w0, int& h0, int& f0, int& b) {
        int bid = (int)blockIdx.x + j * (int)gridDim.x;
        {
            const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        n0 = (bid % ntn) * 64;
        int t = bid / ntn;
        w0 = (t % ntw) * 8; t /= ntw;
        h0 = (t % nth) * 8; t /= nth;
        f0 = (t % ntf) * TFO;
        b = t / ntf;
    };
    if (nsteps == 0) return;
#ifdef DPC_CONV_STAMPS
    // attribution builds (tools/conv_stamps_w4.py; results INVALID: the totals overwrite the head of the output): per wave, shader cycles
    // spent in each phase, summed over the launch.  MFMA waves: 0 taps, 1 accumulator drain, 2 chunk barrier, 3 epilogue writes,
    // 4 epilogue barriers.  Loader waves: 0 request, 1 wait for the halo, 2 activation / transform / split / LDS writes, 3 deferred stores,
    // 4 chunk barrier, 5 epilogue.  Slot 7: lifetime.
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    const unsigned long long tbirth = tlast;
    auto lap = [&](int k) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tacc[k] += t - tlast; tlast = t; };
    auto lap_out = [&]() {
        tacc[7] = __builtin_amdgcn_s_memtime() - tbirth;
        if (lane == 0) {
            float* rec = p.out + ((long long)blockIdx.x * 8 + wave) * 8;
            for (int i = 0; i < 8; ++i) rec[i] = (float)tacc[i];
        }
    };
#else
    auto lap = [&](int) {};
    auto lap_out = [&]() {};
#endif

    if (wave >= 4) {
        // ======================================================================================= loader waves
        // (see conv3w.hip for why this path is written the way it is: raw buffer loads with hardware range checking, wave-uniform
        // frame validity, one fused multiply-add from the raw input to the activation argument, hand-counted vmcnt)
        __builtin_amdgcn_s_setprio(2);
        const int ltid = tid - 256;
        const bool two = ltid + 256 < ITEMS;              // threads 0..143 own a second item
        int hdst[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = ltid + 256 * i, hw = q >> 2, quad = q & 3;
            hdst[i] = slot0(0, hw / 10, hw % 10, quad >> 1) + (quad & 1) * 8;
        }
        const int hslot = (ltid & 3) * 4;
        constexpr unsigned OOB = 0xC0000000u;             // >= num_records of every buffer (conv3w_supported): the load returns 0
        const int fstride = p.H * p.W;
        const unsigned nrec0 = (unsigned)((long long)p.F * fstride * p.C0 * 4), nrec1 = (unsigned)((long long)p.F * fstride * p.C1 * 4);
        // issue-stage tile state
        int hpt[2];                                       // point index of (frame f0 - 1, h, w); may be negative
        int inhw[2];                                      // 1: the item's (h, w) lies inside the plane
        unsigned fokm = 0;                                // wave-uniform: bit fi = input frame f0 - 1 + fi exists
        const float* xb0 = nullptr;
        const float* xb1 = nullptr;
        int b_cur = 0;
        auto setup_tile = [&](int j) {
            int n0, w0, h0, f0, b;
            decode(j, n0, w0, h0, f0, b);
            b_cur = b;
            xb0 = p.a0 + (long long)b * p.F * fstride * p.C0;
            xb1 = p.a1 ? p.a1 + (long long)b * p.F * fstride * p.C1 : nullptr;
            fokm = 0;
#pragma unroll
            for (int fi = 0; fi < HFI; ++fi)
                if ((unsigned)(f0 - 1 + fi) < (unsigned)p.F) fokm |= 1u << fi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hw = (ltid + 256 * i) >> 2;
                const int h = h0 - 1 + hw / 10, w = w0 - 1 + hw % 10;
                inhw[i] = ((i == 0 || two) && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) ? 1 : 0;
                hpt[i] = ((f0 - 1) * p.H + h) * p.W + w;
            }
        };
        // Two-stage pipeline (conv3w.hip): the raw halo of step s + 2 is requested before step s + 1 is activated, transformed, split
        // and written.  The loads are inline asm and the waits hand-counted; no other vector-memory LOAD exists on this path.
        constexpr int NLOADS = 12 + (GN ? 2 : 0);
        static_assert(NLOADS == (GN ? 14 : 12), "the s_waitcnt immediates in landed() are NLOADS");
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto issue = [&](int kc, f32x4 (&d)[2][HFI], f32x4 (&cf)[2]) {
            if (GN) {
                const int c = kc * KC + hslot;
                const f32x4* src = reinterpret_cast<const f32x4*>(p.in_coef + (long long)p.B * K * 5) +
                                   ((long long)b_cur * (K >> 2) + ((c < K ? c : 0) >> 2)) * 2;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cf[i]) : "v"(src + i) : "memory");
            }
            const int c0 = kc * KC;                       // wave-uniform: a chunk lies in ONE source (C0 % 16 == 0 with a concat)
            const bool s1 = c0 >= p.C0;
            const int cs = s1 ? p.C1 : p.C0;
            const int cc = c0 - (s1 ? p.C0 : 0) + hslot;
            const unsigned long long base = reinterpret_cast<unsigned long long>(s1 ? xb1 : xb0);
            i32x4 rs;                                     // raw buffer resource: base, stride 0, num_records (bytes), 32-bit data format
            rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
            rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
            rs.z = __builtin_amdgcn_readfirstlane((int)(s1 ? nrec1 : nrec0));
            rs.w = 0x00020000;
#ifdef DPC_W4_ATTR_NOH          // attribution builds: no halo loads (compile-time form of bit 2)
            const bool cok = false;
            if (true) return;
#else
            const bool cok = c0 + hslot < K && !(CONV_DBG_BUILD && (p.dbg & 2));
#endif
            const int fbytes = fstride * cs * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned v0 = (inhw[i] != 0 && cok) ? (unsigned)((hpt[i] * cs + cc) * 4) : OOB;
#pragma unroll
                for (int fi = 0; fi < HFI; ++fi) {        // frames outside the tensor: negative / past-the-end offsets of the per-sample
                                                          // buffer read 0 as well -- all 12 loads are unconditional
                    const unsigned vo = v0 + (unsigned)(fi * fbytes);
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d[i][fi]) : "v"(vo), "s"(rs) : "memory");
                }
            }
        };
        // wait until at most `newer` younger loads are in flight, i.e. until everything requested for (d, cf) has landed
        auto landed = [&](f32x4 (&d)[2][HFI], f32x4 (&cf)[2], bool newer) {
            if (newer) {
                if (GN) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int fi = 0; fi < HFI; ++fi) asm volatile("" : "+v"(d[i][fi]));
            if (GN) {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(cf[i]));
            }
        };
        // activation + B^T + split + LDS write of one item.
        //   GN: z = fma(x, A, B) = y log2 e with the folded coefficients (launch_gn_finalize_fused), SiLU(y) log2(e) = z * rcp(1 + exp2(-z)):
        //       the operand pre-scale of this kernel is log2 e (undone in the epilogue).  Lanes outside the plane get A = B = 0 (the zero
        //       padding applies to the ACTIVATED tensor), frames outside the tensor are skipped wave-uniformly.
        //   split: hi = f16(s v), lo = f16(s v - hi), s = 2 folded into v_fma_mix on the plain path.  No clamp (see the header).
        auto finish_item = [&](f32x4 (&d)[HFI], unsigned fok, int inflag, const f32x4& Ac, const f32x4& Bc, int dst0) {
            if (GN) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                const bool in = inflag != 0;
                const f32x4 A = in ? Ac : zero, B = in ? Bc : zero;
#pragma unroll
                for (int fi = 0; fi < HFI; ++fi)
                    if ((fok >> fi) & 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float z = __builtin_fmaf(d[fi][e], A[e], B[e]);
                            d[fi][e] = z * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(-z) + 1.0f);
                        }
                        if (fi & 1) __builtin_amdgcn_sched_barrier(0);
                    }
            }
            // plane 1 of a point is its plane-0 address ^ 32; frame offsets are multiples of 64, so both planes take them as
            // ds_write immediates on top of two base registers
            unsigned char* q0 = halo + dst0;
            unsigned char* q1 = halo + (dst0 ^ 32);
            auto put = [&](int k, const f32x4& v) {
                uint2 p1, p2;
                if constexpr (GN) {           // the fused activation already carries its scale (log2 e): plain convert + remainder
                    p1.x = cvt_pk_f16(v.x, v.y);
                    p1.y = cvt_pk_f16(v.z, v.w);
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(p2.x) : "v"(v.x), "v"(v.y), "v"(p1.x));
                    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(p2.y) : "v"(v.z), "v"(v.w), "v"(p1.y));
                } else {                      // plain input: h1 = fp16(2 x), h2 = fp16(2 x - h1)
                    asm("v_fma_mixlo_f16 %0, %1, 2.0, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, 2.0, 0 op_sel_hi:[0,0,0]"
                        : "=&v"(p1.x) : "v"(v.x), "v"(v.y));
                    asm("v_fma_mixlo_f16 %0, %1, 2.0, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, 2.0, 0 op_sel_hi:[0,0,0]"
                        : "=&v"(p1.y) : "v"(v.z), "v"(v.w));
                    asm("v_fma_mixlo_f16 %0, %1, 2.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 2.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(p2.x) : "v"(v.x), "v"(v.y), "v"(p1.x));
                    asm("v_fma_mixlo_f16 %0, %1, 2.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 2.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                        : "=&v"(p2.y) : "v"(v.z), "v"(v.w), "v"(p1.y));
                }
                *reinterpret_cast<uint2*>(q0 + k * FPL) = p1;
                *reinterpret_cast<uint2*>(q1 + k * FPL) = p2;
            };
            // B^T, element-wise on purpose (packed f32 VALU ops cost the co-resident MFMA wave more than the scalar ops they replace;
            // the file is built with -fno-slp-vectorize and -ffp-contract=off: every fma below is written out)
            f32x4 a, b, t;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = d[4][e] - d[2][e]; b[e] = d[3][e] - d[1][e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(2.0f, b[e], a[e]);
            put(3, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(-0.5f, b[e], a[e]);
            put(4, t);
            __builtin_amdgcn_sched_barrier(0);            // (two transformed frames at a time: keeps the loader inside the register budget)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(1.5f, b[e], __builtin_fmaf(-2.0f, d[2][e], d[0][e] + d[4][e]));
            put(0, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(1.5f, a[e], __builtin_fmaf(-2.0f, d[3][e], d[5][e] + d[1][e]));
            put(5, t);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(2.5f, d[3][e], __builtin_fmaf(0.5f, d[2][e], d[4][e] - d[1][e]));
            put(1, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(0.5f, d[3][e], __builtin_fmaf(-2.5f, d[2][e], d[4][e] + d[1][e]));
            put(2, t);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto finish = [&](int kc, unsigned fok, int in0, int in1, f32x4 (&d)[2][HFI], const f32x4 (&cf)[2], int boff) {
            if (CONV_DBG_BUILD && (p.dbg & 32)) return;
#ifdef DPC_W4_ATTR_NOL
            return;
#endif
            const bool cok = kc * KC + hslot < K;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 A = (GN && cok) ? cf[0] : zero, B = (GN && cok) ? cf[1] : zero;
            finish_item(d[0], fok, in0, A, B, hdst[0] + boff);
            if (two) finish_item(d[1], fok, in1, A, B, hdst[1] + boff);
        };
        // ---- output transform + bias + GroupNorm partial sums + stores of a finished tile (see the MFMA waves' epilogue).
        // Exchange layout (written by the MFMA waves): component k, channel half nt, point P = h * 8 + w of the 8 x 8 plane tile is a
        // 128-byte row of 32 channels at ((k * 2 + nt) * 64 + P) * 128; its eight 16-byte channel quads sit at slot (quad ^ (P & 7)).
        // A loader lane takes column w = lane >> 3 and quad j = lane & 7: eight lanes read one full row (conflict-free), and a store
        // instruction writes 8 points x 128 contiguous bytes.  Loader wave l: unit u = 0..3 = (plane row 2 l + (u >> 1), channel half u & 1),
        // all four output frames of it.
        const int lwv = wave - 4;
        f32x4 pv1[8];                                     // frames f0+2, f0+3 of the previous tile's four units, not yet stored (see drain)
        // store addressing: a wave-uniform 64-bit base per store (SGPRs: sample, frame, row of the unit, channel half) + ONE 32-bit lane
        // offset (column, channel quad) -- 64-bit lane addresses per row cost the loader 2 registers each
        const float* pend_sb = nullptr;                   // uniform base of (frame f0 + 2, row 2 l, half 0) of the pending tile
        bool pend_ok0 = false, pend_ok1 = false;
        const int pend_sh = kchunks >= 8 ? 0 : kchunks >= 4 ? 1 : 2;      // 8 stores over min(8, kchunks) steps
        const long long ostr_f = (long long)p.H * p.W * p.N, ostr_h = (long long)p.W * p.N;
        const unsigned lane_off = (unsigned)(((lane >> 3) * p.N + 4 * (lane & 7)) * 4);
        // element offset of pending / immediate row t = 2 u + (frame & 1) relative to (frame pair base, row 2 l, half 0)
        auto row_off = [&](int t) { return (long long)(t & 1) * ostr_f + (long long)(t >> 2) * ostr_h + ((t >> 1) & 1) * 32; };
        auto store_row = [&](const float* sb, const f32x4& v) {
            // (s_nop: a VALU write to the data registers of a >64-bit store needs 2 wait states, and the hazard recogniser does not
            // look into inline asm)
            asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(lane_off), "v"(v), "s"(sb) : "memory");
        };
        auto epilogue = [&](int j, const unsigned char* xch) {
            int n0, w0, h0, f0, b;
            decode(j, n0, w0, h0, f0, b);
            const int col = lane >> 3, quad = lane & 7;
            const long long tile = ((long long)(f0 / TFO) * nth + h0 / 8) * ntw + w0 / 8;
            f32x4 o0[8];                                   // frames f0, f0+1 of the four units: [2 u + frame]
            // (the lane's exchange offset passes through an empty asm: the 24 read addresses derived from it -- the area exceeds the 64 KB
            // reach of a ds_read offset -- are then formed here, not hoisted into registers that live across the whole tile loop)
            int xo = ((2 * lwv * 8 + col) * 128) + ((quad ^ col) << 4);
            asm volatile("" : "+v"(xo));
            wg_barrier();                                  // E1: the six components are in LDS
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ntr = u & 1;
                const unsigned char* xl = xch + xo + (ntr * 64 + (u >> 1) * 8) * 128;
                auto rd = [&](int k) { return *reinterpret_cast<const f32x4*>(xl + k * 16384); };
                {
                    const f32x4 m1 = rd(1), m2 = rd(2), m3 = rd(3), m4 = rd(4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s = m1[e] + m2[e], dd = m1[e] - m2[e];
                        o0[2 * u][e] = (s + m3[e]) + m4[e];
                        o0[2 * u + 1][e] = __builtin_fmaf(-2.0f, m4[e], __builtin_fmaf(0.5f, m3[e], dd));
                        pv1[2 * u][e] = __builtin_fmaf(4.0f, m4[e], __builtin_fmaf(0.25f, m3[e], s));
                        pv1[2 * u + 1][e] = __builtin_fmaf(-8.0f, m4[e], __builtin_fmaf(0.125f, m3[e], dd));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);         // (four + two reads of a unit in flight: registers)
                {
                    const f32x4 m0 = rd(0), m5 = rd(5);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o0[2 * u][e] += m0[e]; pv1[2 * u + 1][e] += m5[e]; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            lds_done_barrier();                            // E2: the buffer returns to the halo pipeline; the MFMA waves go on
            // de-scale + bias, GroupNorm partial sums of the OUTPUT (this lane: 4 channels x 2 rows x valid frames of one column, one
            // channel half at a time), dwordx4 stores (asm: they must not enter hipcc's vmcnt bookkeeping of this path; they retire in
            // order with the loads, so a later vmcnt(12 | 14) also waits for them)
            const float* sb0 = p.out + ((((long long)b * p.F + f0) * p.H + h0 + 2 * lwv) * p.W + w0) * p.N + n0;
            pend_sb = sb0 + 2 * ostr_f;
            pend_ok0 = f0 + 2 < p.F;
            pend_ok1 = f0 + 3 < p.F;
            float* gdst = p.gn_part ? p.gn_part + ((((long long)b * ((long long)ntf * nth * ntw) + tile) * 4 + lwv) * p.N + n0 + 4 * quad) * 2 : nullptr;
#pragma unroll
            for (int ntr = 0; ntr < 2; ++ntr) {
                float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
                const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + 4 * quad + ntr * 32) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int t = (tt >> 1) * 4 + ntr * 2 + (tt & 1);      // rows of this half: units ntr, ntr + 2, both frame parities
                    const bool ok0 = f0 + (t & 1) < p.F, ok1 = (t & 1) ? pend_ok1 : pend_ok0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = o0[t][e] * descale + bv[e];
                        const float v1 = pv1[t][e] * descale + bv[e];
                        o0[t][e] = v;
                        pv1[t][e] = v1;
                        if (ok0) { gs[e] += v; gq[e] += v * v; }
                        if (ok1) { gs[e] += v1; gq[e] += v1 * v1; }
                    }
                    if (ok0) store_row(sb0 + row_off(t), o0[t]);
                }
                if (gdst) {
#pragma unroll
                    for (int m = 8; m <= 32; m <<= 1)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gs[e] += __shfl_xor(gs[e], m, 64);
                            gq[e] += __shfl_xor(gq[e], m, 64);
                        }
                    if (lane < 8) {
                        const f32x4 t0 = {gs[0], gq[0], gs[1], gq[1]}, t1 = {gs[2], gq[2], gs[3], gq[3]};
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(gdst + ntr * 64), "v"(t0) : "memory");
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(gdst + ntr * 64 + 4), "v"(t1) : "memory");
                    }
                }
            }
        };
        // Half of a tile's row stores (frames f0+2, f0+3: 8 per lane) are NOT issued in the epilogue: every CU finishes its tile at the
        // same moment and the simultaneous stores run at the HBM write limit.  They drain one or two per step behind the NEXT tile's
        // halo production (conv3w.hip).
        auto drain = [&](int slot) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if ((t >> pend_sh) == slot && ((t & 1) ? pend_ok1 : pend_ok0)) store_row(pend_sb + row_off(t), pv1[t]);
        };
        struct StepState { unsigned fok; int in0, in1; int kc; f32x4 cf[2]; };
        f32x4 ra[2][HFI], rb[2][HFI];                     // raw halo registers of two steps in flight (roles alternate: no copies)
        StepState sa{}, sb{};
        int lj = 0, lkc = 0;                              // cursor of the issue stage
        auto advance = [&]() { if (++lkc == kchunks) { lkc = 0; ++lj; setup_tile(lj < ntiles ? lj : ntiles - 1); } };
        auto request = [&](f32x4 (&d)[2][HFI], StepState& st) {
            issue(lkc, d, st.cf);
            st.fok = fokm; st.in0 = inhw[0]; st.in1 = inhw[1]; st.kc = lkc;
        };
        setup_tile(0);
        request(ra, sa);
        if (nsteps > 1) { advance(); request(rb, sb); }
        landed(ra, sa.cf, nsteps > 1);
        finish(sa.kc, sa.fok, sa.in0, sa.in1, ra, sa.cf, 0);
        lds_done_barrier();                               // buffer 0 holds step 0
        // iteration s: X = raw data of step s + 1 (requested one iteration ago); request step s + 2 into Y, finish X.  A tile is an
        // EVEN number of steps (conv3w_supported: K % 32 == 0), so the two register sets alternate without copies, the epilogue
        // exists once in the code and the buffer a tile leaves behind is always buffer 1.
        auto body = [&](long long s, f32x4 (&X)[2][HFI], StepState& sx, f32x4 (&Y)[2][HFI], StepState& sy, int boff, int slot) {
            if (s + 2 < nsteps) { advance(); request(Y, sy); }
            lap(0);
            if (s + 1 < nsteps) landed(X, sx.cf, s + 2 < nsteps);
            lap(1);
            if (s + 1 < nsteps) finish(sx.kc, sx.fok, sx.in0, sx.in1, X, sx.cf, boff);   // the MFMA waves left that buffer at the previous barrier
            lap(2);
            drain(slot);
            lap(3);
            lds_done_barrier();
            lap(4);
        };
        long long s = 0;
        for (int j = 0; j < ntiles; ++j) {
            for (int kc = 0; kc < kchunks; kc += 2) {
                body(s, rb, sb, ra, sa, HB1, kc);
                body(s + 1, ra, sa, rb, sb, 0, kc + 1);
                s += 2;
            }
            if (!(CONV_DBG_BUILD && (p.dbg & 8))) epilogue(j, halo + XCH);
            lap(5);
        }
        for (int slot = 0; slot < 8; ++slot) drain(slot);       // the last tile
        __builtin_amdgcn_s_waitcnt(0);
        lap_out();
        return;
    }

    // =========================================================================================== MFMA waves
    // wave w: component A = w, all 64 channels (tiles accA[slab q][nt]); component B = 4 + (w >> 1), channel half w & 1 (accB[slab q])
    int lh, lw;
    lane_hw(l31, lh, lw);
    const int compA = wave, compB = 4 + (wave >> 1), ntB = wave & 1;
    int aaddr[9];                                         // fragment address of component A, rows 0-3, buffer 0
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) aaddr[dh * 3 + dw] = slot0(0, lh + dh, lw + dw, hh) + compA * FPL;
    const int offB = (compB - compA) * FPL;

    f32x16 accA[2][2], accB[2];
    f16x8 wA[3][2][2], wB[3][2];
    f16x8 aA[2][2], aB[2][2];

    // weight streams: taps 0..8 of chunk 0, ..., of the last chunk, then the next tile; stream B = stream A + a wave-uniform distance
    const long long wstride = (long long)p.Npad * WROW, wtap = wstride * kchunks;
    const bool dbg_w0 = CONV_DBG_BUILD && (p.dbg & 4);   // attribution: every wave streams component 0's weights, stream B = stream A (L1 hits)
    const unsigned char* wroot = reinterpret_cast<const unsigned char*>(p.wpw) + (dbg_w0 ? 0ll : (long long)compA * NTAPS * wtap);
    const long long wdelta = dbg_w0 ? 0ll : (long long)(compB - compA) * NTAPS * wtap + (long long)ntB * 32 * WROW;
    // FRAGMENT-order pack (as the implicit GEMM's since r03): per (component, tap, chunk) and 32-channel column block
    // [plane 2][k-half 2][n 32][8 fp16] = 2 KB, so the 16 bytes lane (n, k-half) feeds to one MFMA are contiguous across the wave -- a
    // fragment load is 1 KB = 8 cache lines (the [n][plane][16 k] rows of conv3w: 32 half-used 64-byte rows per instruction)
    const int wlo = hh * 512 + l31 * 16;
    const unsigned char* wlane = wroot;
    const unsigned char* wnext = wroot;
    int wtap_i = 0, wkc_i = 0, wtile = 0;
    auto tile_n0 = [&](int j) {
        int n0, w0, h0, f0, b;
        decode(j < ntiles ? j : ntiles - 1, n0, w0, h0, f0, b);
        return n0;
    };
    bool dbg_w_once = false;                              // attribution bit 16: weight fragments are loaded for the first two taps only
    auto ldw = [&](f16x8 (&da)[2][2], f16x8 (&db)[2]) {
#ifdef DPC_W4_ATTR_NOW          // compile-time form of bit 16 (no branch in the stream): tools/conv_stamps_w4.py attribution builds
        if (dbg_w_once) return;
#endif
        if (CONV_DBG_BUILD && (p.dbg & 16) && dbg_w_once) return;
        const unsigned char* src = wnext + wlo;
        if (++wtap_i == NTAPS) {
            wtap_i = 0;
            if (++wkc_i == kchunks) { wkc_i = 0; ++wtile; wlane = wroot + (long long)tile_n0(wtile) * WROW; }
            wnext = wlane + wkc_i * wstride;
        } else {
            wnext += wtap;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) da[nt][pl] = *reinterpret_cast<const f16x8*>(src + nt * 32 * WROW + pl * 1024);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) db[pl] = *reinterpret_cast<const f16x8*>(src + wdelta + pl * 1024);
    };
    wlane = wroot + (long long)tile_n0(0) * WROW;
    wnext = wlane;
    ldw(wA[0], wB[0]);
    ldw(wA[1], wB[1]);
    if (CONV_DBG_BUILD && (p.dbg & 16)) {
        ldw(wA[2], wB[2]);
        dbg_w_once = true;
    }
#ifdef DPC_W4_ATTR_NOW
    ldw(wA[2], wB[2]);
    dbg_w_once = true;
#endif

    int boff = 0;
    // fragments of slab q (plane rows 4 q .. 4 q + 3) of a component's transformed frame for tap (dh, dw): both planes
    bool dbg_a_once = false;                              // attribution bit 64: activation fragments are read from LDS once per launch
    auto lda = [&](f16x8 (&dst)[2], int addr, int q) {
#ifdef DPC_W4_ATTR_NOA
        if (dbg_a_once) return;
#endif
        if (CONV_DBG_BUILD && (p.dbg & 64) && dbg_a_once) return;
        dst[0] = *reinterpret_cast<const f16x8*>(halo + addr + q * 2560);
        dst[1] = *reinterpret_cast<const f16x8*>(halo + (addr ^ 32) + q * 2560);
    };

    wg_barrier();                                         // step 0 is in buffer 0
    for (int j = 0; j < ntiles; ++j) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { accA[q][0][r] = 0.f; accA[q][1][r] = 0.f; accB[q][r] = 0.f; }
        }
        for (int kc = 0; kc < kchunks; ++kc) {
            lda(aA[0], aaddr[0] + boff, 0);               // (component B's set follows four MFMAs into the first tap, see tap_body)
            lda(aA[1], aaddr[0] + boff, 1);
            auto tap_body = [&](int tap) {
                const int r = tap % 3;
                ldw(wA[(tap + 2) % 3], wB[(tap + 2) % 3]);   // two taps ahead (the ring runs on across chunks and tiles)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};     // small terms first; PA: activation plane, PB: weight plane
                // Three groups of six MFMAs: component A rows 0-3, rows 4-7, component B; each re-loads, behind its fourth MFMA, the set
                // whose last reader is four MFMAs back and whose next reader is eight MFMAs ahead.
                // ---- group 0: A rows 0-3; re-load component B's set (both slabs) of THIS tap
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    if (term == 2) {
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_keep(accA[0][0], aB[0][0], aB[0][1], aB[1][0], aB[1][1]);      // (spent B operands keep their registers up to here)
                        lda(aB[0], aaddr[tap] + (boff + offB), 0);
                        lda(aB[1], aaddr[tap] + (boff + offB), 1);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        accA[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wA[r][nt][PB[term]], aA[0][PA[term]], accA[0][nt], 0, 0, 0);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // ---- group 1: A rows 4-7; re-load A rows 0-3 of the NEXT tap
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    if (term == 2) {
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_keep(accA[1][0], aA[0][0], aA[0][1]);
                        if (tap < NTAPS - 1) lda(aA[0], aaddr[tap + 1 < NTAPS ? tap + 1 : tap] + boff, 0);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        accA[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wA[r][nt][PB[term]], aA[1][PA[term]], accA[1][nt], 0, 0, 0);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // ---- group 2: component B (one channel half), both slabs; re-load A rows 4-7 of the NEXT tap
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    if (term == 2) {
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_keep(accB[0], aA[1][0], aA[1][1]);
                        if (tap < NTAPS - 1) lda(aA[1], aaddr[tap + 1 < NTAPS ? tap + 1 : tap] + boff, 1);
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        accB[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wB[r][PB[term]], aB[q][PA[term]], accB[q], 0, 0, 0);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) tap_body(tap);
            lap(0);
            if (CONV_DBG_BUILD && (p.dbg & 64)) dbg_a_once = true;
#ifdef DPC_W4_ATTR_NOA
            dbg_a_once = true;
#endif
            // MFMA B-operand guard (third hazard): the next chunk's first fragment loads may be given the registers of the
            // fragments the last MFMAs read (hipcc re-uses them: tools/mfma_war_audit.py) -- the wave waits for the last MFMA of both
            // component-B chains (the youngest six MFMAs of the chunk) before it enters the barrier
            mfma_drain(accB[0]);
            mfma_drain(accB[1]);
            __builtin_amdgcn_sched_barrier(0);
            lap(1);
            wg_barrier();                                  // next chunk's buffer is complete; this one may be overwritten
            lap(2);
            boff = HB1 - boff;
        }

        // ---- epilogue.  acc[q][..][4g + e] = point (rows 4q.., lane_hw(l31)), channel nt*32 + 8g + 4hh + e of the component.
        // The output transform crosses waves AND is done by the loader waves: the four MFMA waves park their accumulators in the
        // exchange area (the last chunk's buffer 1 and the tail behind it), the loader waves read and combine between two barriers.
        if (CONV_DBG_BUILD && (p.dbg & 8)) continue;
        // (row of point P = (4 q + lh) * 8 + lw; this lane's quads 2 g + hh go to slot (2 g + hh) ^ (P & 7) = 2 g ^ (hh ^ lw))
        unsigned char* xw = halo + XCH + (lh * 8 + lw) * 128;
        const int tsw = hh ^ lw;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {accA[q][nt][4 * g], accA[q][nt][4 * g + 1], accA[q][nt][4 * g + 2], accA[q][nt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(xw + compA * 16384 + nt * 8192 + q * 4096 + ((tsw ^ (2 * g)) << 4)) = v;
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {accB[q][4 * g], accB[q][4 * g + 1], accB[q][4 * g + 2], accB[q][4 * g + 3]};
                *reinterpret_cast<f32x4*>(xw + compB * 16384 + ntB * 8192 + q * 4096 + ((tsw ^ (2 * g)) << 4)) = v;
            }
        }
        lap(3);
        lds_done_barrier();                                // E1: the six components are in LDS
        wg_barrier();                                      // E2: the loader waves have read them
        lap(4);
    }
    lap_out();
}

bool conv3w_f43_enabled() {
    static const int ok = debug_switch("DPC_CONV3W_F43", 1);
    return ok != 0;
}

int launch_conv3w4(const Conv3hParams& p, hipStream_t s) {
    using namespace f3c;
    const long long tiles = (long long)p.B * ((p.F + 3) / 4) * (p.H / 8) * (p.W / 8);
    const long long nwg = tiles * (p.Npad / 64);
    DPC_REQUIRE(nwg < (1ll << 31), "conv3w4: too many tiles");
    static int ncu = 0;
    static DeviceOnce once;
    if (!once) {
        int dev = 0;
        hipDeviceProp_t prop;
        DPC_HIP(hipGetDevice(&dev));
        DPC_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = std::max(8, prop.multiProcessorCount / 8 * 8);
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w4_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w4_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w4_kernel<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w4_kernel<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, w4::LDS_BYTES));
        once = true;
    }
    Conv3hParams pd = p;
    pd.total_wg = (int)nwg;
    const unsigned grid = (unsigned)std::min<long long>(nwg, cu_budget(ncu));
    const bool wide = p.Npad % 128 == 0 && p.N > 64;       // profile class only (launch_conv3f3's ProfScope uses the same rule)
    if (p.in_coef) {
        if (wide) hipLaunchKernelGGL((conv3w4_kernel<true, 128>), dim3(grid), dim3(512), w4::LDS_BYTES, s, pd);
        else hipLaunchKernelGGL((conv3w4_kernel<true, 64>), dim3(grid), dim3(512), w4::LDS_BYTES, s, pd);
    } else {
        if (wide) hipLaunchKernelGGL((conv3w4_kernel<false, 128>), dim3(grid), dim3(512), w4::LDS_BYTES, s, pd);
        else hipLaunchKernelGGL((conv3w4_kernel<false, 64>), dim3(grid), dim3(512), w4::LDS_BYTES, s, pd);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ---- weight transform + pre-split: reference [N][K][3][3][3] fp32 -> [6 k][9 taps][kchunks][Npad / 32][2 planes][2 k-halves][32 n][8] fp16 (x 2^12)
__global__ void pack_weights_w4_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int N, int Npad, int K,
                                       int kchunks, int* __restrict__ ovf) {
    const long long total = (long long)54 * kchunks * Npad * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % 16);
        long long r = i / 16;
        const int n = (int)(r % Npad);
        r /= Npad;
        const int kc = (int)(r % kchunks);
        const int kt = (int)(r / kchunks);               // k * 9 + (dh * 3 + dw)
        const int k = kt / 9, t9 = kt % 9;
        const int c = kc * 16 + kk;
        float v = 0.f;
        if (n < N && c < K) {
            const float* g = w + ((long long)n * K + c) * 27 + t9;
            const float g0 = g[0], g1 = g[9], g2 = g[18];
            float u;
            switch (k) {
                case 0: u = g0; break;
                case 1: u = ((g0 + g2) + g1) * (1.0f / 3.0f); break;
                case 2: u = (g1 - (g0 + g2)) * (1.0f / 3.0f); break;
                case 3: u = -(__builtin_fmaf(16.0f, g0, __builtin_fmaf(8.0f, g1, 4.0f * g2))) * (1.0f / 15.0f); break;
                case 4: u = __builtin_fmaf(4.0f, g2, __builtin_fmaf(-2.0f, g1, g0)) * (1.0f / 15.0f); break;
                default: u = g2; break;
            }
            v = u * f3c::SW;
            if (!(fabsf(v) <= 65504.f)) atomicOr(ovf, 1);
            v = f3c::sat16(v);
        }
        const unsigned p1 = f3c::cvt_pk_f16(v, 0.f) & 0xffffu;
        const float h1 = (float)__builtin_bit_cast(f3c::f16x2, p1).x;
        const unsigned p2 = f3c::cvt_pk_f16(v - h1, 0.f) & 0xffffu;
        // fragment order: 32-column block n >> 5, then [plane][k-half][n & 31][8]
        unsigned short* dst = wp + (((long long)kt * kchunks + kc) * Npad + (n & ~31)) * 32 + ((kk >> 3) * 32 + (n & 31)) * 8 + (kk & 7);
        dst[0] = (unsigned short)p1;
        dst[512] = (unsigned short)p2;
    }
}

size_t conv3w4_packed_bytes(int Npad, int K) { return (size_t)54 * ((K + 15) / 16) * Npad * 64; }

int launch_pack_weights_w4(const float* w, void* wp, int N, int Npad, int K, hipStream_t s) {
    const int kchunks = (K + 15) / 16;
    const long long total = (long long)54 * kchunks * Npad * 16;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_w4_kernel, dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), N, Npad, K,
                       kchunks, f16x3_weight_overflow_flag());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
