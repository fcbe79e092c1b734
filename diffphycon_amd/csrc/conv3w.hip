// Conv3d 3x3x3 / stride 1 / pad 1, channels-last, f16x3 arithmetic -- Winograd F(2,3) along the FRAME axis on the loader-wave /
// persistent skeleton of conv3f3c.hip.
//
// Why: conv3f3c runs its 27-tap MFMA stream at the chip's power limit (MFMA-busy x clock is constant), so the
// only lever left for the dominant kernel is fewer matrix products.  The 1-D minimal-filtering form over frames computes two
// output frames from four transformed input frames with 4 instead of 6 tap products:
//     V0 = d0 - d2,  V1 = d1 + d2,  V2 = d2 - d1,  V3 = d1 - d3                    (input frames d0..d3 = 2p-1 .. 2p+2)
//     U0 = g0,  U1 = (g0 + g1 + g2) / 2,  U2 = (g0 - g1 + g2) / 2,  U3 = g2         (frame taps g0..g2 of one (dh, dw))
//     m_k = sum over (dh, dw, cin) U_k V_k;     out(2p) = m0 + m1 + m2,   out(2p+1) = m1 - m2 - m3
// i.e. four independent (1,3,3) convolutions of 9 taps per frame PAIR: 36 tap products per two output frames instead of 54
// (2/3 of the matrix work).  (h, w) stay direct: every tap is still an LDS offset of the MFMA operand fragment.
//   * Output tile 4 x 8 x 8 (two frame pairs), 64 output channels; 6 input frames x 10 x 10 halo per 16-channel chunk.
//   * Loader waves 4-7: request the 6 halo frames of an (h, w, channel-quad) item one chunk ahead (raw buffer loads, hand-counted
//     vmcnt), apply the producer's GroupNorm + (scale, shift) + SiLU, form V0..V3 of both pairs in fp32, split into the two fp16
//     planes and write the 8 TRANSFORMED frames to the double-buffered swizzled halo (800 points x 64 B).  Operand pre-scale:
//     2^3 for a plain input (folded into the split's v_fma_mix), 4 log2(e) for the fused activation (it falls out of the SiLU
//     evaluation); |V| <= 2 |d|, so the fp16 range ends at |x| = 4094 / 5676 (beyond it the operand is inf and the output NaN -- not
//     clamped; 4094 is the range every f16x3 kernel guarantees and the always-on sentinel watches, common.h).
//   * MFMA waves 0-3: wave k owns Winograd component k of the whole tile (4 slabs of 32 points x 64 channels, transposed
//     accumulators exactly as conv3f3c), streams ITS transformed weights U_k ([4][9 taps][chunk][n][2 planes][16] fp16 made by
//     launch_pack_weights_w3) with the same running pointer / 3-deep register ring, 9 taps x 24 MFMAs per chunk.
//   * Epilogue: the output transform crosses waves.  Per frame pair the four waves park their 64 x 64 component in the halo
//     buffer they just left (64 KB), then wave w reads the three components of output frame parity (w & 1), channel half (w >> 1),
//     combines, adds bias, emits the GroupNorm partial sums and stores dwordx4.  Four extra workgroup barriers per tile; the
//     loader waves join them (they would otherwise refill that buffer).
// Rounding: U_k are formed in fp32 from the fp32 weights before the split, V_k in fp32 after the fused activation; the products
// are the same 22-bit f16x3 products with fp32 accumulation.  F(2,3) has transform constants 1 and 1/2 only: against an fp64
// convolution the error stays inside the bound the direct kernels are tested to (tests/test_gpu_ops.py: 3e-6 of the output range,
// also for inputs scaled to 1e-3 where the un-prescaled remainder plane sits in fp16's subnormals), and the full-width U-Nets
// agree with the oracle to 2e-5 at the FULL extents of S64 / S128 / J128 (tests/test_gpu_unet3d.py: test_full_extent_vs_oracle).
// Perf attribution only (env DPC_CONV_DBG, Conv3hParams::dbg; results are INVALID except for 512): 2 the loader skips its global
// loads, 32 the loader does nothing but the barriers, 4 every MFMA wave streams component 0's weights, 8 no epilogue, 512 the second
// frame pair's stores are issued in the epilogue instead of deferred, 64 (r05) the loader keeps its global loads and its 16 LDS writes per
// item but does NO activation / transform / split -- it writes the raw loaded bits, masked to finite fp16 patterns (random mantissas: the
// MFMA stream sees live, changing operands, unlike bit 32 whose static LDS content lowers the power draw and raises the clock): the
// honest ceiling of a design in which something else (a producer pass + LDS-DMA) delivers the operand planes.
// Reference op: nn.Conv3d(dim, dim_out, (3,3,3), padding=(1,1,1)) in Block (video_diffusion_pytorch_conv3d.py:189-204).
#include "common.h"
#include "f3c.h"

namespace dpc {

namespace w3 {
constexpr float SAW = 8.0f;                 // activation pre-scale of the un-normalised path, applied inside the operand split
                                            // (v_fma_mix, the constant in an SGPR): 65504 / 8 / 2 (|V| <= 2 |d|) = 4094 is
                                            // exactly the activation range every f16x3 kernel guarantees, and the remainder plane
                                            // stays a NORMAL fp16 down to |x| = 2^-6 (r02's SAW = 1: 2^-3; inputs of magnitude 1e-3
                                            // then lost 4 bits -- measured 1.3e-5 of the output range instead of < 3e-6)
constexpr int TFO = 4;                      // output frames per tile
constexpr int HFI = 6;                      // input halo frames
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITEMS = 400;                  // loader work items: 100 (h, w) x 4 channel quads
}  // namespace w3

// GN: the loader applies the producer's GroupNorm + (scale, shift) + SiLU (Conv3hParams::in_coef)
// NC: 64 / 128 = the profile class of the layer (N == 64 / N > 64, as the direct kernels' column-tile width).  The code does not depend on
// it: it only makes rocprofv3 report the two classes bench.py distinguishes as separate kernels.
template <bool GN, int NC>
__global__ __launch_bounds__(512, 2) void conv3w_kernel(Conv3hParams p) {
    using namespace f3c;
    using namespace w3;
    constexpr int MT = 4, NT = 2, NTAPS = 9;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w3[];
    unsigned char* halo = smem_w3;                      // two buffers at 0 and HBS

    // operand pre-scales undone in the epilogue: activations SAW = 2^3 (plain input) or 4 log2(e) (fused GroupNorm + SiLU, see the loader)
    const float descale = GN ? (float)(1.0 / (4.0 * 1.4426950408889634 * 4096.0)) : 1.0f / (SAW * SW);
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
    // per wave: deltas between consecutive stamps ([0..26]) and the raw start / end clocks ([28..31]); tools/conv_stamps_w.py
    unsigned long long tst[29];
    int nst = 0;
    auto stamp = [&]() { if (nst < 29) tst[nst++] = __builtin_amdgcn_s_memtime(); };
    auto stamp_out = [&]() {
        const unsigned long long tend = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            float* rec = p.out + ((long long)blockIdx.x * 8 + wave) * 32;
            for (int i = 1; i < 28; ++i) rec[i - 1] = i < nst ? (float)(tst[i] - tst[i - 1]) : 0.f;
            unsigned* ru = reinterpret_cast<unsigned*>(rec);
            ru[28] = (unsigned)tst[0]; ru[29] = (unsigned)(tst[0] >> 32);
            ru[30] = (unsigned)tend; ru[31] = (unsigned)(tend >> 32);
        }
    };
#else
    auto stamp = [&]() {};
    auto stamp_out = [&]() {};
#endif
    stamp();

    if (wave >= 4) {
        // ======================================================================================= loader waves
        // The loader shares each SIMD's VALU issue with an MFMA wave that leaves it ~5 slots per 32-cycle MFMA, and a chunk is only
        // 216 MFMAs long: the per-chunk instruction count decides whether the matrix pipe waits.  Hence: raw buffer loads whose
        // out-of-range lanes read 0 (no exec-mask branches, 32-bit offsets), frame validity as wave-uniform branches, ONE fused
        // multiply-add from the raw input to the (pre-scaled) activation argument, exp2 / rcp hardware transcendentals.
        __builtin_amdgcn_s_setprio(2);                    // (0 ... 3 measured: no difference)
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
        int inhw[2];                                      // 1: the item's (h, w) lies inside the plane (a VGPR flag: lane masks held in
                                                          // SGPR pairs across the pipeline stages were spilling the scalar file)
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
        // Two-stage pipeline: the raw halo of step s + 2 is requested (issue) before step s + 1 is activated, transformed, split
        // and written (finish): the HBM / L2 latency of a chunk hides behind the VALU work and the barrier wait of the previous one.
        // The loads are inline asm and the waits hand-counted: hipcc's own vmcnt bookkeeping degrades to vmcnt(0) across this
        // loop's divergent `two` region and back edge, which made every finish stage wait for the loads issued just before it.
        // No other vector-memory instruction exists on the loader path, so the count is exact: NLOADS per request, in order.
        constexpr int NLOADS = 12 + (GN ? 2 : 0);
        static_assert(NLOADS == (GN ? 14 : 12), "the s_waitcnt immediates in landed() are NLOADS");
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        auto issue = [&](int kc, f32x4 (&d)[2][HFI], f32x4 (&cf)[2]) {
            if (GN) {
                // the folded GroupNorm coefficients (A, B) log2(e) of this chunk's channels (second table of in_coef, see
                // launch_gn_finalize_fused) are requested BEFORE the halo (in-order return)
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
            const bool cok = c0 + hslot < K && !(p.dbg & 2);
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
            // (the registers pass through an empty asm so that no use of them is scheduled above the wait)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int fi = 0; fi < HFI; ++fi) asm volatile("" : "+v"(d[i][fi]));
            if (GN) {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(cf[i]));
            }
        };
        // activation + transform + split + LDS write of one item.  Every instruction here displaces the co-resident MFMA wave
        // (measured: ~6 cycles of matrix-pipe time per loader instruction, tools/conv_stamps_w.py), so the arithmetic is folded:
        //   GN: Block.forward (...conv3d.py:196-204) is norm -> x (scale + 1) + shift -> SiLU.  With A = rstd gamma (scale + 1),
        //       B = (beta - mu rstd gamma)(scale + 1) + shift, both times log2(e):  z = fma(x, A, B) = y log2 e,
        //       SiLU(y) * 4 log2(e) = z * rcp(fma(exp2(-z), 1/4, 1/4)): the operand pre-scale of this kernel is 4 log2 e = 5.77
        //       (any scale works, it is undone in the epilogue) and costs nothing.  Lanes outside the plane get A = B = 0 (the
        //       zero padding applies to the ACTIVATED tensor), frames outside the tensor are skipped wave-uniformly.
        //   split: hi = f16(v) is one v_cvt_pk_f16_f32 per two elements, lo = f16(v - hi) ONE v_fma_mix per element (f32 + f16 ->
        //       f16).  The un-normalised path carries no pre-scale (operands keep 22 bits down to |x| = 2^-3 and an absolute 2^-25
        //       below: residual streams are O(1); the direct kernels' 2^4 bought 2^-29 at one more VALU op per element).  No clamp: |s V| > 65504 becomes inf and the output
        //       NaN / inf -- loud, never a silently clamped product (range check: dpc_unet3d_set_range_check).
        auto finish_item = [&](f32x4 (&d)[HFI], unsigned fok, int inflag, const f32x4& Ac, const f32x4& Bc, int dst0) {
            if (CONV_DBG_BUILD && (p.dbg & 64)) {             // attribution only: raw bits -> LDS (see the header), one v_and per dword
                unsigned char* q0 = halo + dst0;
                unsigned char* q1 = halo + (dst0 ^ 32);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f32x4 v = d[k % HFI];
                    uint2 p1, p2;
                    p1.x = __builtin_bit_cast(unsigned, v.x) & 0x3bff3bffu; p1.y = __builtin_bit_cast(unsigned, v.y) & 0x3bff3bffu;
                    p2.x = __builtin_bit_cast(unsigned, v.z) & 0x3bff3bffu; p2.y = __builtin_bit_cast(unsigned, v.w) & 0x3bff3bffu;
                    *reinterpret_cast<uint2*>(q0 + k * 6400) = p1;
                    *reinterpret_cast<uint2*>(q1 + k * 6400) = p2;
                }
                return;
            }
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
                            d[fi][e] = z * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-z), 0.25f, 0.25f));
                        }
                        if (fi & 1) __builtin_amdgcn_sched_barrier(0);
                    }
            }
            // plane 1 of a point is its plane-0 address ^ 32; frame offsets are multiples of 64, so both planes take them as
            // ds_write immediates on top of two base registers
            unsigned char* q0 = halo + dst0;
            unsigned char* q1 = halo + (dst0 ^ 32);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                // (element-wise on purpose: packed f32 VALU ops cost the co-resident MFMA wave more than the two scalar ops they
                // replace -- MI355X_MICROARCH.md, "price of one filler beside MFMAs"; the file is built with -fno-slp-vectorize)
                f32x4 v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = d[2 * pr][e], d1 = d[2 * pr + 1][e], d2 = d[2 * pr + 2][e], d3 = d[2 * pr + 3][e];
                    v[0][e] = d0 - d2; v[1][e] = d1 + d2; v[2][e] = d2 - d1; v[3][e] = d1 - d3;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint2 p1, p2;
                    if constexpr (GN) {       // the fused activation already carries its scale (4 log2 e): plain convert + remainder
                        p1.x = cvt_pk_f16(v[k].x, v[k].y);
                        p1.y = cvt_pk_f16(v[k].z, v[k].w);
                        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "=&v"(p2.x) : "v"(v[k].x), "v"(v[k].y), "v"(p1.x));
                        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "=&v"(p2.y) : "v"(v[k].z), "v"(v[k].w), "v"(p1.y));
                    } else {                  // plain input: h1 = fp16(8 x), h2 = fp16(8 x - h1), the scale folded into v_fma_mix (SAW)
                        const float sc = SAW;         // (8.0 is not an inline constant: one SGPR operand)
                        asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
                            : "=&v"(p1.x) : "v"(v[k].x), "v"(v[k].y), "s"(sc));
                        asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
                            : "=&v"(p1.y) : "v"(v[k].z), "v"(v[k].w), "s"(sc));
                        asm("v_fma_mixlo_f16 %0, %1, %4, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %4, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "=&v"(p2.x) : "v"(v[k].x), "v"(v[k].y), "v"(p1.x), "s"(sc));
                        asm("v_fma_mixlo_f16 %0, %1, %4, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, %4, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                            : "=&v"(p2.y) : "v"(v[k].z), "v"(v[k].w), "v"(p1.y), "s"(sc));
                    }
                    // (perf attribution, DPC_ENABLE_CONV_DBG builds: 128 = the remainder plane of the activations is written as zeros, 256 = with
                    //  its five low mantissa bits cleared -- how much of the launch time is the data-dependent power draw of the small-term MFMAs)
                    if (CONV_DBG_BUILD && (p.dbg & 128)) { p2.x = 0; p2.y = 0; }
                    if (CONV_DBG_BUILD && (p.dbg & 256)) { p2.x &= 0xffe0ffe0u; p2.y &= 0xffe0ffe0u; }
                    *reinterpret_cast<uint2*>(q0 + (pr * 4 + k) * 6400) = p1;
                    *reinterpret_cast<uint2*>(q1 + (pr * 4 + k) * 6400) = p2;
                    if (GN && (k & 1)) __builtin_amdgcn_sched_barrier(0);     // (register budget of the fused-activation variant)
                }
                __builtin_amdgcn_sched_barrier(0);        // (one frame pair at a time: keeps the loader inside the register budget)
            }
        };
        auto finish = [&](int kc, unsigned fok, int in0, int in1, f32x4 (&d)[2][HFI], const f32x4 (&cf)[2], int boff) {
            if (p.dbg & 32) return;
            const bool cok = kc * KC + hslot < K;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 A = (GN && cok) ? cf[0] : zero, B = (GN && cok) ? cf[1] : zero;
            finish_item(d[0], fok, in0, A, B, hdst[0] + boff);
            if (two) finish_item(d[1], fok, in1, A, B, hdst[1] + boff);
        };
        // ---- output transform + bias + GroupNorm partial sums + stores of a finished tile (see the MFMA waves' epilogue).
        // Loader wave w takes output-frame parity w & 1 and channel half w >> 1 of both frame pairs; lane = point (lane_hw(l31) of
        // the 4-row slab q), registers = channels 8g + 4hh + e -- the accumulator layout of the MFMA waves.
        const int lwv = wave - 4, par = lwv & 1, ntr = lwv >> 1;
        // Exchange layout (written by the MFMA waves): component k, channel half nt, point P = h * 8 + w of the 8 x 8 plane tile is a
        // 128-byte row of 32 channels at ((k * 2 + nt) * 64 + P) * 128; its eight 16-byte channel quads sit at slot (quad ^ (P & 7)).
        // A loader lane takes column w = lane >> 3 and quad j = lane & 7: eight lanes read one full row (conflict-free), and a
        // store instruction writes 8 points x 128 contiguous bytes -- full cache lines (the accumulator layout, lane = point with 16-byte
        // pieces of 32 rows per store, left the stores issue-bound at ~300 cycles each: tools/conv_stamps_w.py).
        f32x4 pv1[8];                                     // finished rows of the previous tile's second frame pair, not yet stored (see drain)
        float* pend_base = nullptr;
        bool pend_ok = false;
        const int pend_sh = kchunks >= 8 ? 0 : kchunks >= 4 ? 1 : 2;      // 8 stores over min(8, kchunks) steps
        auto epilogue = [&](int j, const unsigned char* xch) {
            int n0, w0, h0, f0, b;
            decode(j, n0, w0, h0, f0, b);
            const int col = lane >> 3, quad = lane & 7;
            const int nbase = n0 + ntr * 32 + 4 * quad;
            const long long tile = ((long long)(f0 / TFO) * nth + h0 / 8) * ntw + w0 / 8;
            const unsigned char* xl = xch + (ntr * 64 + col) * 128 + ((quad ^ col) << 4);
            // combine the three components of this wave's output frame (between the two barriers that fence the buffer): row i = h
            auto gather = [&](f32x4 (&o)[8]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    auto rd = [&](int k) { return *reinterpret_cast<const f32x4*>(xl + k * 16384 + i * 1024); };
                    const f32x4 m1 = rd(1), m2 = rd(2);
                    const f32x4 m03 = rd(par ? 3 : 0);
                    o[i] = par ? (m1 - m2) - m03 : (m03 + m1) + m2;
                    if (i & 1) __builtin_amdgcn_sched_barrier(0);      // (at most 6 reads in flight: registers)
                }
            };
            // de-scale + bias, GroupNorm partial sums of the OUTPUT (this lane: 4 channels x 8 rows of one column), dwordx4 stores
            // (asm: the stores must not enter hipcc's vmcnt bookkeeping of this path -- see landed(); they retire in order with the
            // loads, so a later vmcnt(12 | 14) also waits for them), then the sums over the 8 columns (lane bits 3-5) and one
            // 32-byte row of GroupNorm partial-sum entry 2 pr + par of the tile per quad.
            auto emit = [&](int pr, f32x4 (&o)[8]) {
                const int f = f0 + 2 * pr + par;
                float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
                const bool now = !pr || (p.dbg & 512);             // (512: A/B switch -- no deferred stores)
                if (pr) pend_ok = f < p.F && !now;
                if (f < p.F) {
                    const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nbase) : f32x4{0.f, 0.f, 0.f, 0.f};
                    float* base = p.out + ((((long long)b * p.F + f) * p.H + h0) * p.W + w0 + col) * p.N + nbase;
                    if (pr) pend_base = base;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = o[i][e] * descale + bv[e];
                            o[i][e] = v;
                            gs[e] += v;
                            gq[e] += v * v;
                        }
                        // (s_nop: a VALU write to the data registers of a >64-bit store needs 2 wait states, and the hazard
                        // recogniser does not look into inline asm)
                        if (now) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(base + (long long)i * p.W * p.N), "v"(o[i]) : "memory");
                    }
                }
                if (p.gn_part) {
#pragma unroll
                    for (int m = 8; m <= 32; m <<= 1)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gs[e] += __shfl_xor(gs[e], m, 64);
                            gq[e] += __shfl_xor(gq[e], m, 64);
                        }
                    if (lane < 8) {
                        float* gdst = p.gn_part + ((((long long)b * ((long long)ntf * nth * ntw) + tile) * 4 + 2 * pr + par) * p.N + nbase) * 2;
                        const f32x4 t0 = {gs[0], gq[0], gs[1], gq[1]}, t1 = {gs[2], gq[2], gs[3], gq[3]};
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(gdst), "v"(t0) : "memory");
                        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(gdst + 4), "v"(t1) : "memory");
                    }
                }
            };
            f32x4 o0[8];
            wg_barrier();                                  // E1: pair 0 is in LDS
            gather(o0);
            lds_done_barrier();                            // E2: the MFMA waves may overwrite it with pair 1
            wg_barrier();                                  // E3
            gather(pv1);
            lds_done_barrier();                            // E4: the buffer returns to the halo pipeline; the MFMA waves go on
            stamp();
            emit(0, o0);
            stamp();
            emit(1, pv1);
        };
        // Half of a tile's row stores (its second frame pair: 8 per lane) are NOT issued in the epilogue: every CU finishes its tile at
        // the same moment and 32 MB of simultaneous stores run at the HBM write limit (~500 cycles per store instruction, 9 k cycles
        // per tile in front of the next tile's first chunk).  They drain one or two per step behind the NEXT tile's halo production
        // (all 16 do not fit the fused-activation variant's registers).
        auto drain = [&](int slot) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if ((t >> pend_sh) == slot && pend_ok)
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(pend_base + (long long)t * p.W * p.N), "v"(pv1[t]) : "memory");
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
        stamp();
        // iteration s: X = raw data of step s + 1 (requested one iteration ago); request step s + 2 into Y, finish X.  A tile is an
        // EVEN number of steps (conv3w_supported: K % 32 == 0), so the two register sets alternate without copies, the epilogue
        // exists once in the code and the buffer a tile leaves behind is always buffer 1.
        auto body = [&](long long s, f32x4 (&X)[2][HFI], StepState& sx, f32x4 (&Y)[2][HFI], StepState& sy, int boff, int slot) {
            if (s + 2 < nsteps) { advance(); request(Y, sy); }
            stamp();
            if (s + 1 < nsteps) landed(X, sx.cf, s + 2 < nsteps);
            stamp();
            if (s + 1 < nsteps) finish(sx.kc, sx.fok, sx.in0, sx.in1, X, sx.cf, boff);   // the MFMA waves left that buffer at the previous barrier
            drain(slot);
            stamp();
            lds_done_barrier();
            stamp();
        };
        long long s = 0;
        for (int j = 0; j < ntiles; ++j) {
            for (int kc = 0; kc < kchunks; kc += 2) {
                body(s, rb, sb, ra, sa, HBS, kc);
                body(s + 1, ra, sa, rb, sb, 0, kc + 1);
                s += 2;
            }
            if (!(p.dbg & 8)) epilogue(j, halo + HBS);
            stamp();
        }
        for (int slot = 0; slot < 8; ++slot) drain(slot);       // the last tile
        __builtin_amdgcn_s_waitcnt(0);
        stamp_out();
        return;
    }

    // =========================================================================================== MFMA waves (wave = component k)
    int lh, lw;
    lane_hw(l31, lh, lw);
    int aaddr[9];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) aaddr[dh * 3 + dw] = slot0(wave, lh + dh, lw + dw, hh);

    f32x16 acc[MT][NT];
    f16x8 w[3][NT][2];
    f16x8 a[MT][2];

    // weight stream of component k: taps 0..8 of chunk 0, ..., of the last chunk, then the next tile; one wave-uniform pointer
    const long long wstride = (long long)p.Npad * WROW, wtap = wstride * kchunks;
    const unsigned char* wroot = reinterpret_cast<const unsigned char*>(p.wpw) + (long long)((p.dbg & 4) ? 0 : wave) * NTAPS * wtap;
    const int wlo = l31 * WROW + hh * 16;
    const unsigned char* wlane = wroot;
    const unsigned char* wnext = wroot;
    int wtap_i = 0, wkc_i = 0, wtile = 0;
    auto tile_n0 = [&](int j) {
        int n0, w0, h0, f0, b;
        decode(j < ntiles ? j : ntiles - 1, n0, w0, h0, f0, b);
        return n0;
    };
    auto ldw = [&](f16x8 (&dst)[NT][2]) {
        const unsigned char* src = wnext + wlo;
        if (++wtap_i == NTAPS) {
            wtap_i = 0;
            if (++wkc_i == kchunks) { wkc_i = 0; ++wtile; wlane = wroot + (long long)tile_n0(wtile) * WROW; }
            wnext = wlane + wkc_i * wstride;
        } else {
            wnext += wtap;
        }
        if (CONV_DBG_BUILD && (p.dbg & 1024)) {           // attribution only (r06; results INVALID): every weight fragment is requested TWICE,
                                                          // the second time from the neighbour component's stream as an LDS-DMA into the unused
                                                          // tail of halo buffer 0 (no destination registers) -- the L2 -> L1 weight traffic per
                                                          // MFMA of an F(4,3) form whose register budget halves the fragment reuse
            const unsigned char* s2 = src + (long long)((((wave + 1) & 3) - wave) * NTAPS) * wtap;
            const int m0v = 52 * 1024 + wave * 1024;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(s2 + nt * 32 * WROW + pl * 32), "s"(m0v) : "memory", "m0");
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) dst[nt][pl] = *reinterpret_cast<const f16x8*>(src + nt * 32 * WROW + pl * 32);
    };
    wlane = wroot + (long long)tile_n0(0) * WROW;
    wnext = wlane;
    ldw(w[0]);
    ldw(w[1]);

    int boff = 0;
    // A fragments of slab pair `pr` (= frame pair pr: transformed frame 4 pr + k, rows 0-3 and 4-7) for tap (dh, dw)
    auto lda_pair = [&](int tap, int pr) {
        const int a0 = (aaddr[tap] ^ boff);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int off = pr * 4 * 6400 + 4 * q * 640;
            a[2 * pr + q][0] = *reinterpret_cast<const f16x8*>(halo + a0 + off);
            a[2 * pr + q][1] = *reinterpret_cast<const f16x8*>(halo + (a0 ^ 32) + off);
        }
    };

    wg_barrier();                                         // step 0 is in buffer 0
    stamp();
    for (int j = 0; j < ntiles; ++j) {
        int n0, w0, h0, f0, b;
        decode(j, n0, w0, h0, f0, b);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        for (int kc = 0; kc < kchunks; ++kc) {
            lda_pair(0, 0);                               // (set 1 follows four MFMAs into the first tap, see tap_body)
            auto tap_body = [&](int tap) {
                ldw(w[(tap + 2) % 3]);                    // two taps ahead (the ring runs on across chunks and tiles)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};     // small terms first; PA: activation plane, PB: weight plane
                // Fragment re-loads (r03): a set is re-loaded FOUR MFMAs after the last MFMA that reads it was issued, and eight MFMAs
                // before its next reader -- set 1 (pair 1 of this tap) inside group (tap, 0), set 0 (pair 0 of the next tap) inside
                // group (tap, 1).  The earlier order issued the ds_read right behind the group that had just read the set and relied
                // on "an MFMA starts within its 32-cycle slot, the LDS return needs >= 64 cycles": true only while no foreign wave
                // issues MFMAs on the SIMD -- two processes sharing the GPU produced one wrong fragment element now and then
                // (third hazard; tools/rank_stress.py).  Same MFMA order: results are bit-identical.
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
                    for (int term = 0; term < 3; ++term) {
                        if (term == 1) {
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                            if (pr == 0) lda_pair(tap, 1);
                            else if (tap < NTAPS - 1) lda_pair(tap + 1, 0);
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[2 * pr + q][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    w[tap % 3][nt][PB[term]], a[2 * pr + q][PA[term]], acc[2 * pr + q][nt], 0, 0, 0);
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) tap_body(tap);
            // MFMA B-operand guard (see igemm6.hip): nothing may overwrite the activation fragments while the last MFMA reads them
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (j < 2) stamp();
            wg_barrier();                                  // next chunk's buffer is complete; this one may be overwritten
            if (j < 2) stamp();
            boff ^= HBS;
        }

        // ---- epilogue.  acc[2 pr + q][nt][4g + e] = component k of pair pr, point (rows 4q.., lane_hw(l31)), channel nt*32 + 8g + 4hh + e.
        // The output transform crosses waves AND is done by the loader waves (they idle here anyway, and a tile's 128 KB of output
        // stores -- issue-bound while every CU bursts at once -- then drain beside the next tile's MFMAs instead of in front of
        // them): per frame pair the four MFMA waves park their component in the buffer of the last chunk (boff ^ HBS after the
        // toggle; 64 KB, layout at the loader's epilogue), the loader waves read and combine it between two barriers.
        unsigned char* xch = halo + (boff ^ HBS);
        if (p.dbg & 8) continue;
        // (row of point P = (4 q + lh) * 8 + lw; this lane's quads 2 g + hh go to slot (2 g + hh) ^ (P & 7) = 2 g ^ (hh ^ lw))
        unsigned char* xw = xch + (wave * 128 + lh * 8 + lw) * 128;
        const int tsw = hh ^ lw;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[2 * pr + q][nt][4 * g], acc[2 * pr + q][nt][4 * g + 1], acc[2 * pr + q][nt][4 * g + 2],
                                         acc[2 * pr + q][nt][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(xw + nt * 8192 + q * 4096 + ((tsw ^ (2 * g)) << 4)) = v;
                    }
            lds_done_barrier();                            // E1 / E3: pair pr is in LDS
            if (j < 2) stamp();
            wg_barrier();                                  // E2 / E4: the loader waves have read it
            if (j < 2) stamp();
        }
    }
    stamp_out();
}

static int conv3w_enabled() {
    static const int ok = debug_switch("DPC_CONV3W", 1);
    return ok;
}

// shape-only rule (never the batch): the Winograd form is taken for 3x3x3 convolutions whose planes tile by 8 x 8 and whose
// channel counts fit the 64-wide column tile; F % 4 != 0 runs partial frame tiles (F >= 16 only, as the direct kernels)
bool conv3w_shape_ok(int F, int H, int W, int N, int Npad) {
    return conv3w_enabled() && H % 8 == 0 && W % 8 == 0 && N % 64 == 0 && N == Npad && (F % 4 == 0 || F >= 16);
}

bool conv3w_supported(const Conv3hParams& p) {
    // buffer addressing of the loader: one chunk = one source, per-sample tensors below the out-of-range marker (3 GB)
    const long long smax = (long long)p.F * p.H * p.W * std::max(p.C0, p.C1) * 4;
    return p.wpw && p.kd != 1 && conv3w_shape_ok(p.F, p.H, p.W, p.N, p.Npad) && p.C0 % 4 == 0 && p.C1 % 4 == 0 && p.act_scale == 0.f &&
           (p.C1 == 0 || p.C0 % 16 == 0) && (p.C0 + p.C1) % 32 == 0 && smax < 0xC0000000ll - 0x40000000ll;
}

long long conv3w_gn_entries(int F, int H, int W) { return (long long)((F + 3) / 4) * (H / 8) * (W / 8) * 4; }

int launch_conv3w(const Conv3hParams& p, hipStream_t s) {
    using namespace f3c;
    if (conv3w_f43_enabled()) return launch_conv3w4(p, s);      // F(4,3) form (conv3w4.hip): the pack in p.wpw is then its 6-component one
    const long long tiles = (long long)p.B * ((p.F + 3) / 4) * (p.H / 8) * (p.W / 8);
    const long long nwg = tiles * (p.Npad / 64);
    DPC_REQUIRE(nwg < (1ll << 31), "conv3w: too many tiles");
    static int ncu = 0;
    static DeviceOnce once;
    if (!once) {
        int dev = 0;
        hipDeviceProp_t prop;
        DPC_HIP(hipGetDevice(&dev));
        DPC_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = std::max(8, prop.multiProcessorCount / 8 * 8);
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w_kernel<false, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w_kernel<true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        once = true;
    }
    Conv3hParams pd = p;
    pd.total_wg = (int)nwg;
    const unsigned grid = (unsigned)std::min<long long>(nwg, cu_budget(ncu));
    const bool wide = p.Npad % 128 == 0 && p.N > 64;       // profile class only (launch_conv3f3's ProfScope uses the same rule)
    if (p.in_coef) {
        if (wide) hipLaunchKernelGGL((conv3w_kernel<true, 128>), dim3(grid), dim3(512), 2 * HBS, s, pd);
        else hipLaunchKernelGGL((conv3w_kernel<true, 64>), dim3(grid), dim3(512), 2 * HBS, s, pd);
    } else {
        if (wide) hipLaunchKernelGGL((conv3w_kernel<false, 128>), dim3(grid), dim3(512), 2 * HBS, s, pd);
        else hipLaunchKernelGGL((conv3w_kernel<false, 64>), dim3(grid), dim3(512), 2 * HBS, s, pd);
    }
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ---- weight transform + pre-split: reference [N][K][3][3][3] fp32 -> [4 k][9 taps][kchunks][Npad][2 planes][16] fp16 (x 2^12)
__global__ void pack_weights_w3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int N, int Npad, int K,
                                       int kchunks, int* __restrict__ ovf) {
    const long long total = (long long)36 * kchunks * Npad * 16;
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
            const float u = k == 0 ? g0 : k == 3 ? g2 : k == 1 ? ((g0 + g2) + g1) * 0.5f : ((g0 + g2) - g1) * 0.5f;
            v = u * f3c::SW;
            if (!(fabsf(v) <= 65504.f)) atomicOr(ovf, 1);
            v = f3c::sat16(v);
        }
        const unsigned p1 = f3c::cvt_pk_f16(v, 0.f) & 0xffffu;
        const float h1 = (float)__builtin_bit_cast(f3c::f16x2, p1).x;
        const unsigned p2 = f3c::cvt_pk_f16(v - h1, 0.f) & 0xffffu;
        unsigned short* dst = wp + (((long long)kt * kchunks + kc) * Npad + n) * 32 + kk;
        dst[0] = (unsigned short)p1;
        dst[16] = (unsigned short)p2;
    }
}

size_t conv3w_packed_bytes(int Npad, int K) {
    return conv3w_f43_enabled() ? conv3w4_packed_bytes(Npad, K) : (size_t)36 * ((K + 15) / 16) * Npad * 64;
}

// (the pack follows the form launch_conv3w will run: one process-wide switch decides both)
int launch_pack_weights_w3(const float* w, void* wp, int N, int Npad, int K, hipStream_t s) {
    if (conv3w_f43_enabled()) return launch_pack_weights_w4(w, wp, N, Npad, K, s);
    const int kchunks = (K + 15) / 16;
    const long long total = (long long)36 * kchunks * Npad * 16;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_w3_kernel, dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), N, Npad, K,
                       kchunks, f16x3_weight_overflow_flag());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
