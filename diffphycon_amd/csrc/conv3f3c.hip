// Conv3d 3x3x3 / stride 1 / pad 1, channels-last, f16x3 arithmetic (conv3f3.hip) -- loader-wave / persistent form.
//
// r02 timeline of conv3f3b (tools/conv_stamps.py, 64 -> 64 @ 8x32x64x64, shader cycles per workgroup): 136 k, of which the
// MFMA stream needs 83 k.  The rest: tile prologue 13.5 k, per channel chunk ~2.3 k halo-load issue + 1.5 k GroupNorm/SiLU/
// split + 1.5 k hand-over + ~3 k of tap stalls behind the halo loads (vmcnt retires in order, so every weight fragment issued
// after a halo load waits for it), epilogue 7.5 k (128 dword stores per lane).  All of it is work of the SAME four waves that
// feed the matrix cores.  Here the roles are split:
//   * waves 0-3 ("MFMA waves", one per SIMD) only read A fragments from LDS, stream weight fragments from L2 and issue MFMAs;
//   * waves 4-7 ("loader waves", the second wave of each SIMD) load the next 16-channel halo chunk from HBM/L2, apply the
//     producer's GroupNorm + (scale, shift) + SiLU, pre-scale, split into the two fp16 planes and write it to LDS;
//   * the halo is DOUBLE-buffered: 64 B per point (2 planes x 32 B) with no padding -- bank conflicts are removed by an XOR
//     swizzle of the four 16-byte slots of a point instead (slot ^= ((w >> 2) & 1) | (((h >> 1) & 1) << 1): the 16 lanes that
//     ds_read_b128 services together then cover all 16 four-bank groups for every tap offset) -- so 1000 points x 64 B x 2
//     buffers = 128 KB fit the 160 KB LDS, and ONE s_barrier per chunk is the whole hand-over;
//   * the grid is persistent (one workgroup per CU walking its XCD's share of the tiles) and the (tile, chunk) sequence is
//     one continuous stream: the first chunk of the next tile is loaded during the last chunk of the current one;
//   * accumulators are kept TRANSPOSED (MFMA A operand = weights, B operand = activations: lane = point, registers =
//     channels), so a lane owns 4 consecutive channels per register quad and the epilogue is 32 global_store_dwordx4 per
//     lane instead of 128 dword stores; the GroupNorm partial sums are reduced across lanes with a 16-shuffle transpose tree.
// Every wave stays inside 256 registers (two waves per SIMD): the A fragments are a single rolling set (the next tap's
// fragments of a slab pair are re-loaded right after that pair's MFMAs were issued, behind the other pair's 12 MFMAs).
// Arithmetic, weight pack ([tap][chunk][n][2 planes][16] fp16, launch_pack_weights_f3) and GroupNorm partial-sum layout are
// those of conv3f3b_kernel; results are bit-identical to it except for the summation order inside the GroupNorm partials.
// Reference op: nn.Conv3d(dim, dim_out, (3,3,3), padding=(1,1,1)) in Block (video_diffusion_pytorch_conv3d.py:189-204).
// r05, KD = 1 (the 2-D nets' 3x3 convolutions, one IMAGE per frame: model/burgers_1d/unet.py:134-191 Block / ResnetBlock; the jellyfish
// surrogates): GroupNorm is per image there, so the fusion of the 3-D path needs per-frame bookkeeping --
//   * statistics epilogue: one partial-sum entry per (image, 8 x 8 plane tile) = [F][nth * ntw][N][2] (a wave reduces the two slabs of
//     each of its two frames separately: 2 x the shuffles of the 3-D form);
//   * fused-input loader: a loader thread's items are grouped by frame (item i = frame i >> 1, point slot (ltid >> 2) + 64 (i & 1)), so
//     the folded coefficients (A, B) log2(e) of launch_gn_finalize_fused's second table are fetched once per (frame, chunk) and the
//     activation is z = fma(x, A, B), SiLU(y) sa = z rcp(fma(exp2(-z), c, c)), c = log2(e) / sa -- the operand pre-scale is free, as in
//     conv3w.hip's loader.  Zero padding applies to the ACTIVATED tensor: out-of-image points stay 0.
#include "common.h"
#include "f3c.h"

namespace dpc {


// BN = 64 : 4 x 1 MFMA waves over an 8 x 8 x 8 output tile (halo 10 x 10 x 10);  BN = 128: 2 x 2 waves over 4 x 8 x 8 (6 x 10 x 10).
// Wave (wm, wn) owns frames 2 wm, 2 wm + 1 (slab mt = frame 2 wm + (mt >> 1), rows 4 (mt & 1) .. +3) and BN / WN channels.
// KD = 3: the 3x3x3 convolution.  KD = 1: a (1,3,3) convolution (the 2-D nets' 3x3 convs, images on the frame axis: 9 taps, no
// frame halo) -- few taps and small K make the per-tile prologue / epilogue of conv3f3b the larger part of its time there.
template <int BN, int KD>
__global__ __launch_bounds__(512, 2) void conv3f3c_kernel(Conv3hParams p) {
    fp16_ovfl_enable();                                 // (common.h: operand conversions saturate in hardware)
    using namespace f3c;
    constexpr int WM = BN == 64 ? 4 : 2, WN = 4 / WM, MT = 4, NT = 2;
    constexpr int TF = 2 * WM, HF = TF + KD - 1, NTAPS = 9 * KD, FPAD = KD / 2;
    constexpr int NLOG = HF * 100;                      // halo points: 1000 / 600
    constexpr int HLOADS = (NLOG * 4 + 255) / 256;      // 16 / 10 quads per loader thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_f3c[];
    unsigned char* halo = smem_f3c;                     // two buffers at 0 and HBS

    // run-time activation scale for the 2-D form only (surr.hip's calibrated backward convs); a compile-time constant in the
    // register-tight 3-D kernel
    const float sa = (KD == 1 && p.act_scale != 0.f) ? p.act_scale : SA;
    const float descale = 1.0f / (sa * SW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int ntn = p.Npad / BN;
    const int ntf = (p.F + TF - 1) / TF, nth = p.H / 8, ntw = p.W / 8;
    const int K = p.C0 + p.C1, kchunks = p.kchunks;
    const int nb = p.total_wg;                          // tile workgroups of the launch; this workgroup walks wg, wg + grid, ...
    const int ntiles = nb > (int)blockIdx.x ? (nb - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const long long nsteps = (long long)ntiles * kchunks;
    // XCD-aware order: consecutive tile indices (shared halo planes, same weights) stay on the XCD whose L2 already holds them
    auto decode = [&](int j, int& n0, int& w0, int& h0, int& f0, int& b) {
        int bid = (int)blockIdx.x + j * (int)gridDim.x;
        {
            const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        }
        n0 = (bid % ntn) * BN;
        int t = bid / ntn;
        w0 = (t % ntw) * 8; t /= ntw;
        h0 = (t % nth) * 8; t /= nth;
        f0 = (t % ntf) * TF;
        b = t / ntf;
    };
    if (nsteps == 0) return;

    if (wave >= 4) {
        // ======================================================================================= loader waves
        const int ltid = tid - 256;
        // item -> (frame, point, channel quad).  KD = 3: items in point order.  KD = 1: grouped by frame (see the header): every item of a
        // thread knows its frame at compile time, which is what the per-image GroupNorm coefficients of the fused input need.
        constexpr int NITEM = KD == 1 ? HF * 2 : HLOADS;
        auto item_point = [&](int i, int& pf, int& ph, int& pw, int& quad, bool& valid) {
            if (KD == 1) {
                const int slot = (ltid >> 2) + 64 * (i & 1);
                pf = i >> 1; ph = slot / 10; pw = slot % 10; quad = ltid & 3;
                valid = slot < 100;
                if (!valid) { ph = 0; pw = 0; }
            } else {
                const int q = ltid + 256 * i, pt = q >> 2;
                pf = pt / 100; ph = (pt / 10) % 10; pw = pt % 10; quad = q & 3;
                valid = pt < NLOG;
            }
        };
        int hdst[NITEM];
        unsigned hvalid = 0;
#pragma unroll
        for (int i = 0; i < NITEM; ++i) {
            int pf, ph, pw, quad;
            bool valid;
            item_point(i, pf, ph, pw, quad, valid);
            hdst[i] = slot0(pf, ph, pw, quad >> 1) + (quad & 1) * 8;
            if (valid) hvalid |= 1u << i;
        }
        const int hslot = (ltid & 3) * 4;
        unsigned hokm = 0;
        int hpt[NITEM];
        const float* xb0 = nullptr;
        const float* xb1 = nullptr;
        int b_cur = 0, f0_cur = 0;
        auto setup_tile = [&](int j) {
            int n0, w0, h0, f0, b;
            decode(j, n0, w0, h0, f0, b);
            b_cur = b;
            f0_cur = f0;
            xb0 = p.a0 + (long long)b * p.F * p.H * p.W * p.C0;
            xb1 = p.a1 ? p.a1 + (long long)b * p.F * p.H * p.W * p.C1 : nullptr;
            hokm = 0;
#pragma unroll
            for (int i = 0; i < NITEM; ++i) {
                int pf, ph, pw, quad;
                bool valid;
                item_point(i, pf, ph, pw, quad, valid);
                const int f = f0 - FPAD + pf, h = h0 - 1 + ph, w = w0 - 1 + pw;
                if (valid && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) hokm |= 1u << i;
                hpt[i] = (f * p.H + h) * p.W + w;
            }
        };
        // load -> producer's GroupNorm + (scale + 1, shift) + SiLU when fused (Block.forward, ...conv3d.py:196-204; the zero
        // padding applies to the ACTIVATED tensor, so out-of-range points stay 0) -> pre-scale, split -> LDS buffer `boff`
        auto produce = [&](int kc, int boff) {
            const int c = kc * KC + hslot;
            const float* src;
            int cs, cc;
            if (c < p.C0) { src = xb0; cs = p.C0; cc = c; }
            else { src = xb1; cs = p.C1; cc = c - p.C0; }
            const bool cok = c < K;
            f32x4 hreg[NITEM];
#pragma unroll
            for (int i = 0; i < NITEM; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (cok && ((hokm >> i) & 1)) v = *reinterpret_cast<const f32x4*>(src + (long long)hpt[i] * cs + cc);
                hreg[i] = v;
            }
            bool scaled = false;                          // true: hreg already carries the operand pre-scale sa
            if (KD == 1) {
                if (p.in_coef && cok && !(CONV_DBG_BUILD && (p.dbg & 64))) {
                    // per-image folded coefficients (second table of launch_gn_finalize_fused, "batch" = the F images of this launch)
                    const float cl = 1.4426950408889634f / sa;
                    const f32x4* tab = reinterpret_cast<const f32x4*>(p.in_coef + (long long)p.F * K * 5);
#pragma unroll
                    for (int pf = 0; pf < HF; ++pf) {
                        const int img = min(f0_cur + pf, p.F - 1);
                        const f32x4* cf = tab + ((long long)img * (K >> 2) + (c >> 2)) * 2;
                        const f32x4 A = cf[0], Bc = cf[1];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const int i = 2 * pf + u;
                            const bool in = (hokm >> i) & 1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float z = __builtin_fmaf(hreg[i][e], A[e], Bc[e]);
                                const float y = z * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-z), cl, cl));
                                hreg[i][e] = in ? y : 0.f;
                            }
                        }
                    }
                    scaled = true;
                }
            } else if (p.in_coef && cok) {
                const f32x4* cf = reinterpret_cast<const f32x4*>(p.in_coef) + ((long long)b_cur * (K >> 2) + (c >> 2)) * 5;
                const f32x4 mu = cf[0], ga = cf[1], be = cf[2], sc = cf[3], sh = cf[4];
#pragma unroll
                for (int i = 0; i < NITEM; ++i) {
                    if ((hokm >> i) & 1) {
                        f32x4 y = (hreg[i] - mu) * ga + be;
                        y = y * sc + sh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.0f + expf(-y[e]));
                        hreg[i] = y;
                    }
                }
            }
            if (CONV_DBG_BUILD && (p.dbg & 64)) {
                // perf attribution only (DPC_ENABLE_CONV_DBG builds, results INVALID): the loader keeps its loads and LDS writes but does no
                // activation / pre-scale / split -- the raw bits, masked to finite fp16 patterns, go to LDS: what a loader that only COPIES
                // pre-split planes would cost, with live operands for the MFMA stream (as conv3w.hip's bit 64)
#pragma unroll
                for (int i = 0; i < NITEM; ++i) {
                    if ((hvalid >> i) & 1) {
                        uint2 p1, p2;
                        p1.x = __builtin_bit_cast(unsigned, hreg[i].x) & 0x3bff3bffu; p1.y = __builtin_bit_cast(unsigned, hreg[i].y) & 0x3bff3bffu;
                        p2.x = __builtin_bit_cast(unsigned, hreg[i].z) & 0x3bff3bffu; p2.y = __builtin_bit_cast(unsigned, hreg[i].w) & 0x3bff3bffu;
                        const int d = hdst[i] + boff;
                        *reinterpret_cast<uint2*>(halo + d) = p1;
                        *reinterpret_cast<uint2*>(halo + (d ^ 32)) = p2;
                    }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < NITEM; ++i) {
                if ((hvalid >> i) & 1) {
                    uint2 p1, p2;
                    split2(scaled ? hreg[i] : hreg[i] * sa, p1, p2);
                    const int d = hdst[i] + boff;
                    *reinterpret_cast<uint2*>(halo + d) = p1;
                    *reinterpret_cast<uint2*>(halo + (d ^ 32)) = p2;
                }
            }
        };
        int j = 0, kc = 0;
        setup_tile(0);
        produce(0, 0);
        lds_done_barrier();                               // buffer 0 holds step 0
        for (long long s = 0; s < nsteps; ++s) {
            if (++kc == kchunks) { kc = 0; ++j; }
            if (s + 1 < nsteps) {
                if (kc == 0) setup_tile(j);
                produce(kc, ((int)(s + 1) & 1) * HBS);    // the MFMA waves left that buffer at the previous barrier
            }
            lds_done_barrier();
        }
        return;
    }

    // =========================================================================================== MFMA waves
    const int wm = wave / WN, wn = wave % WN;
    int lh, lw;
    lane_hw(l31, lh, lw);
    // A-fragment addresses: one per (dh, dw) tap offset (the swizzle depends on the absolute halo row / column), plane 0;
    // frame and slab offsets are immediates, plane 1 = address ^ 32, the other buffer = address ^ HBS
    int aaddr[9];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) aaddr[dh * 3 + dw] = slot0(wm * 2, lh + dh, lw + dw, hh);

    f32x16 acc[MT][NT];
    f16x8 w[3][NT][2];
    f16x8 a[MT][2];

    // the weight stream is walked with ONE running pointer: taps 0..26 of chunk 0, ..., of the last chunk, then the next tile.
    // The pointer is wave-uniform (SGPRs); the lane part is a single 32-bit offset.
    const long long wstride = (long long)p.Npad * WROW, wtap = wstride * kchunks;
    const unsigned char* wroot = reinterpret_cast<const unsigned char*>(p.wp) + (long long)wn * (BN / WN) * WROW;
    const int wlo = l31 * WROW + hh * 16;
    const unsigned char* wlane = wroot;                  // + n0 * WROW of the tile whose weights are being streamed
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
    // A fragments of slab pair `pr` (slabs 2 pr, 2 pr + 1) for tap (df, dh, dw)
    auto lda_pair = [&](int tap, int pr) {
        const int df = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
        const int a0 = (aaddr[dh * 3 + dw] ^ boff);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int mt = 2 * pr + q;
            const int off = (df + (mt >> 1)) * 6400 + (4 * (mt & 1)) * 640;
            a[mt][0] = *reinterpret_cast<const f16x8*>(halo + a0 + off);
            a[mt][1] = *reinterpret_cast<const f16x8*>(halo + (a0 ^ 32) + off);
        }
    };

#ifdef DPC_CONV_STAMPS
    // per MFMA wave: [0] kernel start -> first barrier passed, then for tiles 0 and 1: per chunk (taps, barrier wait), epilogue;
    // [28..31] raw start / end clocks.  tools/conv_stamps.py
    unsigned long long tst[30];
    int nst = 0;
    auto stamp = [&]() { if (nst < 30) tst[nst++] = __builtin_amdgcn_s_memtime(); };
#else
    auto stamp = [&]() {};
#endif
    stamp();
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

        // ---- epilogue: lane = point (slab mt, lane_hw(l31)), registers 4g..4g+3 = channels nt*32 + 8g + 4hh .. +3
        const int nbase = n0 + wn * (BN / WN) + 4 * hh;
        const long long tile = ((long long)(f0 / TF) * nth + h0 / 8) * ntw + w0 / 8;
        // GroupNorm partial sums of the OUTPUT: KD = 3 one entry per (tile, wave row) over the wave's 4 slabs (statistics per sample);
        // KD = 1 one entry per (image, plane tile): the wave's two frames are two images, NGRP = 2 groups of two slabs each
        constexpr int NGRP = KD == 1 ? 2 : 1, MPG = MT / NGRP;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 bv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bv[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nbase + nt * 32 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int grp = 0; grp < NGRP; ++grp) {
                float gs[16], gq[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { gs[r] = 0.f; gq[r] = 0.f; }
#pragma unroll
                for (int mi = 0; mi < MPG; ++mi) {
                    const int mt = grp * MPG + mi;
                    const int f = f0 + wm * 2 + (mt >> 1);
                    if (f >= p.F) continue;                  // partial frame tile
                    float* base = p.out + ((((long long)b * p.F + f) * p.H + h0 + 4 * (mt & 1) + lh) * p.W + w0 + lw) * p.N + nbase + nt * 32;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[mt][nt][4 * g + e] * descale + bv[g][e];
                            gs[4 * g + e] += v[e];
                            gq[4 * g + e] += v[e] * v[e];
                        }
                        if (!(CONV_DBG_BUILD && (p.dbg & 8))) *reinterpret_cast<f32x4*>(base + 8 * g) = v;      // (bit 8: perf attribution -- no output stores)
                    }
                }
                float* gdst = nullptr;
                if (p.gn_part) {
                    if (KD == 1) {
                        const int f = f0 + wm * 2 + grp;
                        if (f < p.F) gdst = p.gn_part + (((long long)f * (nth * ntw) + (h0 / 8) * ntw + w0 / 8) * p.N) * 2;
                    } else {
                        gdst = p.gn_part + (((long long)b * ((long long)ntf * nth * ntw) + tile) * WM + wm) * p.N * 2;
                    }
                }
                if (p.gn_part) {                         // (wave-uniform: every lane takes part in the shuffles)
                    // Transpose tree over the 32 lanes of a half-wave: each step halves the registers a lane still owns and adds its
                    // partner's copy of them (8 + 4 + 2 + 1 shuffles), a last plain exchange folds lanes 2k, 2k + 1; lane l31 then holds
                    // the total of register r = 8 b4 + 4 b3 + 2 b2 + b1 (b_i = bit i of l31).  Fixed order: deterministic.
                    float tot[2];
#pragma unroll
                    for (int which = 0; which < 2; ++which) {
                        float* x = which ? gq : gs;
#pragma unroll
                        for (int half = 8; half >= 1; half >>= 1) {
                            const bool up = (l31 & (half * 2)) != 0;          // lane bit 4, 3, 2, 1 for half = 8, 4, 2, 1
#pragma unroll
                            for (int i = 0; i < half; ++i) {
                                const float send = up ? x[i] : x[i + half];
                                const float keep = up ? x[i + half] : x[i];
                                x[i] = keep + __shfl_xor(send, half * 2, 64);
                            }
                        }
                        tot[which] = x[0] + __shfl_xor(x[0], 1, 64);
                    }
                    if (gdst && (l31 & 1) == 0) {
                        const int r = ((l31 >> 4) & 1) * 8 + ((l31 >> 3) & 1) * 4 + ((l31 >> 2) & 1) * 2 + ((l31 >> 1) & 1);
                        const int n = n0 + wn * (BN / WN) + nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        gdst[n * 2] = tot[0];
                        gdst[n * 2 + 1] = tot[1];
                    }
                }
                asm volatile("" ::: "memory");
            }
        }
        if (j < 2) stamp();
    }
#ifdef DPC_CONV_STAMPS
    {
        const unsigned long long tend = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            float* rec = p.out + ((long long)blockIdx.x * 4 + wave) * 32;
            for (int i = 1; i < nst && i < 28; ++i) rec[i - 1] = (float)(tst[i] - tst[i - 1]);
            unsigned* ru = reinterpret_cast<unsigned*>(rec);
            ru[28] = (unsigned)tst[0]; ru[29] = (unsigned)(tst[0] >> 32);
            ru[30] = (unsigned)tend; ru[31] = (unsigned)(tend >> 32);
        }
    }
#endif
}

bool conv3f3c_supported(const Conv3hParams& p) {
    static const int ok = debug_switch("DPC_CONV3F3C", 1);
    const bool wide = p.Npad % 128 == 0 && p.N > 64;
    const int tf = wide ? 4 : 8;
    // (kd == 1: any image count -- partial frame tiles -- so that the kernel choice, and the GroupNorm fusion that rests on it, depends on
    //  the layer shape only, never on the batch)
    return ok && p.H % 8 == 0 && p.W % 8 == 0 && p.N % 64 == 0 && p.N == p.Npad && (p.kd == 1 || p.F % tf == 0 || p.F >= 16) &&
           p.C0 % 4 == 0 && p.C1 % 4 == 0;
}

bool conv3f3c_flat_gn_ok(int N, int Npad, int H, int W) {
    static const int ok = debug_switch("DPC_CONV3F3C", 1) && debug_switch("DPC_CONV2D_LOADER_WAVES", 1) && debug_switch("DPC_CONV2D_FUSED_GN", 1);
    return ok && H % 8 == 0 && W % 8 == 0 && N % 64 == 0 && N == Npad;
}
long long conv3f3c_flat_gn_entries(int H, int W) { return (long long)(H / 8) * (W / 8); }

int launch_conv3f3c(const Conv3hParams& p, hipStream_t s) {
    using namespace f3c;
    const bool wide = p.Npad % 128 == 0 && p.N > 64;
    const int tf = wide ? 4 : 8;
    const long long tiles = (long long)p.B * ((p.F + tf - 1) / tf) * (p.H / 8) * (p.W / 8);
    const long long nwg = tiles * (p.Npad / (wide ? 128 : 64));
    DPC_REQUIRE(nwg < (1ll << 31), "conv3f3c: too many tiles");
    static int ncu = 0;
    static DeviceOnce once;
    if (!once) {
        int dev = 0;
        hipDeviceProp_t prop;
        DPC_HIP(hipGetDevice(&dev));
        DPC_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = std::max(8, prop.multiProcessorCount / 8 * 8);
        DPC_HIP(hipFuncSetAttribute((const void*)conv3f3c_kernel<64, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3f3c_kernel<128, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3f3c_kernel<64, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        DPC_HIP(hipFuncSetAttribute((const void*)conv3f3c_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        once = true;
    }
    Conv3hParams pd = p;
    pd.total_wg = (int)nwg;
    const unsigned grid = (unsigned)std::min<long long>(nwg, cu_budget(ncu));
    if (p.kd == 1) {
        if (wide) hipLaunchKernelGGL((conv3f3c_kernel<128, 1>), dim3(grid), dim3(512), 2 * HBS, s, pd);
        else hipLaunchKernelGGL((conv3f3c_kernel<64, 1>), dim3(grid), dim3(512), 2 * HBS, s, pd);
    } else if (wide) hipLaunchKernelGGL((conv3f3c_kernel<128, 3>), dim3(grid), dim3(512), 2 * HBS, s, pd);
    else hipLaunchKernelGGL((conv3f3c_kernel<64, 3>), dim3(grid), dim3(512), 2 * HBS, s, pd);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
