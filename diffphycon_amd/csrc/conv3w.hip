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
//   * Loader waves 4-7: load the 6 halo frames of an (h, w, channel-quad) item, apply the producer's GroupNorm + (scale, shift) +
//     SiLU, form V0..V3 of both pairs in fp32, pre-scale by 2^3 (|V| <= 2 |d|: same |x| <= 4094 range as the direct kernels),
//     split into the two fp16 planes and write the 8 TRANSFORMED frames to the double-buffered swizzled halo (800 points x 64 B).
//   * MFMA waves 0-3: wave k owns Winograd component k of the whole tile (4 slabs of 32 points x 64 channels, transposed
//     accumulators exactly as conv3f3c), streams ITS transformed weights U_k ([4][9 taps][chunk][n][2 planes][16] fp16 made by
//     launch_pack_weights_w3) with the same running pointer / 3-deep register ring, 9 taps x 24 MFMAs per chunk.
//   * Epilogue: the output transform crosses waves.  Per frame pair the four waves park their 64 x 64 component in the halo
//     buffer they just left (64 KB), then wave w reads the three components of output frame parity (w & 1), channel half (w >> 1),
//     combines, adds bias, emits the GroupNorm partial sums and stores dwordx4.  Four extra workgroup barriers per tile; the
//     loader waves join them (they would otherwise refill that buffer).
// Rounding: U_k are formed in fp32 from the fp32 weights before the split, V_k in fp32 after the fused activation; the products
// are the same 22-bit f16x3 products with fp32 accumulation.  F(2,3) has transform constants 1 and 1/2 only: against an fp64
// convolution the error is that of the direct kernel within a factor ~1.5 (tests/test_gpu_ops.py, tools/f16x3_error.py).
// Reference op: nn.Conv3d(dim, dim_out, (3,3,3), padding=(1,1,1)) in Block (video_diffusion_pytorch_conv3d.py:189-204).
#include "common.h"
#include "f3c.h"

namespace dpc {

namespace w3 {
constexpr float SAW = 8.0f;                 // activation pre-scale (the transformed operand is a sum of two activations)
constexpr int TFO = 4;                      // output frames per tile
constexpr int HFI = 6;                      // input halo frames
constexpr int NPT = 800;                    // transformed halo points per buffer: 8 frames x 10 x 10
constexpr int ITEMS = 400;                  // loader work items: 100 (h, w) x 4 channel quads
}  // namespace w3

__global__ __launch_bounds__(512, 2) void conv3w_kernel(Conv3hParams p) {
    using namespace f3c;
    using namespace w3;
    constexpr int MT = 4, NT = 2, NTAPS = 9;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w3[];
    unsigned char* halo = smem_w3;                      // two buffers at 0 and HBS

    const float descale = 1.0f / (SAW * SW);
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

    if (wave >= 4) {
        // ======================================================================================= loader waves
        const int ltid = tid - 256;
        const bool two = ltid + 256 < ITEMS;              // threads 0..143 own a second item
        int hdst[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = ltid + 256 * i, hw = q >> 2, quad = q & 3;
            hdst[i] = slot0(0, hw / 10, hw % 10, quad >> 1) + (quad & 1) * 8;
        }
        const int hslot = (ltid & 3) * 4;
        unsigned hokm = 0;                                // bit 6 i + fi: input frame fi of item i is inside the tensor
        int hpt[2];
        const long long fstride = (long long)p.H * p.W;
        const float* xb0 = nullptr;
        const float* xb1 = nullptr;
        int b_cur = 0;
        auto setup_tile = [&](int j) {
            int n0, w0, h0, f0, b;
            decode(j, n0, w0, h0, f0, b);
            b_cur = b;
            xb0 = p.a0 + (long long)b * p.F * p.H * p.W * p.C0;
            xb1 = p.a1 ? p.a1 + (long long)b * p.F * p.H * p.W * p.C1 : nullptr;
            hokm = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hw = (ltid + 256 * i) >> 2;
                const int h = h0 - 1 + hw / 10, w = w0 - 1 + hw % 10;
                const bool in = (i == 0 || two) && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
#pragma unroll
                for (int fi = 0; fi < HFI; ++fi)
                    if (in && (unsigned)(f0 - 1 + fi) < (unsigned)p.F) hokm |= 1u << (6 * i + fi);
                hpt[i] = ((f0 - 1) * p.H + h) * p.W + w;
            }
        };
        auto produce = [&](int kc, int boff) {
            const int c = kc * KC + hslot;
            const float* src;
            int cs, cc;
            if (c < p.C0) { src = xb0; cs = p.C0; cc = c; }
            else { src = xb1; cs = p.C1; cc = c - p.C0; }
            const bool cok = c < K;
            f32x4 d[2][HFI];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int fi = 0; fi < HFI; ++fi) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (cok && ((hokm >> (6 * i + fi)) & 1))
                        v = *reinterpret_cast<const f32x4*>(src + ((long long)hpt[i] + fi * fstride) * cs + cc);
                    d[i][fi] = v;
                }
            if (p.in_coef && cok) {
                // producer's GroupNorm + (scale + 1, shift) + SiLU (Block.forward, ...conv3d.py:196-204); the zero padding applies
                // to the ACTIVATED tensor, so out-of-range frames / rows / columns stay 0
                const f32x4* cf = reinterpret_cast<const f32x4*>(p.in_coef) + ((long long)b_cur * (K >> 2) + (c >> 2)) * 5;
                const f32x4 mu = cf[0], ga = cf[1], be = cf[2], sc = cf[3], sh = cf[4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int fi = 0; fi < HFI; ++fi)
                        if ((hokm >> (6 * i + fi)) & 1) {
                            f32x4 y = (d[i][fi] - mu) * ga + be;
                            y = y * sc + sh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = y[e] / (1.0f + expf(-y[e]));
                            d[i][fi] = y;
                        }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i == 0 || two) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) {
                        const f32x4 d0 = d[i][2 * pr], d1 = d[i][2 * pr + 1], d2 = d[i][2 * pr + 2], d3 = d[i][2 * pr + 3];
                        const f32x4 v[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            uint2 p1, p2;
                            split2(v[k] * SAW, p1, p2);
                            const int dst = hdst[i] + (pr * 4 + k) * 6400 + boff;
                            *reinterpret_cast<uint2*>(halo + dst) = p1;
                            *reinterpret_cast<uint2*>(halo + (dst ^ 32)) = p2;
                        }
                    }
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
            if (kc == 0) {                                // tile finished: the MFMA waves exchange components through buffer s & 1
                wg_barrier(); wg_barrier(); wg_barrier(); wg_barrier();
            }
        }
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
    const unsigned char* wroot = reinterpret_cast<const unsigned char*>(p.wpw) + (long long)wave * NTAPS * wtap;
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

    const int par = wave & 1, ntr = wave >> 1;            // epilogue role: output frame parity, channel half
    wg_barrier();                                         // step 0 is in buffer 0
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
            lda_pair(0, 0);
            lda_pair(0, 1);
            auto tap_body = [&](int tap) {
                ldw(w[(tap + 2) % 3]);                    // two taps ahead (the ring runs on across chunks and tiles)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};     // small terms first; PA: activation plane, PB: weight plane
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
                    for (int term = 0; term < 3; ++term)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[2 * pr + q][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    w[tap % 3][nt][PB[term]], a[2 * pr + q][PA[term]], acc[2 * pr + q][nt], 0, 0, 0);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    if (tap < NTAPS - 1) lda_pair(tap + 1, pr);   // rolling A set: re-load behind the other pair's MFMAs
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll
            for (int tap = 0; tap < NTAPS; ++tap) tap_body(tap);
            // MFMA B-operand guard (see igemm6.hip): nothing may overwrite the activation fragments while the last MFMA reads them
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            wg_barrier();                                  // next chunk's buffer is complete; this one may be overwritten
            boff ^= HBS;
        }

        // ---- epilogue.  acc[2 pr + q][nt][4g + e] = component k of pair pr, point (rows 4q.., lane_hw(l31)), channel nt*32 + 8g + 4hh + e.
        // Exchange area = the buffer of the last chunk (boff ^ HBS after the toggle): [k][q][nt][g][lane] x 16 B = 64 KB.
        unsigned char* xch = halo + (boff ^ HBS);
        const int nbase = n0 + ntr * 32 + 4 * hh;
        const long long tile = ((long long)(f0 / TFO) * nth + h0 / 8) * ntw + w0 / 8;
        f32x4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bv[g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nbase + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
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
                        *reinterpret_cast<f32x4*>(xch + (((((wave * 2 + q) * 2 + nt) * 4 + g) * 64 + lane) << 4)) = v;
                    }
            lds_done_barrier();
            f32x4 o[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    auto rd = [&](int k) {
                        return *reinterpret_cast<const f32x4*>(xch + (((((k * 2 + q) * 2 + ntr) * 4 + g) * 64 + lane) << 4));
                    };
                    const f32x4 m1 = rd(1), m2 = rd(2);
                    const f32x4 m03 = rd(par ? 3 : 0);
                    o[q][g] = par ? (m1 - m2) - m03 : (m03 + m1) + m2;
                }
            lds_done_barrier();
            const int f = f0 + 2 * pr + par;
            float gs[16], gq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { gs[r] = 0.f; gq[r] = 0.f; }
            if (f < p.F) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float* base = p.out + ((((long long)b * p.F + f) * p.H + h0 + 4 * q + lh) * p.W + w0 + lw) * p.N + nbase;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = o[q][g][e] * descale + bv[g][e];
                            gs[4 * g + e] += v[e];
                            gq[4 * g + e] += v[e] * v[e];
                        }
                        *reinterpret_cast<f32x4*>(base + 8 * g) = v;
                    }
                }
            }
            if (p.gn_part) {
                // GroupNorm statistics of the OUTPUT: per channel, this wave's 2 slabs x 32 points of frame f (zeros for a frame past
                // the end).  Transpose tree over the 32 lanes of a half-wave as in conv3f3c; entry = 2 pr + par of the tile.
                float* gdst = p.gn_part + (((long long)b * ((long long)ntf * nth * ntw) + tile) * 4 + 2 * pr + par) * p.N * 2;
                float tot[2];
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    float* x = which ? gq : gs;
#pragma unroll
                    for (int half = 8; half >= 1; half >>= 1) {
                        const bool up = (l31 & (half * 2)) != 0;
#pragma unroll
                        for (int i = 0; i < half; ++i) {
                            const float send = up ? x[i] : x[i + half];
                            const float keep = up ? x[i + half] : x[i];
                            x[i] = keep + __shfl_xor(send, half * 2, 64);
                        }
                    }
                    tot[which] = x[0] + __shfl_xor(x[0], 1, 64);
                }
                if ((l31 & 1) == 0) {
                    const int r = ((l31 >> 4) & 1) * 8 + ((l31 >> 3) & 1) * 4 + ((l31 >> 2) & 1) * 2 + ((l31 >> 1) & 1);
                    const int n = n0 + ntr * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    gdst[n * 2] = tot[0];
                    gdst[n * 2 + 1] = tot[1];
                }
            }
            asm volatile("" ::: "memory");
        }
    }
}

static int conv3w_enabled() {
    static const int ok = [] { const char* e = getenv("DPC_CONV3W"); return e ? atoi(e) : 1; }();
    return ok;
}

// shape-only rule (never the batch): the Winograd form is taken for 3x3x3 convolutions whose planes tile by 8 x 8 and whose
// channel counts fit the 64-wide column tile; F % 4 != 0 runs partial frame tiles (F >= 16 only, as the direct kernels)
bool conv3w_shape_ok(int F, int H, int W, int N, int Npad) {
    return conv3w_enabled() && H % 8 == 0 && W % 8 == 0 && N % 64 == 0 && N == Npad && (F % 4 == 0 || F >= 16);
}

bool conv3w_supported(const Conv3hParams& p) {
    return p.wpw && p.kd != 1 && conv3w_shape_ok(p.F, p.H, p.W, p.N, p.Npad) && p.C0 % 4 == 0 && p.C1 % 4 == 0 && p.act_scale == 0.f;
}

long long conv3w_gn_entries(int F, int H, int W) { return (long long)((F + 3) / 4) * (H / 8) * (W / 8) * 4; }

int launch_conv3w(const Conv3hParams& p, hipStream_t s) {
    using namespace f3c;
    const long long tiles = (long long)p.B * ((p.F + 3) / 4) * (p.H / 8) * (p.W / 8);
    const long long nwg = tiles * (p.Npad / 64);
    DPC_REQUIRE(nwg < (1ll << 31), "conv3w: too many tiles");
    static int ncu = 0;
    static bool once = false;
    if (!once) {
        int dev = 0;
        hipDeviceProp_t prop;
        DPC_HIP(hipGetDevice(&dev));
        DPC_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = std::max(8, prop.multiProcessorCount / 8 * 8);
        DPC_HIP(hipFuncSetAttribute((const void*)conv3w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HBS));
        once = true;
    }
    Conv3hParams pd = p;
    pd.total_wg = (int)nwg;
    const unsigned grid = (unsigned)std::min<long long>(nwg, ncu);
    hipLaunchKernelGGL(conv3w_kernel, dim3(grid), dim3(512), 2 * HBS, s, pd);
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

size_t conv3w_packed_bytes(int Npad, int K) { return (size_t)36 * ((K + 15) / 16) * Npad * 64; }

int launch_pack_weights_w3(const float* w, void* wp, int N, int Npad, int K, hipStream_t s) {
    const int kchunks = (K + 15) / 16;
    const long long total = (long long)36 * kchunks * Npad * 16;
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(pack_weights_w3_kernel, dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), N, Npad, K,
                       kchunks, f16x3_weight_overflow_flag());
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
