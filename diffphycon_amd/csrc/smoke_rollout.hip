// Smoke PDE evaluator on gfx950: the reference's phi-based rollout (dataset/apps/evaluate_solver.py:205-310 `solver`,
// :118-147 `get_envolve`; phi/flow.py:294-327; phi/math/nd.py:332-427,602-614; phi/solver/sparse.py:27-128;
// phi/solver/base.py:56-104; phi/math/scipy_backend.py:58-102,181-185) as ONE persistent kernel:
// one 1024-thread workgroup per trajectory runs all time steps, including the data-dependent CG loop, without any
// host round trip.  Everything is fp64 (density storage fp32) with the reference's exact operation order, so the
// result is bit-identical to the NumPy/SciPy run:
//
//  * np.sum is a fixed tree (8192-element chunks, pairwise halves, <=128-element leaves with 8 strided accumulators).
//    Thread (g, j) = (tid / 8, tid % 8) owns accumulator j of "group" g (a pairwise node of <= 136 elements = one
//    or two leaves), i.e. elements off + j + 8k: the leaf partial sums are THREAD-LOCAL sequential adds, the 8-lane
//    combine ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) is three xor-shuffles, and the tree over the 2 x 64 groups is a
//    6-level xor butterfly that every wave performs redundantly from 1 KB of LDS (one barrier per dot product pair).
//  * the CSC mat-vec accumulates a row in column order (y-1,x),(y,x-1),(y,x),(y,x+1),(y+1,x); the momentum vector
//    lives in LDS (129 KB of the CU's 160 KB) with zero padding, residual / A*m stay in VGPRs, x streams through L2.
//  * CG quirk kept: residual and momentum alias during iteration 1 (base.py:74), giving m = r + b*r there.
//  * advection = scipy's generic linear interpn on a float32 field with fp64 weights (1*(1-y0))*(1-y1)..., samples
//    whose clamped back-traced coordinate exceeds n-1 read 0.
#include <vector>

#include "common.h"

namespace dpc {

constexpr int SM_THREADS = 1024;
constexpr int SM_GROUPS = 128;
constexpr int SM_NS = 17;            // slots per thread: element base + 8*s (a group spans <= 136 contiguous elements)
constexpr int SM_FULL = 15;          // slots [0, SM_FULL) are valid in every lane (host-checked: every leaf >= 120 elements or
                                     // part of a 2-leaf group of >= 120); only slots 15 and 16 need predication
constexpr int SM_PAD = 160;          // zero padding (doubles) on both sides of the LDS momentum vector (>= n + 8*2)

struct SmokeParams {
    int n, n1, N, rim, nb, target, maxrem;
    int B, num_t, nt, nx, ti, si;
    int max_it, dens_f32, ofs, oss, mode;     // mode 0: rollout, 1: pressure solve only (div in x_in), 2: one advect
    double accuracy, dt;
    const float* vel0;
    long long vel0_bstride;
    const float* dens0;
    const float* c1;
    const float* c2;
    void* densitys;
    void* zero_densitys;
    double* velocitys;
    double* smoke_out;
    int* cg_iters;
    // workspace
    double* v;         // [B][n1*n1*2]
    double* x;         // [B][N+1]
    float* d;          // [B][4][n*n]
    const int* grp;    // [128][4] offA, lenA, offB, lenB (pairwise groups of a length-N vector)
    const unsigned char* cf;   // [N+1]
    const unsigned char* vm;   // [n1*n1] bit0: component 0 (x) mask, bit1: component 1 (y) mask
    const unsigned char* bk;   // [n1*n1] 0 = no bucket, k+1 = bucket k
};

// ------------------------------------------------------------------------------------------------ domain tables
__global__ void smoke_tables_kernel(const signed char* __restrict__ fluid, const signed char* __restrict__ active, int n,
                                    unsigned char* __restrict__ cf, unsigned char* __restrict__ vm,
                                    unsigned char* __restrict__ bk, int nb, const int* __restrict__ rect) {
    const int n1 = n + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1 * n1) return;
    const int y = i / n1, x = i % n1;
    auto fl = [&](int yy, int xx) -> int { return (yy < 0 || yy >= n || xx < 0 || xx >= n) ? 1 : (int)fluid[yy * n + xx]; };
    auto ac = [&](int yy, int xx) -> int { return (yy < 0 || yy >= n || xx < 0 || xx >= n) ? 0 : (int)active[yy * n + xx]; };
    // phi/flow.py:456-473: velocity mask = min of the two adjacent (padded-with-ones) fluid cells
    const int m0 = min(fl(y, x), fl(y, x - 1)), m1 = min(fl(y, x), fl(y - 1, x));
    vm[i] = (unsigned char)((m0 ? 1 : 0) | (m1 ? 2 : 0));
    int b = 0;
    for (int k = 0; k < nb; ++k) {       // evaluate_solver.py:150-171 (later rectangles do not overlap earlier ones)
        const int by = rect[4 * k], bx = rect[4 * k + 1], ly = rect[4 * k + 2], lx = rect[4 * k + 3];
        if (y >= by && y < by + ly && x >= bx && x < bx + lx) b = k + 1;
    }
    bk[i] = (unsigned char)b;
    if (y < n && x < n) {                // phi/solver/sparse.py:44-76
        const int c = ac(y, x);
        const int lo0 = ac(y - 1, x) * c, up0 = ac(y + 1, x) * c, lo1 = ac(y, x - 1) * c, up1 = ac(y, x + 1) * c;
        int diag = fl(y + 1, x) + fl(y - 1, x) + fl(y, x + 1) + fl(y, x - 1);
        diag = max(diag, 1);             // min(center, -1) on the negated sum
        cf[y * n + x] = (unsigned char)((lo0 ? 1 : 0) | (lo1 ? 2 : 0) | (up1 ? 4 : 0) | (up0 ? 8 : 0) | (diag << 4));
    }
    if (i == 0) cf[n * n] = 0;           // dummy element
}

// ------------------------------------------------------------------------------------------------ reductions
// A group is a CONTIGUOUS run of lenA + lenB elements (leaf B follows leaf A and lenA is a multiple of 8 whenever B
// exists), so lane j's slot s is element base + 8*s with base = off + j: slots [0, kA) are accumulator j of leaf A,
// [kA, K) of leaf B, and slot K holds this lane's remainder element (numpy adds the n % 8 tail sequentially).
struct BlockCtx {
    int tid, g, j, lane, wave;
    int base;               // first element of this lane
    int K, nvalid;          // K regular slots; nvalid = K + (this lane owns a remainder element, slot K)
    bool two;               // the group has two leaves: slots [0, 8) leaf A, [8, K) leaf B, remainder in B
    int maxrem;
};

__device__ __forceinline__ double shx(double v, int m) { return __shfl_xor(v, m, 64); }

// NV simultaneous sums of one group (1 or 2 pairwise leaves) of per-slot values val(s, p[NV]); the result is identical
// on the 8 lanes of the group.  Invalid slots must evaluate to 0.  Leaf A of a two-leaf group is always 64 elements
// (slots [0, 8)), so the sequential accumulator chain is split once, at slot 8: single-leaf groups continue the chain,
// two-leaf groups restart it from 0 for leaf B (host-checked in build_group_table).
template <int NV, class F>
__device__ __forceinline__ void group_sums(const BlockCtx& c, F val, double (&out)[NV]) {
    double a[NV], ch[NV], rem[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { a[v] = 0.0; rem[v] = 0.0; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        double p[NV]; val(s, p);
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v] = a[v] + p[v];
        if (s % 4 == 3) {
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(a[v]) :: "memory");
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) ch[v] = c.two ? 0.0 : a[v];
#pragma unroll
    for (int s = 8; s < SM_FULL; ++s) {
        double p[NV]; val(s, p);
#pragma unroll
        for (int v = 0; v < NV; ++v) ch[v] = ch[v] + p[v];
        if (s % 4 == 3) {
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(ch[v]) :: "memory");
        }
    }
#pragma unroll
    for (int s = SM_FULL; s < SM_NS; ++s) {
        double p[NV]; val(s, p);
#pragma unroll
        for (int v = 0; v < NV; ++v) { ch[v] = ch[v] + ((s < c.K) ? p[v] : 0.0); rem[v] = (s == c.K) ? p[v] : rem[v]; }
    }
    // single leaf: a = chain, b = 0; two leaves: a = first 8 slots, b = chain
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double aa = c.two ? a[v] : ch[v], bb = c.two ? ch[v] : 0.0;
        aa = aa + shx(aa, 1); aa = aa + shx(aa, 2); aa = aa + shx(aa, 4);
        bb = bb + shx(bb, 1); bb = bb + shx(bb, 2); bb = bb + shx(bb, 4);
        if (c.maxrem > 0) {
            const int base = c.lane & ~7;
            for (int q = 0; q < c.maxrem; ++q) {
                const double t = __shfl(rem[v], base + q, 64);
                aa = aa + (c.two ? 0.0 : t);
                bb = bb + (c.two ? t : 0.0);
            }
        }
        out[v] = aa + bb;
    }
}

// Tree over the 128 group sums stored in LDS (2 chunks x 64 groups, adjacent pairs), identical in every lane.
__device__ __forceinline__ double tree_sum(const BlockCtx& c, const double* gs) {
    double v = gs[2 * c.lane] + gs[2 * c.lane + 1];
    v = v + shx(v, 1); v = v + shx(v, 2); v = v + shx(v, 4); v = v + shx(v, 8); v = v + shx(v, 16);
    const double c0 = __shfl(v, 0, 64), c1 = __shfl(v, 32, 64);
    return c0 + c1;
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = fmax(v, shx(v, m));
    return v;
}

// ------------------------------------------------------------------------------------------------ CG
// A*m for element e from the zero-padded LDS vector (index 0 of `m` = element 0).
__device__ __forceinline__ double stencil(const double* me, unsigned c, int n) {
    double y = 0.0 + ((c & 1) ? 1.0 : 0.0) * me[-n];
    y = y + ((c & 2) ? 1.0 : 0.0) * me[-1];
    y = y + (-(double)(c >> 4)) * me[0];
    y = y + ((c & 4) ? 1.0 : 0.0) * me[1];
    y = y + ((c & 8) ? 1.0 : 0.0) * me[n];
    return y;
}

// Solves A x = k (k passed in r[] with invalid slots 0, already stored in the LDS vector by the caller, x zeroed by
// the caller).  Returns the iteration count.  Branch-free per slot: slots >= SM_FULL that a lane does not own read
// in-range (padded) addresses, get a zero stencil byte and send their stores to a per-thread dummy location.
__device__ __forceinline__ int cg_solve(const SmokeParams& P, const BlockCtx& c, double* __restrict__ xg, double* m,
                                        double* gs, double* wmax, const unsigned char* cf, double (&r)[SM_NS]) {
    double Am[SM_NS];
    double* mb = m + c.base;
    double* xb = xg + c.base;
    const unsigned char* cb = cf + c.base;
    double* mdummy = m + (P.N + 1 + SM_PAD) + c.tid;          // LDS [1024] behind the padded vector
    double* xdummy = xg + (P.N + 1) + c.tid;                  // global [1024] behind x
    double* mst[SM_NS - SM_FULL];
    double* xst[SM_NS - SM_FULL];
#pragma unroll
    for (int s = SM_FULL; s < SM_NS; ++s) {
        mst[s - SM_FULL] = (s < c.nvalid) ? mb + 8 * s : mdummy;
        xst[s - SM_FULL] = (s < c.nvalid) ? xb + 8 * s : xdummy;
    }
    auto apply = [&]() {
#pragma unroll
        for (int s = 0; s < SM_NS; ++s) {
            unsigned cc = cb[8 * s];
            if (s >= SM_FULL) cc = (s < c.nvalid) ? cc : 0u;
            Am[s] = stencil(mb + 8 * s, cc, P.n);
            // pin the consumer here: otherwise the scheduler issues all 85 LDS loads first and spills their results
            // (VGPR budget at 16 waves/CU: 128)
            if (s % 2 == 1) asm volatile("" : "+v"(Am[s]), "+v"(Am[s - 1]) :: "memory");
        }
    };
    apply();
    double rmax = 0.0;
#pragma unroll
    for (int s = 0; s < SM_NS; ++s) rmax = fmax(rmax, fabs(r[s]));
    rmax = wave_max(rmax);
    if (c.lane == 0) wmax[c.wave] = rmax;
    __syncthreads();
    rmax = wmax[0];
#pragma unroll
    for (int w = 1; w < SM_THREADS / 64; ++w) rmax = fmax(rmax, wmax[w]);
    int it = 0;
    bool first = true;
    while (rmax >= P.accuracy && it < P.max_it) {
        double g12[2];
        group_sums<2>(c, [&](int s, double (&p)[2]) { const double mv = mb[8 * s]; p[0] = mv * Am[s]; p[1] = mv * r[s]; }, g12);
        if (c.j == 0) { gs[c.g] = g12[0]; gs[SM_GROUPS + c.g] = g12[1]; }
        __syncthreads();
        const double tmp = tree_sum(c, gs), mr = tree_sum(c, gs + SM_GROUPS);
        const double a = mr / tmp;
        double lmax = 0.0;
#pragma unroll
        for (int s = 0; s < SM_NS; ++s) {
            const double xn = xb[8 * s] + a * mb[8 * s];
            if (s < SM_FULL) xb[8 * s] = xn; else *xst[s - SM_FULL] = xn;
            r[s] = r[s] - a * Am[s];
            lmax = fmax(lmax, fabs(r[s]));
            if (s % 4 == 3) asm volatile("" : "+v"(r[s]), "+v"(lmax) :: "memory");
        }
        double g3[1];
        group_sums<1>(c, [&](int s, double (&p)[1]) { p[0] = r[s] * Am[s]; }, g3);
        lmax = wave_max(lmax);
        if (c.j == 0) gs[2 * SM_GROUPS + c.g] = g3[0];
        if (c.lane == 0) wmax[16 + c.wave] = lmax;
        __syncthreads();
        const double rAm = tree_sum(c, gs + 2 * SM_GROUPS);
        rmax = wmax[16];
#pragma unroll
        for (int w = 1; w < SM_THREADS / 64; ++w) rmax = fmax(rmax, wmax[16 + w]);
        const double b = -rAm / tmp;
        // own-element update only (neighbours were read by the stencil two barriers ago)
#pragma unroll
        for (int s = 0; s < SM_NS; ++s) {
            const double mn = r[s] + b * (first ? r[s] : mb[8 * s]);
            if (s < SM_FULL) mb[8 * s] = mn; else *mst[s - SM_FULL] = mn;
        }
        first = false;
        __syncthreads();
        apply();
        ++it;
    }
    return it;
}

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ void store_dens(const SmokeParams& P, void* base, long long frame_idx, int y, int x, float val) {
    // output frame [n1/oss, n1/oss]; only cells on the sub-sampling lattice are written
    if ((y % P.oss) | (x % P.oss)) return;
    const int w = P.n1 / P.oss;
    const long long o = frame_idx * w * w + (long long)(y / P.oss) * w + x / P.oss;
    if (P.dens_f32) ((float*)base)[o] = val; else ((double*)base)[o] = (double)val;
}

__device__ __forceinline__ void setup_ctx(const SmokeParams& P, BlockCtx& c) {
    c.tid = threadIdx.x; c.g = c.tid >> 3; c.j = c.tid & 7; c.lane = c.tid & 63; c.wave = c.tid >> 6;
    c.maxrem = P.maxrem;
    const int offA = P.grp[4 * c.g], lenA = P.grp[4 * c.g + 1], lenB = P.grp[4 * c.g + 3];
    c.base = offA + c.j;
    c.K = (lenA + lenB) / 8;
    c.two = lenB > 0;
    c.nvalid = c.K + (c.j < (lenA + lenB) % 8 ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------ the kernel
// Pressure solve only: k = x on entry, x = solution on exit (operator-level entry for the parity tests).
__global__ __launch_bounds__(SM_THREADS) void smoke_cg_kernel(SmokeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* m_all = (double*)smem;
    double* m = m_all + SM_PAD;
    double* gs = m_all + (SM_PAD + P.N + 1 + SM_PAD) + SM_THREADS;
    double* wmax = gs + 3 * SM_GROUPS;
    unsigned char* cf = (unsigned char*)(wmax + 48);
    const int b = blockIdx.x, N = P.N;
    BlockCtx c;
    setup_ctx(P, c);
    const int tid = c.tid;
    for (int i = tid; i < SM_PAD + N + 1 + SM_PAD; i += SM_THREADS) m_all[i] = 0.0;
    for (int i = tid; i <= N + SM_PAD; i += SM_THREADS) cf[i] = i <= N ? P.cf[i] : (unsigned char)0;
    double* xg = P.x + (long long)b * (N + 1 + SM_THREADS);
    __syncthreads();
    double r[SM_NS];
#pragma unroll
    for (int s = 0; s < SM_NS; ++s) { const double t = xg[c.base + 8 * s]; r[s] = (s < c.nvalid) ? t : 0.0; }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SM_NS; ++s)
        if (s < c.nvalid) { m[c.base + 8 * s] = r[s]; xg[c.base + 8 * s] = 0.0; }
    __syncthreads();
    const int it = cg_solve(P, c, xg, m, gs, wmax, cf, r);
    if (tid == 0 && P.cg_iters) P.cg_iters[b] = it;
}

__global__ __launch_bounds__(SM_THREADS) void smoke_rollout_kernel(SmokeParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* m_all = (double*)smem;
    double* m = m_all + SM_PAD;                                        // element 0
    double* gs = m_all + (SM_PAD + P.N + 1 + SM_PAD) + SM_THREADS;      // [3][128] (behind the dummy store slots)
    double* wmax = gs + 3 * SM_GROUPS;                                  // [32] wave maxima, [32..40) bucket totals
    double* outs = wmax + 32;                                           // [8] smoke_outs, evaluate_solver.py:236
    unsigned char* cf = (unsigned char*)(wmax + 48);                    // [N+1]
    const int b = blockIdx.x;
    const int n = P.n, n1 = P.n1, N = P.N;
    BlockCtx c;
    setup_ctx(P, c);
    const int tid = c.tid;
    for (int i = tid; i < SM_PAD + N + 1 + SM_PAD; i += SM_THREADS) m_all[i] = 0.0;
    for (int i = tid; i <= N + SM_PAD; i += SM_THREADS) cf[i] = i <= N ? P.cf[i] : (unsigned char)0;
    double* xg = P.x + (long long)b * (N + 1 + SM_THREADS);
    double* v = P.v + (long long)b * n1 * n1 * 2;
    __syncthreads();

    float* dA = P.d + (long long)b * 4 * n * n;        // density (current), then +n*n: next; +2: zero-density cur; +3: next
    float* dcur = dA;
    float* dnxt = dA + (long long)n * n;
    float* zcur = dA + 2LL * n * n;
    float* znxt = dA + 3LL * n * n;
    const float* vel0 = P.vel0 + (long long)b * P.vel0_bstride;
    const float* dens0 = P.dens0 + (long long)b * P.nx * P.nx;
    const float* c1 = P.c1 + (long long)b * P.nt * P.nx * P.nx;
    const float* c2 = P.c2 + (long long)b * P.nt * P.nx * P.nx;
    const int oT = (P.num_t + P.ofs - 1) / P.ofs;
    const int ow = n1 / P.oss;
    if (tid < 8) outs[tid] = 0.0;

    // ---- frame 0: fields from the inputs (evaluate_solver.py:223-264)
    for (int i = tid; i < n1 * n1 * 2; i += SM_THREADS) v[i] = (double)vel0[i];
    for (int i = tid; i < n * n; i += SM_THREADS) {
        const int y = i / n, x = i % n;
        const float d0 = dens0[(y / P.si) * P.nx + x / P.si];
        dcur[i] = d0; zcur[i] = d0;
    }
    __syncthreads();

    for (int frame = 0; frame < P.num_t; ++frame) {
        if (frame > 0) {
            const int fr = frame - 1;
            const float* c1f = c1 + (long long)(fr / P.ti) * P.nx * P.nx;
            const float* c2f = c2 + (long long)(fr / P.ti) * P.nx * P.nx;
            // ---- get_envolve :118-140: interior keeps the previous velocity, the rim takes the control; then mask
            for (int i = tid; i < n1 * n1; i += SM_THREADS) {
                const int y = i / n1, x = i % n1;
                const bool interior = y >= P.rim && y < n1 - P.rim && x >= P.rim && x < n1 - P.rim;
                const unsigned mk = P.vm[i];
                double a0, a1;
                if (interior) { a0 = v[2 * i]; a1 = v[2 * i + 1]; }
                else { const int ci = (y / P.si) * P.nx + x / P.si; a0 = (double)c1f[ci]; a1 = (double)c2f[ci]; }
                v[2 * i] = a0 * ((mk & 1) ? 1.0 : 0.0);
                v[2 * i + 1] = a1 * ((mk & 2) ? 1.0 : 0.0);
            }
            __syncthreads();
            // ---- divergence (nd.py:367-377) -> r, m ; x = 0
            double r[SM_NS];
#pragma unroll
            for (int s = 0; s < SM_NS; ++s) {
                const int e = c.base + 8 * s;
                double dv = 0.0;
                if (s < c.nvalid) {
                    const int y = e / n, x = e % n;
                    const int i00 = (y * n1 + x) * 2;
                    dv = (v[i00 + 2 * n1 + 1] - v[i00 + 1]) + (v[i00 + 2] - v[i00]);
                    m[e] = dv;
                    xg[e] = 0.0;
                }
                r[s] = dv;
            }
            __syncthreads();
            const int it = cg_solve(P, c, xg, m, gs, wmax, cf, r);
            if (tid == 0 && P.cg_iters) P.cg_iters[(long long)b * (P.num_t - 1) + fr] = it;
            __syncthreads();
            // ---- v -= mask * grad p ; v *= mask   (flow.py:318-327, nd.py:602-614 symmetric padding; :144-145)
            for (int i = tid; i < n1 * n1; i += SM_THREADS) {
                const int y = i / n1, x = i % n1;
                const unsigned mk = P.vm[i];
                const int yc = min(y, n - 1), xc = min(x, n - 1);
                const double pc = xg[yc * n + xc];
                const double gy = pc - xg[max(y - 1, 0) * n + xc];
                const double gx = pc - xg[yc * n + max(x - 1, 0)];
                const double k0 = (mk & 1) ? 1.0 : 0.0, k1 = (mk & 2) ? 1.0 : 0.0;
                v[2 * i] = (v[2 * i] - gx * k0) * k0;
                v[2 * i + 1] = (v[2 * i + 1] - gy * k1) * k1;
            }
            __syncthreads();
            // ---- advect both density fields (nd.py:422-427, scipy_backend.py:58-78)
            for (int i = tid; i < n * n; i += SM_THREADS) {
                const int y = i / n, x = i % n;
                const int i00 = (y * n1 + x) * 2;
                const double cvy = (v[i00 + 2 * n1 + 1] + v[i00 + 1]) / 2.0;
                const double cvx = (v[i00 + 2] + v[i00]) / 2.0;
                double cy = (double)(float)y - cvy * P.dt;
                double cx = (double)(float)x - cvx * P.dt;
                cy = fmax(0.0, fmin((double)n, cy));
                cx = fmax(0.0, fmin((double)n, cx));
                const bool oob = cy > (double)(n - 1) || cx > (double)(n - 1);
                const int iy = min(max((int)floor(cy), 0), n - 2), ix = min(max((int)floor(cx), 0), n - 2);
                const double y0 = (cy - (double)iy) / 1.0, y1 = (cx - (double)ix) / 1.0;
                const double w00 = (1.0 * (1.0 - y0)) * (1.0 - y1), w01 = (1.0 * (1.0 - y0)) * y1;
                const double w10 = (1.0 * y0) * (1.0 - y1), w11 = (1.0 * y0) * y1;
                const int q = iy * n + ix;
                double a = 0.0 + (double)dcur[q] * w00;
                a = a + (double)dcur[q + 1] * w01; a = a + (double)dcur[q + n] * w10; a = a + (double)dcur[q + n + 1] * w11;
                double z = 0.0 + (double)zcur[q] * w00;
                z = z + (double)zcur[q + 1] * w01; z = z + (double)zcur[q + n] * w10; z = z + (double)zcur[q + n + 1] * w11;
                dnxt[i] = oob ? 0.f : (float)a;
                znxt[i] = oob ? 0.f : (float)z;
            }
            __syncthreads();
            { float* t = dcur; dcur = dnxt; dnxt = t; t = zcur; zcur = znxt; znxt = t; }
        }
        // ---- bucket accounting on the zero-density field (:248-262, :279-304): group g = row g of the padded array
        double acc[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] = 0.0;
        float own[16];
        unsigned char ob[16];
        {
            const int y = c.g;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int x = c.j + 8 * k;
                const float dz = (y < n && x < n) ? zcur[y * n + x] : 0.f;
                const unsigned bkv = (y < n1 && x < n1) ? P.bk[y * n1 + x] : 0;
                own[k] = dz; ob[k] = (unsigned char)bkv;
                const double a = (double)dz;
                acc[0] = acc[0] + a * (bkv ? 1.0 : 0.0);         // concat
#pragma unroll
                for (int q = 0; q < 7; ++q) acc[1 + q] = acc[1 + q] + a * ((bkv == (unsigned)(q + 1)) ? 1.0 : 0.0);
                acc[8] = acc[8] + a;                              // everything
                acc[9] = acc[9] + a * (bkv ? 0.0 : 1.0);          // set_zero applied
            }
        }
        double* gs10 = m_all;      // the momentum vector is dead here
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            double a = acc[k];
            a = a + shx(a, 1); a = a + shx(a, 2); a = a + shx(a, 4);
            if (c.j == 0) gs10[k * SM_GROUPS + c.g] = a;
        }
        __syncthreads();
        double tot[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) tot[k] = tree_sum(c, gs10 + k * SM_GROUPS);
        __syncthreads();
        for (int i = tid; i < 10 * SM_GROUPS; i += SM_THREADS) gs10[i] = 0.0;    // restore the zero padding
        double remaining = tot[8];
        double o[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) o[q] = outs[q];
        if (tot[0] > 0.0) {
#pragma unroll
            for (int q = 0; q < 7; ++q) if (q < P.nb) o[q] = o[q] + tot[1 + q];
            remaining = tot[9];
            const int y = c.g;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int x = c.j + 8 * k;
                if (y < n && x < n) zcur[y * n + x] = (float)((double)own[k] * (ob[k] ? 0.0 : 1.0));
            }
        }
        double so = 0.0, tgt = 0.0;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            if (q < P.nb) so = so + o[q];
            if (q == P.target) tgt = o[q];
        }
        const double smoke = tgt / (so + remaining);
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < 7; ++q) outs[q] = o[q];
        }
        // ---- outputs of this frame
        if (frame % P.ofs == 0) {
            const long long fo = (long long)b * oT + frame / P.ofs;
            if (tid == 0 && P.smoke_out) P.smoke_out[fo] = smoke;
            if (P.densitys || P.zero_densitys) {
                for (int i = tid; i < n1 * n1; i += SM_THREADS) {
                    const int y = i / n1, x = i % n1;
                    const bool in = y < n && x < n;
                    if (P.densitys) store_dens(P, P.densitys, fo, y, x, in ? dcur[y * n + x] : 0.f);
                    if (P.zero_densitys) store_dens(P, P.zero_densitys, fo, y, x, in ? zcur[y * n + x] : 0.f);
                }
            }
            if (P.velocitys) {
                for (int i = tid; i < n1 * n1; i += SM_THREADS) {
                    const int y = i / n1, x = i % n1;
                    if ((y % P.oss) | (x % P.oss)) continue;
                    const long long o = (fo * ow * ow + (long long)(y / P.oss) * ow + x / P.oss) * 2;
                    P.velocitys[o] = v[2 * i]; P.velocitys[o + 1] = v[2 * i + 1];
                }
            }
        }
        __syncthreads();
    }
}

// Single advection step (operator-level entry for the parity tests): vel fp64 [n1,n1,2], dens f32 [n,n] -> out f32.
__global__ void smoke_advect_kernel(const double* __restrict__ v, const float* __restrict__ dens, float* __restrict__ out,
                                    int n, double dt) {
    const int n1 = n + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * n) return;
    const int y = i / n, x = i % n;
    const int i00 = (y * n1 + x) * 2;
    const double cvy = (v[i00 + 2 * n1 + 1] + v[i00 + 1]) / 2.0;
    const double cvx = (v[i00 + 2] + v[i00]) / 2.0;
    double cy = (double)(float)y - cvy * dt;
    double cx = (double)(float)x - cvx * dt;
    cy = fmax(0.0, fmin((double)n, cy));
    cx = fmax(0.0, fmin((double)n, cx));
    const bool oob = cy > (double)(n - 1) || cx > (double)(n - 1);
    const int iy = min(max((int)floor(cy), 0), n - 2), ix = min(max((int)floor(cx), 0), n - 2);
    const double y0 = (cy - (double)iy) / 1.0, y1 = (cx - (double)ix) / 1.0;
    const double w00 = (1.0 * (1.0 - y0)) * (1.0 - y1), w01 = (1.0 * (1.0 - y0)) * y1;
    const double w10 = (1.0 * y0) * (1.0 - y1), w11 = (1.0 * y0) * y1;
    const int q = iy * n + ix;
    double a = 0.0 + (double)dens[q] * w00;
    a = a + (double)dens[q + 1] * w01; a = a + (double)dens[q + n] * w10; a = a + (double)dens[q + n + 1] * w11;
    out[i] = oob ? 0.f : (float)a;
}

// ------------------------------------------------------------------------------------------------ host side
// numpy's pairwise tree for a length-N contiguous fp64 sum, cut at nodes of <= 136 elements ("groups").
struct PwGroup { int offA, lenA, offB, lenB; };
static void pw_groups(int off, int n, std::vector<PwGroup>& out, int depth, int& maxdepth, int& mindepth) {
    if (n <= 128) {
        out.push_back({off, n, 0, 0});
        maxdepth = std::max(maxdepth, depth); mindepth = std::min(mindepth, depth);
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    if (n <= 136) {
        out.push_back({off, n2, off + n2, n - n2});
        maxdepth = std::max(maxdepth, depth); mindepth = std::min(mindepth, depth);
        return;
    }
    pw_groups(off, n2, out, depth + 1, maxdepth, mindepth);
    pw_groups(off + n2, n - n2, out, depth + 1, maxdepth, mindepth);
}

// Fills tab[128][4]; returns false when the tree of N does not fit the kernel's fixed 2 x 64 butterfly.
static bool build_group_table(int N, int* tab, int* maxrem) {
    for (int i = 0; i < SM_GROUPS * 4; ++i) tab[i] = 0;
    for (int g = 0; g < SM_GROUPS; ++g) { tab[4 * g] = N; tab[4 * g + 2] = N; }
    *maxrem = 0;
    if (N < 64 || N > 2 * 8192) return false;
    for (int c = 0, s = 0; s < N; ++c, s += 8192) {
        std::vector<PwGroup> gr;
        int maxd = 0, mind = 1 << 30;
        const int len = std::min(8192, N - s);
        pw_groups(s, len, gr, 0, maxd, mind);
        if (maxd != mind || gr.size() > 64 || (gr.size() & (gr.size() - 1))) return false;   // perfect binary tree only
        for (size_t i = 0; i < gr.size(); ++i) {
            const PwGroup& q = gr[i];
            if (q.lenA < 8 || (q.lenB && q.lenB < 8)) return false;
            if (q.lenB && (q.lenA != 64 || q.offB != q.offA + q.lenA)) return false;
            if ((q.lenA + q.lenB + 7) / 8 > SM_NS || (q.lenA + q.lenB) / 8 < SM_FULL) return false;
            int* t = tab + 4 * (64 * c + i);
            t[0] = q.offA; t[1] = q.lenA; t[2] = q.lenB ? q.offB : N; t[3] = q.lenB;
            *maxrem = std::max(*maxrem, std::max(q.lenA % 8, q.lenB % 8));
        }
    }
    return true;
}

static size_t smoke_lds_bytes(int N) {
    return (size_t)(SM_PAD + N + 1 + SM_PAD + SM_THREADS) * 8 + (3 * SM_GROUPS + 48) * 8 +
           align_up((size_t)N + 1 + SM_PAD, 16);
}

struct SmokeWs {
    double* v; double* x; float* d; int* grp; unsigned char* cf; unsigned char* vm; unsigned char* bk; int* rect;
    size_t bytes;
};
static SmokeWs carve(void* ws, int n, int B) {
    const int n1 = n + 1, N = n * n;
    size_t o = 0;
    auto take = [&](size_t b) { size_t r = o; o = align_up(o + b, 256); return r; };
    SmokeWs w;
    char* base = (char*)ws;
    w.v = (double*)(base + take((size_t)B * n1 * n1 * 2 * 8));
    w.x = (double*)(base + take((size_t)B * (N + 1 + SM_THREADS) * 8));
    w.d = (float*)(base + take((size_t)B * 4 * N * 4));
    w.grp = (int*)(base + take(SM_GROUPS * 4 * 4));
    w.cf = (unsigned char*)(base + take((size_t)N + 1));
    w.vm = (unsigned char*)(base + take((size_t)n1 * n1));
    w.bk = (unsigned char*)(base + take((size_t)n1 * n1));
    w.rect = (int*)(base + take(8 * 4 * 4));
    w.bytes = o;
    return w;
}

static int smoke_prepare(const dpc_smoke_domain* dom, int B, void* ws, size_t ws_bytes, hipStream_t s, SmokeWs& w,
                         int* maxrem) {
    DPC_REQUIRE(dom && dom->fluid_d && dom->active_d, "smoke: null domain");
    DPC_REQUIRE(dom->n_buckets >= 0 && dom->n_buckets <= 7, "smoke: at most 7 buckets");
    DPC_REQUIRE(dom->n + 1 == 128, "smoke: the accounting tree is laid out for a 128 x 128 padded grid (n = 127)");
    w = carve(ws, dom->n, B);
    DPC_REQUIRE(ws && ws_bytes >= w.bytes, "smoke: workspace too small");
    int tab[SM_GROUPS * 4];
    if (!build_group_table(dom->n * dom->n, tab, maxrem))
        return fail(DPC_ERR_UNSUPPORTED, "smoke: numpy pairwise tree of n*n does not map onto 2 x 64 groups");
    DPC_HIP(hipMemcpyAsync(w.grp, tab, sizeof(tab), hipMemcpyHostToDevice, s));
    DPC_HIP(hipMemcpyAsync(w.rect, dom->bucket_rect, sizeof(int) * 4 * 8, hipMemcpyHostToDevice, s));
    DPC_HIP(hipStreamSynchronize(s));      // tab / dom are host stack memory
    const int n1 = dom->n + 1;
    hipLaunchKernelGGL(smoke_tables_kernel, dim3(cdiv(n1 * n1, 256)), dim3(256), 0, s, (const signed char*)dom->fluid_d,
                       (const signed char*)dom->active_d, dom->n, w.cf, w.vm, w.bk, dom->n_buckets, w.rect);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

static int smoke_launch(SmokeParams& P, int B, hipStream_t s) {
    const size_t lds = smoke_lds_bytes(P.N);
    DPC_REQUIRE(lds <= 160 * 1024, "smoke: grid too large for the 160 KB LDS");
    static DeviceOnce once;
    if (!once) {
        DPC_HIP(hipFuncSetAttribute((const void*)smoke_rollout_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DPC_HIP(hipFuncSetAttribute((const void*)smoke_cg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once = true;
    }
    if (P.mode == 1) hipLaunchKernelGGL(smoke_cg_kernel, dim3(B), dim3(SM_THREADS), lds, s, P);
    else hipLaunchKernelGGL(smoke_rollout_kernel, dim3(B), dim3(SM_THREADS), lds, s, P);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc

using namespace dpc;

extern "C" {

size_t dpc_smoke_workspace_bytes(int n, int B) {
    if (n < 8 || B < 1) return 0;
    return carve(nullptr, n, B).bytes;
}

int dpc_smoke_rollout(const dpc_smoke_domain* dom, const float* init_velocity, int64_t velocity_batch_stride,
                      const float* dens0, const float* c1, const float* c2, int B, int nx, int nt, int num_t, double dt,
                      double accuracy, int max_cg_iter, const dpc_smoke_out* out, void* ws, size_t ws_bytes,
                      dpc_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DPC_REQUIRE(init_velocity && dens0 && c1 && c2 && out, "smoke_rollout: null pointer");
    DPC_REQUIRE(B >= 1 && nx >= 1 && nt >= 1 && num_t >= 1, "smoke_rollout: sizes");
    DPC_REQUIRE(dom && 128 % nx == 0 && num_t % nt == 0, "smoke_rollout: nx must divide 128 and nt must divide num_t");
    DPC_REQUIRE(out->frame_stride >= 1 && out->space_stride >= 1 && 128 % out->space_stride == 0,
                "smoke_rollout: output strides");
    DPC_REQUIRE(dom->target_bucket >= 0 && dom->target_bucket < dom->n_buckets, "smoke_rollout: target bucket");
    SmokeWs w;
    int maxrem = 0;
    int rc = smoke_prepare(dom, B, ws, ws_bytes, s, w, &maxrem);
    if (rc) return rc;
    SmokeParams P{};
    P.n = dom->n; P.n1 = dom->n + 1; P.N = dom->n * dom->n; P.rim = dom->rim; P.nb = dom->n_buckets;
    P.target = dom->target_bucket; P.maxrem = maxrem;
    P.B = B; P.num_t = num_t; P.nt = nt; P.nx = nx; P.ti = num_t / nt; P.si = 128 / nx;
    P.max_it = max_cg_iter; P.dens_f32 = out->density_f32; P.ofs = out->frame_stride; P.oss = out->space_stride; P.mode = 0;
    P.accuracy = accuracy; P.dt = dt;
    P.vel0 = init_velocity; P.vel0_bstride = velocity_batch_stride; P.dens0 = dens0; P.c1 = c1; P.c2 = c2;
    P.densitys = out->densitys; P.zero_densitys = out->zero_densitys; P.velocitys = out->velocitys;
    P.smoke_out = out->smoke_out; P.cg_iters = out->cg_iters;
    P.v = w.v; P.x = w.x; P.d = w.d; P.grp = w.grp; P.cf = w.cf; P.vm = w.vm; P.bk = w.bk;
    // algorithmic traffic if the solver state streamed from HBM (it does not: LDS/VGPR/L2 resident), SURVEY 8(d)
    ProfScope prof(PROF_SMOKE_EVAL, 0, (double)B * (num_t - 1) * 500.0 * 5 * P.N * 16.0, s);
    return smoke_launch(P, B, s);
}

int dpc_smoke_pressure_solve(const dpc_smoke_domain* dom, double* div_to_pressure, int B, double accuracy,
                             int max_cg_iter, int* iterations, void* ws, size_t ws_bytes, dpc_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DPC_REQUIRE(div_to_pressure && B >= 1, "smoke_pressure_solve: arguments");
    SmokeWs w;
    int maxrem = 0;
    int rc = smoke_prepare(dom, B, ws, ws_bytes, s, w, &maxrem);
    if (rc) return rc;
    const int N = dom->n * dom->n;
    DPC_HIP(hipMemcpy2DAsync(w.x, (size_t)(N + 1 + SM_THREADS) * 8, div_to_pressure, (size_t)N * 8, (size_t)N * 8, B,
                             hipMemcpyDeviceToDevice, s));
    SmokeParams P{};
    P.n = dom->n; P.n1 = dom->n + 1; P.N = N; P.maxrem = maxrem; P.B = B; P.mode = 1; P.max_it = max_cg_iter;
    P.accuracy = accuracy; P.cg_iters = iterations; P.ofs = 1; P.oss = 1;
    P.v = w.v; P.x = w.x; P.d = w.d; P.grp = w.grp; P.cf = w.cf; P.vm = w.vm; P.bk = w.bk;
    rc = smoke_launch(P, B, s);
    if (rc) return rc;
    DPC_HIP(hipMemcpy2DAsync(div_to_pressure, (size_t)N * 8, w.x, (size_t)(N + 1 + SM_THREADS) * 8, (size_t)N * 8, B,
                             hipMemcpyDeviceToDevice, s));
    return DPC_OK;
}

int dpc_smoke_advect(const double* velocity, const float* density, float* out, int n, double dt, dpc_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    DPC_REQUIRE(velocity && density && out && n >= 2, "smoke_advect: arguments");
    hipLaunchKernelGGL(smoke_advect_kernel, dim3(cdiv(n * n, 256)), dim3(256), 0, s, velocity, density, out, n, dt);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int dpc_smoke_domain_tables(const dpc_smoke_domain* dom, unsigned char* cf_out, unsigned char* vmask_out,
                            unsigned char* bucket_out, void* ws, size_t ws_bytes, dpc_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    SmokeWs w;
    int maxrem = 0;
    int rc = smoke_prepare(dom, 1, ws, ws_bytes, s, w, &maxrem);
    if (rc) return rc;
    const int n1 = dom->n + 1, N = dom->n * dom->n;
    if (cf_out) DPC_HIP(hipMemcpyAsync(cf_out, w.cf, N, hipMemcpyDeviceToDevice, s));
    if (vmask_out) DPC_HIP(hipMemcpyAsync(vmask_out, w.vm, (size_t)n1 * n1, hipMemcpyDeviceToDevice, s));
    if (bucket_out) DPC_HIP(hipMemcpyAsync(bucket_out, w.bk, (size_t)n1 * n1, hipMemcpyDeviceToDevice, s));
    return DPC_OK;
}

}  // extern "C"
