// What does an instruction of a co-resident wave cost the matrix pipe on gfx950?  (DESIGN.md 6.1b measured "~6 matrix-pipe cycles
// per loader instruction" inside conv3w; this isolates it per instruction type.)
//
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) issue a fixed number of independent v_mfma_f32_32x32x16_f16 (4
// accumulators round-robin, no dependency stall: 8 passes = 32 cycles each when alone); waves 4-7 (the second wave of each SIMD) run
// a filler loop of ONE instruction type until the MFMA waves are done, counting their iterations.  Reported per filler type:
//   cycles per MFMA of the MFMA waves (alone: ~32), filler instructions retired per MFMA, and
//   cost = (cycles per MFMA - alone) / fillers per MFMA  = matrix-pipe cycles lost per co-resident filler instruction.
// Mode "same": the filler instructions are issued by the MFMA wave itself, K per MFMA (independent of the MFMA's registers).
//   hipcc --offload-arch=gfx950 -O3 -o tools/coissue/coissue tools/coissue/coissue.hip && gpurun -- tools/coissue/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Filler { NONE = 0, FMA, FMA_DEP, CVT_PK, FMA_MIX, PK_FMA, EXP, DS_WRITE, DS_READ, MOV, MUL, MED3, NFILL };
static const char* NAMES[NFILL] = {"(none)", "v_fma_f32 x8 independent", "v_fma_f32 dependent chain", "v_cvt_pk_f16_f32", "v_fma_mixlo_f16",
                                   "v_pk_fma_f32", "v_exp_f32", "ds_write_b64", "ds_read_b128", "v_mov_b32", "v_mul_f32", "v_med3_f32"};

constexpr int UNROLL = 16;          // filler instructions per block
constexpr int BLOCKS = 16;          // blocks per look at the `done` flag (an LDS read + wait: ~100 cycles)

template <int F>
__device__ __forceinline__ void filler_block(float (&r)[8], unsigned (&u)[4], unsigned char* lds, int lane) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        if constexpr (F == MUL) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[i & 7]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
        if constexpr (F == MED3) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(r[i & 7]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]), "v"(r[(i + 3) & 7]));
        if constexpr (F == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 7]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
        if constexpr (F == FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[0]) : "v"(r[1]));
        if constexpr (F == CVT_PK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(r[i & 7]), "v"(r[(i + 1) & 7]));
        if constexpr (F == FMA_MIX) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "+v"(u[i & 3]) : "v"(r[i & 7]), "v"(u[(i + 1) & 3]));
        if constexpr (F == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&r[2 * (i & 3)])) : "v"(*reinterpret_cast<double*>(&r[2 * ((i + 1) & 3)])));
        if constexpr (F == EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(r[i & 7]) : "v"(r[(i + 1) & 7]));
        if constexpr (F == DS_WRITE) asm volatile("ds_write_b64 %0, %1" :: "v"(lane * 8 + (i & 3) * 512), "v"(*reinterpret_cast<double*>(&r[0])) : "memory");
        if constexpr (F == DS_READ) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<f32x4*>(&r[4 * (i & 1)])) : "v"(lane * 16 + (i & 3) * 1024) : "memory");
        if constexpr (F == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i & 3]) : "v"(u[(i + 1) & 3]));
    }
    if constexpr (F == DS_WRITE || F == DS_READ) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int F, int K>
__device__ __forceinline__ void same_wave(float (&r)[8], unsigned (&u)[4], int lane) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if constexpr (F == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i & 7]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
        if constexpr (F == MUL) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[i & 7]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]));
        if constexpr (F == MED3) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(r[i & 7]) : "v"(r[(i + 1) & 7]), "v"(r[(i + 2) & 7]), "v"(r[(i + 3) & 7]));
        if constexpr (F == CVT_PK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i & 3]) : "v"(r[i & 7]), "v"(r[(i + 1) & 7]));
        if constexpr (F == FMA_MIX) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "+v"(u[i & 3]) : "v"(r[i & 7]), "v"(u[(i + 1) & 3]));
        if constexpr (F == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&r[2 * (i & 3)])) : "v"(*reinterpret_cast<double*>(&r[2 * ((i + 1) & 3)])));
        if constexpr (F == EXP) asm volatile("v_exp_f32 %0, %1" : "=v"(r[i & 7]) : "v"(r[(i + 1) & 7]));
        if constexpr (F == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i & 3]) : "v"(u[(i + 1) & 3]));
        if constexpr (F == DS_WRITE) asm volatile("ds_write_b64 %0, %1" :: "v"(lane * 8 + (i & 3) * 512), "v"(*reinterpret_cast<double*>(&r[0])) : "memory");
        if constexpr (F == DS_READ) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<f32x4*>(&r[4 * (i & 1)])) : "v"(lane * 16 + (i & 3) * 1024) : "memory");
    }
}

// out[block][0] = cycles of MFMA wave 0, [1] = filler blocks of wave 4
template <int F, int SAME_K>
__global__ __launch_bounds__(512, 1) void coissue_kernel(unsigned long long* out, int n_mfma) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    __shared__ volatile int done;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) done = 0;
    __syncthreads();
    float r[8];
    unsigned u[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = 1.0f + 1e-7f * (lane + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = lane + i;
    if (wave < 4) {
        f16x8 a, b;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        f32x16 acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < n_mfma; it += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
                if constexpr (SAME_K > 0) same_wave<F, SAME_K>(r, u, lane);
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) s += acc[k][0];
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) atomicAdd((int*)&done, 1);
        if (wave == 0 && lane == 0) out[blockIdx.x * 4 + 0] = t1 - t0;
        if (s == 123.456f) out[blockIdx.x * 4 + 3] = (unsigned long long)(r[0] + u[0]);       // (keep everything live)
    } else {
        unsigned long long iters = 0;
        if constexpr (F != NONE && SAME_K == 0) {
            while (done < 4) {
#pragma unroll 1
                for (int b = 0; b < BLOCKS; ++b) filler_block<F>(r, u, lds, lane);
                iters += BLOCKS;
            }
        }
        if (wave == 4 && lane == 0) out[blockIdx.x * 4 + 1] = iters;
        if (r[0] == 123.456f && u[0] == 77) out[blockIdx.x * 4 + 3] = 1;
    }
}

template <int F, int SAME_K>
static void run(const char* name, unsigned long long* d_out, int nblk, int n_mfma, double base) {
    hipLaunchKernelGGL((coissue_kernel<F, SAME_K>), dim3(nblk), dim3(512), 0, 0, d_out, n_mfma);
    hipLaunchKernelGGL((coissue_kernel<F, SAME_K>), dim3(nblk), dim3(512), 0, 0, d_out, n_mfma);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 4);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, it = 0;
    for (int b = 0; b < nblk; ++b) { cyc += (double)h[b * 4]; it += (double)h[b * 4 + 1]; }
    cyc /= nblk; it /= nblk;
    // s_memtime counts at 100 MHz on gfx9 (constant clock): report in "memtime ticks" and relative to the unloaded run
    const double per_mfma = cyc / n_mfma;
    const double fill_per_mfma = SAME_K > 0 ? SAME_K : it * UNROLL / n_mfma;
    if (base <= 0) printf("%-34s %s  ticks/MFMA %8.4f\n", name, SAME_K ? "same wave " : "other wave", per_mfma);
    else printf("%-34s %s  ticks/MFMA %8.4f (x%5.2f of alone)  fillers/MFMA %6.2f  cost %6.3f MFMA-times per filler = %5.2f cycles at 32 cycles per MFMA\n", name,
                SAME_K ? "same wave " : "other wave", per_mfma, per_mfma / base, fill_per_mfma, (per_mfma / base - 1.0) / fill_per_mfma,
                32.0 * (per_mfma / base - 1.0) / fill_per_mfma);
}

int main() {
    const int nblk = 256, n_mfma = 40000;
    unsigned long long* d_out;
    hipMalloc(&d_out, nblk * 4 * 8);
    hipMemset(d_out, 0, nblk * 4 * 8);
    // baseline
    hipLaunchKernelGGL((coissue_kernel<NONE, 0>), dim3(nblk), dim3(512), 0, 0, d_out, n_mfma);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk * 4);
    hipLaunchKernelGGL((coissue_kernel<NONE, 0>), dim3(nblk), dim3(512), 0, 0, d_out, n_mfma);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double base = 0;
    for (int b = 0; b < nblk; ++b) base += (double)h[b * 4];
    base = base / nblk / n_mfma;
    printf("alone: %.4f memtime ticks per MFMA (one MFMA wave per SIMD, 256 CUs busy)\n", base);
    run<FMA, 0>(NAMES[FMA], d_out, nblk, n_mfma, base);
    run<FMA_DEP, 0>(NAMES[FMA_DEP], d_out, nblk, n_mfma, base);
    run<CVT_PK, 0>(NAMES[CVT_PK], d_out, nblk, n_mfma, base);
    run<FMA_MIX, 0>(NAMES[FMA_MIX], d_out, nblk, n_mfma, base);
    run<PK_FMA, 0>(NAMES[PK_FMA], d_out, nblk, n_mfma, base);
    run<EXP, 0>(NAMES[EXP], d_out, nblk, n_mfma, base);
    run<MOV, 0>(NAMES[MOV], d_out, nblk, n_mfma, base);
    run<DS_WRITE, 0>(NAMES[DS_WRITE], d_out, nblk, n_mfma, base);
    run<DS_READ, 0>(NAMES[DS_READ], d_out, nblk, n_mfma, base);
    run<MUL, 0>(NAMES[MUL], d_out, nblk, n_mfma, base);
    run<MED3, 0>(NAMES[MED3], d_out, nblk, n_mfma, base);
    printf("--- fillers issued by the MFMA wave itself, K per MFMA\n");
#define SAME(F, NAME) run<F, 2>(NAME ", 2 per MFMA", d_out, nblk, n_mfma, base); run<F, 4>(NAME ", 4 per MFMA", d_out, nblk, n_mfma, base); \
                      run<F, 6>(NAME ", 6 per MFMA", d_out, nblk, n_mfma, base); run<F, 8>(NAME ", 8 per MFMA", d_out, nblk, n_mfma, base); \
                      run<F, 12>(NAME ", 12 per MFMA", d_out, nblk, n_mfma, base)
    SAME(FMA, "v_fma_f32");
    SAME(MUL, "v_mul_f32");
    SAME(MED3, "v_med3_f32");
    SAME(CVT_PK, "v_cvt_pk_f16_f32");
    SAME(FMA_MIX, "v_fma_mixlo_f16");
    SAME(PK_FMA, "v_pk_fma_f32");
    SAME(EXP, "v_exp_f32");
    SAME(MOV, "v_mov_b32");
    SAME(DS_READ, "ds_read_b128");
    SAME(DS_WRITE, "ds_write_b64");
    hipFree(d_out);
    return 0;
}
