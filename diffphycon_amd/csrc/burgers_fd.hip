// Burgers explicit finite-difference evaluator: one wavefront per trajectory, the whole state in VGPRs,
// all `steps` Euler steps in ONE launch (the reference issues ~8 torch ops per step: 80k launches).
// Reference: dataset/apps/generate_burgers.py:207-299 (burgers_numeric_solve_free), :95-110 (Diff_mat_1D).
//   per step:  u <- pad0(u[1:-1]);  transport = D1(u^2)/(2dx);  diffusion = nu*D2(u)/dx^2
//              u += dt*(-0.5*transport + diffusion + f[k]),  k advances when j % record == 0,
//              snapshot when (j+1) % record == 0.
// Neighbour cells come from the adjacent lanes with wave shuffles; no LDS, no barriers.
// fp32, op order as the oracle (explicit mul/add, no contraction) => bit-identical to oracle/burgers.py.
#include <math.h>

#include "common.h"

namespace dpc {

struct BurgersCoef {
    float t_l, t_r;        // -1/(2dx), 1/(2dx)  (fp32 of the fp64 value)
    float d_l, d_c, d_r;   // nu/dx^2 * (1,-2,1)
    float dt;
    int steps, record;
};

template <int CPL>
__global__ __launch_bounds__(256) void burgers_fd_kernel(const float* __restrict__ u0, const float* __restrict__ f,
                                                         float* __restrict__ traj, int N, int nx, int num_t,
                                                         BurgersCoef k) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float u[CPL], fc[CPL];
    const int c0 = lane * CPL;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = c0 + i;
        u[i] = c < nx ? u0[(long long)n * nx + c] : 0.f;
        fc[i] = 0.f;
        if (c < nx) traj[((long long)n * (num_t + 1)) * nx + c] = u[i];
    }
    int f_idx = -1, rec = 0;
    for (int j = 0; j < k.steps; ++j) {
        if (j % k.record == 0) {
            ++f_idx;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = c0 + i;
                fc[i] = c < nx ? f[((long long)n * num_t + f_idx) * nx + c] : 0.f;
            }
        }
        float left = __shfl_up(u[CPL - 1], 1, 64);
        float right = __shfl_down(u[0], 1, 64);
        if (lane == 0) left = 0.f;
        if (lane == 63) right = 0.f;
        float un[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const float ul = i == 0 ? left : u[i - 1];
            const float ur = i == CPL - 1 ? right : u[i + 1];
            const float uc = u[i];
            const float transport = __fadd_rn(__fmul_rn(__fmul_rn(ul, ul), k.t_l), __fmul_rn(__fmul_rn(ur, ur), k.t_r));
            const float diffusion =
                __fadd_rn(__fadd_rn(__fmul_rn(ul, k.d_l), __fmul_rn(uc, k.d_c)), __fmul_rn(ur, k.d_r));
            const float rhs = __fadd_rn(__fadd_rn(__fmul_rn(-0.5f, transport), diffusion), fc[i]);
            un[i] = __fadd_rn(uc, __fmul_rn(k.dt, rhs));
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) u[i] = (c0 + i) < nx ? un[i] : 0.f;
        if ((j + 1) % k.record == 0 && rec < num_t) {
            ++rec;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int c = c0 + i;
                if (c < nx) traj[((long long)n * (num_t + 1) + rec) * nx + c] = u[i];
            }
        }
    }
}

int launch_burgers_fd(const float* u0, const float* f, float* traj, int N, int nx, int num_t, double visc, double T,
                      double dt, hipStream_t s) {
    DPC_REQUIRE(nx >= 1 && nx <= 256, "burgers_fd: nx must be in 1..256");
    DPC_REQUIRE(num_t >= 1, "burgers_fd: num_t");
    if (N == 0) return DPC_OK;
    const double dx = 1.0 / (nx + 1);                     // generate_burgers.py:240
    BurgersCoef k;
    k.steps = (int)ceil(T / dt);                          // :243
    k.record = k.steps / num_t;                           // :251
    DPC_REQUIRE(k.record >= 1, "burgers_fd: steps < num_t");
    k.t_l = (float)(-1.0 / (2 * dx));                     // :265
    k.t_r = (float)(1.0 / (2 * dx));
    k.d_l = (float)(visc * 1.0 / (dx * dx));              // :267
    k.d_c = (float)(visc * -2.0 / (dx * dx));
    k.d_r = k.d_l;
    k.dt = (float)dt;
    const int grid = (N + 3) / 4;
    ProfScope prof(PROF_BURGERS, 0, 4.0 * (double)N * nx * (2 * num_t + 2), s);
    if (nx <= 64) hipLaunchKernelGGL(burgers_fd_kernel<1>, dim3(grid), dim3(256), 0, s, u0, f, traj, N, nx, num_t, k);
    else if (nx <= 128) hipLaunchKernelGGL(burgers_fd_kernel<2>, dim3(grid), dim3(256), 0, s, u0, f, traj, N, nx, num_t, k);
    else hipLaunchKernelGGL(burgers_fd_kernel<4>, dim3(grid), dim3(256), 0, s, u0, f, traj, N, nx, num_t, k);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}
}  // namespace dpc
