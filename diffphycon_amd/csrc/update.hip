// Guidance-gradient accumulate + DDPM/DDIM update, one streaming pass (HBM-bound: 26 floats moved per
// 6-channel cell).  Arithmetic follows the reference op-for-op with explicit
// round-to-nearest mul/add (no FMA contraction) so the result is bit-identical to the fp32 CPU oracle:
//   x0  = c1*x - c2*eps_j                               diffusion_2d_smoke.py:620   (clip if DDIM :621)
//   g   = d/d(x0*R) [ -sum_b mean_hw (x0 R)[b,-1,-1] + w_e sum_b mean (x0 R)[b,:,3:5]^2 ]   inference_2d_smoke.py:35-42
//   eps = eps_j + (rho*g + (gamma-1)*pad(eps_w))        :630,638
//   x0  = c1*x - c2*eps                                 :640
//   DDPM: x0.clamp_(-1,1); x' = (m1*x0 + m2*x) + sigma*z          :663, 602-605, 685
//   DDIM: x0 clip; eps = (c1*x - x0)/c2; x' = x0*sqrt(a') + c*eps + sigma*z    :641-643, 771-773
//   x'[:,0,0] = init                                    :720 / :775
#include "common.h"

namespace dpc {

__device__ __forceinline__ float clamp1(float v) { return fminf(fmaxf(v, -1.0f), 1.0f); }

template <int VEC>
__global__ __launch_bounds__(256) void ddpm_update_smoke_kernel(
    const float* __restrict__ x, const float* __restrict__ eps_j, const float* __restrict__ eps_w,
    const float* __restrict__ z, const float* __restrict__ init, const float* __restrict__ rescaler,
    float* __restrict__ x_next, float* __restrict__ x0_out, dpc_step_coef k, int B, int F, int C, int H, int W) {
    const long long HW = (long long)H * W;
    const long long total = (long long)B * F * C * HW / VEC;
    const float g_obj = -1.0f / (float)(H * W);
    const float e_den = (float)((long long)F * 2 * H * W);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long e0 = i * VEC;
        const long long hw = e0 % HW;
        long long r = e0 / HW;
        const int c = (int)(r % C);
        r /= C;
        const int f = (int)(r % F);
        const int b = (int)(r / F);
        float xv[VEC], ej[VEC], ew[VEC], zv[VEC], out[VEC], x0v[VEC];
        if (VEC == 4) {
            *reinterpret_cast<f32x4*>(xv) = *reinterpret_cast<const f32x4*>(x + e0);
            *reinterpret_cast<f32x4*>(ej) = *reinterpret_cast<const f32x4*>(eps_j + e0);
        } else {
            xv[0] = x[e0];
            ej[0] = eps_j[e0];
        }
        const bool wch = (c == 3 || c == 4);
        if (wch) {
            const long long ew0 = (((long long)b * F + f) * 2 + (c - 3)) * HW + hw;
            if (VEC == 4) *reinterpret_cast<f32x4*>(ew) = *reinterpret_cast<const f32x4*>(eps_w + ew0);
            else ew[0] = eps_w[ew0];
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) ew[v] = 0.f;
        }
        const bool has_z = (z != nullptr) && k.mode != 2;
        if (has_z) {
            if (VEC == 4) *reinterpret_cast<f32x4*>(zv) = *reinterpret_cast<const f32x4*>(z + e0);
            else zv[0] = z[e0];
        }
        const bool objective_cell = (f == F - 1) && (c == C - 1);
        const float rc = rescaler[c];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float x0 = __fsub_rn(__fmul_rn(k.sqrt_recip_ac, xv[v]), __fmul_rn(k.sqrt_recipm1_ac, ej[v]));
            if (k.clip_x_start) x0 = clamp1(x0);
            float g = objective_cell ? g_obj : 0.f;
            if (wch && k.w_energy != 0.f) {
                const float xr = __fmul_rn(x0, rc);
                g = __fadd_rn(g, __fdiv_rn(__fmul_rn(__fmul_rn(k.w_energy, 2.0f), xr), e_den));
            }
            const float grad_final = __fadd_rn(__fmul_rn(k.guide_scale, g), __fmul_rn(k.w_scale, ew[v]));
            float eps = __fadd_rn(ej[v], grad_final);
            x0 = __fsub_rn(__fmul_rn(k.sqrt_recip_ac, xv[v]), __fmul_rn(k.sqrt_recipm1_ac, eps));
            if (k.clip_x_start) x0 = clamp1(x0);
            float xn;
            if (k.mode == 0) {
                x0 = clamp1(x0);
                const float mean = __fadd_rn(__fmul_rn(k.mean_coef1, x0), __fmul_rn(k.mean_coef2, xv[v]));
                xn = has_z ? __fadd_rn(mean, __fmul_rn(k.sigma, zv[v])) : mean;
            } else if (k.mode == 1) {
                if (k.clip_x_start)
                    eps = __fdiv_rn(__fsub_rn(__fmul_rn(k.sqrt_recip_ac, xv[v]), x0), k.sqrt_recipm1_ac);
                xn = __fadd_rn(__fmul_rn(x0, k.mean_coef1), __fmul_rn(k.mean_coef2, eps));
                if (has_z) xn = __fadd_rn(xn, __fmul_rn(k.sigma, zv[v]));
            } else {
                xn = x0;
            }
            x0v[v] = x0;
            out[v] = xn;
        }
        if (f == 0 && c == 0 && k.mode != 2) {
            const long long i0 = (long long)b * HW + hw;
#pragma unroll
            for (int v = 0; v < VEC; ++v) out[v] = init[i0 + v];
        }
        if (VEC == 4) {
            *reinterpret_cast<f32x4*>(x_next + e0) = *reinterpret_cast<f32x4*>(out);
            if (x0_out) *reinterpret_cast<f32x4*>(x0_out + e0) = *reinterpret_cast<f32x4*>(x0v);
        } else {
            x_next[e0] = out[0];
            if (x0_out) x0_out[e0] = x0v[0];
        }
    }
}

int launch_ddpm_update_smoke(const float* x, const float* eps_j, const float* eps_w, const float* z,
                             const float* init, const float* rescaler, float* x_next, float* x0_out,
                             const dpc_step_coef& c, int B, int F, int C, int H, int W, hipStream_t s) {
    DPC_REQUIRE(C >= 5, "ddpm_update_smoke: needs >= 5 channels (controls on 3:5)");
    DPC_REQUIRE(c.mode >= 0 && c.mode <= 2, "ddpm_update_smoke: mode");
    DPC_REQUIRE(!(z == nullptr && c.sigma != 0.f && c.mode != 2), "ddpm_update_smoke: z is null but sigma != 0");
    const long long total = (long long)B * F * C * H * W;
    if (total == 0) return DPC_OK;
    const bool vec = ((long long)H * W) % 4 == 0;
    // algorithmic traffic: x, eps_j, z, x_next on C channels + eps_w on 2 channels (+ x0_out)  => 26 floats / 6-ch cell
    ProfScope prof(PROF_UPDATE, 0, 4.0 * ((double)total * (3 + (z ? 1 : 0) + (x0_out ? 1 : 0)) + (double)total / C * 2), s);
    const long long work = vec ? total / 4 : total;
    const int grid = (int)std::min<long long>((work + 255) / 256, 256 * 8);
    if (vec)
        hipLaunchKernelGGL(ddpm_update_smoke_kernel<4>, dim3(grid), dim3(256), 0, s, x, eps_j, eps_w, z, init, rescaler,
                           x_next, x0_out, c, B, F, C, H, W);
    else
        hipLaunchKernelGGL(ddpm_update_smoke_kernel<1>, dim3(grid), dim3(256), 0, s, x, eps_j, eps_w, z, init, rescaler,
                           x_next, x0_out, c, B, F, C, H, W);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------ Burgers sampler
// set_condition + zero-fill before every step (diffusion_1d_burgers.py:539-553) and the prior model's input (:399-400)
__global__ __launch_bounds__(256) void burgers_prepare_kernel(float* __restrict__ img, float* __restrict__ xw,
                                                              const float* __restrict__ u0, const float* __restrict__ uT,
                                                              int B, int nt, int nx, int cond_idx, int set_zero) {
    const long long total = (long long)B * 2 * nt * nx;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % nx);
        long long r = i / nx;
        const int row = (int)(r % nt);
        r /= nt;
        const int ch = (int)(r % 2);
        const int b = (int)(r / 2);
        float v = img[i];
        if (ch == 0) {
            if (u0 && row == 0) v = u0[(long long)b * nx + xx];
            if (uT && row == cond_idx) v = uT[(long long)b * nx + xx];
            if (set_zero && xx >= nx / 4 && xx < (nx * 3) / 4) v = 0.f;
            img[i] = v;
            if (xw) xw[i] = (row >= 1 && row < cond_idx) ? 0.f : v;
        } else if (xw) {
            xw[i] = v;
        }
    }
}

int launch_burgers_prepare(float* img, float* x_w, const float* u0, const float* uT, int B, int nt, int nx, int cond_idx,
                           int set_zero, hipStream_t s) {
    const long long total = (long long)B * 2 * nt * nx;
    if (total == 0) return DPC_OK;
    ProfScope prof(PROF_UPDATE, 0, 4.0 * (double)total * (x_w ? 2.5 : 1.5), s);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(burgers_prepare_kernel, dim3(grid), dim3(256), 0, s, img, x_w, u0, uT, B, nt, nx, cond_idx, set_zero);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// model_predictions (after the denoiser calls) + p_mean_variance + p_sample, one streaming pass:
//   eps = eps_uw - w_coef*eps_w  (eps_w channel 0 := 0)  | (eps_uw - w_coef*eps_w)/beta          :402-409
//   x0  = a*x - b*eps                                                                               :426
//   eps += dJ/dx(x0) * eta_J ; x0 = a*x - b*eps                                                     :431-434
//   x0.clamp_(-1,1); x' = (m1*x0 + m2*x) + sigma*z                                                  :457-469
__global__ __launch_bounds__(256) void ddpm_update_burgers_kernel(
    const float* __restrict__ x, const float* __restrict__ e_uw, const float* __restrict__ e_w,
    const float* __restrict__ z, const float* __restrict__ ut, float* __restrict__ x_next, float* __restrict__ x0_out,
    float* __restrict__ eps_out, dpc_burgers_coef k, int B, int nt, int nx) {
    const long long total = (long long)B * 2 * nt * nx;
    const float cu = __fdiv_rn(__fmul_rn(2.0f, k.wu), (float)((long long)k.guidance_batch * nx));
    const float cf = __fdiv_rn(__fmul_rn(2.0f, k.wf), (float)k.guidance_batch);
    const float cr = __fmul_rn(2.0f, k.wreg);
    const bool guided = (k.wu != 0.f) || (k.wf != 0.f) || (k.wreg != 0.f);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % nx);
        long long r = i / nx;
        const int row = (int)(r % nt);
        r /= nt;
        const int ch = (int)(r % 2);
        const int b = (int)(r / 2);
        auto mixed_eps = [&](long long j) -> float {
            float e = e_uw[j];
            if (k.two_models) {
                const float w = (ch == 0) ? 0.f : e_w[j];
                e = __fsub_rn(e, __fmul_rn(k.w_coef, w));
                if (k.normalize_beta) e = __fdiv_rn(e, k.prior_beta);
            }
            return e;
        };
        auto x0_at = [&](long long j) -> float {
            return __fsub_rn(__fmul_rn(k.sqrt_recip_ac, x[j]), __fmul_rn(k.sqrt_recipm1_ac, mixed_eps(j)));
        };
        const float xv = x[i];
        float eps = mixed_eps(i);
        float x0 = __fsub_rn(__fmul_rn(k.sqrt_recip_ac, xv), __fmul_rn(k.sqrt_recipm1_ac, eps));
        if (guided) {
            float g = 0.f;
            if (ch == 0 && row <= k.cond_idx) {
                const bool obs = !(k.partially_observed && xx >= nx / 4 && xx < (nx * 3) / 4);
                if (k.wu != 0.f && obs && (row == 0 || row == k.cond_idx)) {
                    const float tgt = ut[((long long)b * 2 + (row == 0 ? 0 : 1)) * nx + xx];
                    g = __fadd_rn(g, __fmul_rn(cu, __fsub_rn(x0, tgt)));
                }
                if (k.wreg != 0.f) {
                    if (row >= 1) g = __fadd_rn(g, __fmul_rn(cr, __fsub_rn(x0, x0_at(i - nx))));
                    if (row < k.cond_idx) g = __fsub_rn(g, __fmul_rn(cr, __fsub_rn(x0_at(i + nx), x0)));
                }
            } else if (ch == 1 && row < k.cond_idx) {
                g = __fmul_rn(cf, x0);
            }
            eps = __fadd_rn(eps, __fmul_rn(g, k.eta_J));
            x0 = __fsub_rn(__fmul_rn(k.sqrt_recip_ac, xv), __fmul_rn(k.sqrt_recipm1_ac, eps));
        }
        if (k.clip_denoised) x0 = clamp1(x0);
        const float mean = __fadd_rn(__fmul_rn(k.mean_coef1, x0), __fmul_rn(k.mean_coef2, xv));
        x_next[i] = z ? __fadd_rn(mean, __fmul_rn(k.sigma, z[i])) : mean;
        if (x0_out) x0_out[i] = x0;
        if (eps_out) eps_out[i] = eps;
    }
}

int launch_ddpm_update_burgers(const float* x, const float* eps_uw, const float* eps_w, const float* z,
                               const float* u_target, float* x_next, float* x0_out, float* eps_out,
                               const dpc_burgers_coef& c, int B, int nt, int nx, hipStream_t s) {
    DPC_REQUIRE(!(c.two_models && !eps_w), "ddpm_update_burgers: two_models needs eps_w");
    DPC_REQUIRE(!(c.wu != 0.f && !u_target), "ddpm_update_burgers: wu != 0 needs u_target");
    DPC_REQUIRE(c.cond_idx >= 1 && c.cond_idx < nt, "ddpm_update_burgers: cond_idx");
    DPC_REQUIRE(c.guidance_batch >= 1, "ddpm_update_burgers: guidance_batch");
    DPC_REQUIRE(x_next != x || c.wreg == 0.f, "ddpm_update_burgers: in-place update is not allowed with wreg != 0 (row neighbours)");
    const long long total = (long long)B * 2 * nt * nx;
    if (total == 0) return DPC_OK;
    ProfScope prof(PROF_UPDATE, 0, 4.0 * (double)total * (3.5 + (z ? 1 : 0) + (x0_out ? 1 : 0) + (eps_out ? 1 : 0)), s);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(ddpm_update_burgers_kernel, dim3(grid), dim3(256), 0, s, x, eps_uw, eps_w, z, u_target, x_next,
                       x0_out, eps_out, c, B, nt, nx);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------ jellyfish sampler
__global__ __launch_bounds__(256) void ddpm_update_jelly_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                                const float* __restrict__ eps_g, const float* __restrict__ z,
                                                                float* __restrict__ pred, float* __restrict__ x0_out,
                                                                dpc_jelly_coef k, int B, int F, int Cx, int ns, long long HW) {
    const int Cd = ns + 1;
    const long long total = (long long)B * F * Cd * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long hw = i % HW;
        long long r = i / HW;
        const int c = (int)(r % Cd);
        const long long bf = r / Cd;
        const int cx = (c < ns) ? c : Cx - 1;
        const float xv = x[(bf * Cx + cx) * HW + hw];
        float x0 = __fsub_rn(__fmul_rn(k.sqrt_recip_ac, xv), __fmul_rn(k.sqrt_recipm1_ac, eps[i]));
        float out;
        if (k.mode == 0) {
            if (k.clip_denoised) x0 = clamp1(x0);
            const float mean = __fadd_rn(__fmul_rn(k.mean_coef1, x0), __fmul_rn(k.mean_coef2, xv));
            out = z ? __fadd_rn(mean, __fmul_rn(k.sigma, z[i])) : mean;
        } else if (k.mode == 1) {
            const float e = eps_g ? eps_g[i] : eps[i];
            out = __fadd_rn(__fmul_rn(x0, k.mean_coef1), __fmul_rn(k.mean_coef2, e));
            if (z) out = __fadd_rn(out, __fmul_rn(k.sigma, z[i]));
        } else {
            out = x0;
        }
        pred[i] = out;
        if (x0_out) x0_out[i] = x0;
    }
}

__global__ __launch_bounds__(256) void jelly_guidance_kernel(float* __restrict__ io, const float* __restrict__ g,
                                                             const float* __restrict__ eps_w, float eta_J, float eta_w,
                                                             int pad_w, float sign, long long BF, int Cd, long long HW) {
    const long long total = BF * Cd * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long hw = i % HW;
        long long r = i / HW;
        const int c = (int)(r % Cd);
        const long long bf = r / Cd;
        float w = 0.f;
        if (eps_w && (!pad_w || c == Cd - 1)) w = eps_w[bf * HW + hw];
        const float gf = __fsub_rn(__fmul_rn(eta_J, g ? g[i] : 0.f), __fmul_rn(eta_w, w));
        io[i] = (sign < 0.f) ? __fsub_rn(io[i], gf) : __fadd_rn(io[i], gf);
    }
}

int launch_ddpm_update_jelly(const float* x, const float* eps, const float* eps_g, const float* z, float* pred,
                             float* x0_out, const dpc_jelly_coef& c, int B, int F, int Cx, int ns, int H, int W,
                             hipStream_t s) {
    DPC_REQUIRE(ns >= 1 && ns < Cx && c.mode >= 0 && c.mode <= 2, "ddpm_update_jelly: bad channel split / mode");
    const long long total = (long long)B * F * (ns + 1) * H * W;
    if (total == 0) return DPC_OK;
    ProfScope prof(PROF_UPDATE, 0, 4.0 * (double)total * (3 + (z ? 1 : 0) + (x0_out ? 1 : 0) + (eps_g ? 1 : 0)), s);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(ddpm_update_jelly_kernel, dim3(grid), dim3(256), 0, s, x, eps, eps_g, z, pred, x0_out, c, B, F, Cx, ns,
                       (long long)H * W);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

int launch_jelly_guidance(float* io, const float* g, const float* eps_w, float eta_J, float eta_w, int pad_w, float sign,
                          int B, int F, int Cd, int H, int W, hipStream_t s) {
    const long long total = (long long)B * F * Cd * H * W;
    if (total == 0) return DPC_OK;
    ProfScope prof(PROF_UPDATE, 0, 4.0 * (double)total * (g ? 3.25 : 2.25), s);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(jelly_guidance_kernel, dim3(grid), dim3(256), 0, s, io, g, eps_w, eta_J, eta_w, pad_w, sign,
                       (long long)B * F, Cd, (long long)H * W);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

// ------------------------------------------------------------------ counter-based normal noise
// Philox4x32-10 keyed by seed; counter = (element/4, draw, trajectory lo, trajectory hi).
// A trajectory's stream depends only on (seed, global trajectory index, draw): 1-GPU and 8-GPU runs of the
// same global batch draw identical noise.  Box-Muller on the four 32-bit outputs.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, int B, long long per_traj,
                                                            uint64_t seed, long long traj0, long long draw) {
    const long long q = (per_traj + 3) / 4;
    const long long total = (long long)B * q;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / q, e4 = i % q;
        const unsigned long long traj = (unsigned long long)(traj0 + b);
        uint32_t c[4] = {(uint32_t)e4, (uint32_t)draw, (uint32_t)traj, (uint32_t)(traj >> 32) ^ (uint32_t)(e4 >> 32 << 16)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float n[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
            const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            n[2 * h] = rad * cs;
            n[2 * h + 1] = rad * sn;
        }
        const long long o = b * per_traj + e4 * 4;
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (e4 * 4 + v < per_traj) out[o + v] = n[v];
    }
}

int launch_philox_normal(float* out, int B, long long per_traj, uint64_t seed, long long traj0, long long draw,
                         hipStream_t s) {
    const long long total = (long long)B * ((per_traj + 3) / 4);
    if (total == 0) return DPC_OK;
    ProfScope prof(PROF_PHILOX, 0, 4.0 * (double)B * per_traj, s);
    const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid), dim3(256), 0, s, out, B, per_traj, seed, traj0, draw);
    DPC_LAUNCH_CHECK();
    return DPC_OK;
}

}  // namespace dpc
