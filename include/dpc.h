/*
 * libdpc — C-ABI boundary of the MI355X-native DiffPhyCon guided-sampling hot path.
 *
 * The reference (AI4Science-WestlakeU/diffphycon) has no FFI layer: the path sits behind plain
 * Python call signatures.  The host side of this repo (diffphycon_amd/, Python) mirrors those
 * signatures and binds the entry points below with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.  Each entry cites the reference code it replaces.
 *
 * Conventions
 *   - every pointer named *_d / x / out / ws is a DEVICE pointer owned by the caller
 *     (PyTorch-ROCm's allocator stays the single owner of HBM); the library never frees it;
 *   - handles own only device copies of weights (re-packed for the kernels) and small tables;
 *   - every entry returns 0 on success or a negative dpc_status; dpc_last_error() gives the
 *     message (thread-local).  No exceptions, no exit() across the boundary;
 *   - kernels are launched on the hipStream_t passed in (void* here so that C callers need
 *     no HIP headers); no host synchronisation happens inside forward/update calls;
 *   - all tensors fp32 contiguous unless stated; index types int64 where the reference uses
 *     torch.long.
 */
#ifndef DPC_H
#define DPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dpc_stream_t; /* hipStream_t */

enum dpc_status {
    DPC_OK = 0,
    DPC_ERR_ARG = -1,        /* bad shape / null pointer / unsupported configuration */
    DPC_ERR_STATE = -2,      /* handle not finalised, missing weight, workspace too small */
    DPC_ERR_HIP = -3,        /* a HIP runtime call failed; message carries hipGetErrorString */
    DPC_ERR_UNSUPPORTED = -4
};

int dpc_version(void);
/* What the build measured on the linked gfx950 code objects (diffphycon_amd/build.py): "packed_fp32_insts=<n>;code_objects=<m>;flags=...".
 * The Python binding refuses a library with n != 0 unless DPC_ALLOW_PACKED_FP32=1 (packed fp32 VALU instructions return wrong lanes
 * when a second kernel is resident on the GPU).  The reference has no counterpart (a build-hygiene entry, not an operator). */
const char* dpc_build_info(void);
/* CU budget of the persistent ("one workgroup per CU") kernels -- 3x3x3 / (1,3,3) convolutions, fused temporal attention: 0 = every CU
 * of the device (default), else at most `cus` (rounded down to a multiple of 8, one share per XCD) workgroups per launch.  The smoke
 * entry script sets 192 while a batch's PDE rollouts (64 CUs for ~2 s: csrc/smoke_rollout.hip) run on a side stream under the next
 * batch's sampling (inference/inference_2d_smoke.py InferencePipeline.run; reference schedule: inference_2d_smoke.py:259-271 runs them
 * one after the other).  Results do not depend on the value (the tile -> workgroup map changes, no arithmetic does).
 * PROCESS-WIDE: one global read at launch-enqueue time by every handle, thread, stream and device of the process -- meant for the
 * one-rank-per-process deployment (one GPU, one pipeline); a process that drives several pipelines must serialise its use. */
int dpc_set_cu_budget(int cus);
const char* dpc_last_error(void);

/* Arithmetic mode of the GEMM-shaped op families ("conv" 3x3x3 convolutions, "igemm" implicit-GEMM ops, "attn" fused
 * attention blocks, "stem" 7x7x7 stem; "all" sets the four): how an fp32 product is evaluated --
 *   "f16x3" (default) 2-way fp16 operand split, 3 MFMAs per product, 22 significant operand bits;
 *   "x6"    exact 3-way bf16 split, 6 MFMAs per product;   "f32"  the native fp32 MFMA.
 * Accumulation and all tensors in HBM are fp32 in every mode.  The reference computes these ops in plain fp32 torch
 * (video_diffusion_pytorch_conv3d.py:189-230; TF32 on the GPUs it targets), so "x6"/"f32" are its exact-product modes.
 * The process-wide setting starts from the environment (DPC_{CONV,IGEMM,ATTN,STEM}_MODE) and is CAPTURED by a U-Net
 * handle when dpc_unet*_create runs: changing it later affects only handles created afterwards.
 * dpc_get_mode(family) returns the process-wide setting ("all"/NULL: "conv=..,igemm=..,attn=..,stem=.."), valid until the
 * next call on the same thread; dpc_unet3d_modes / dpc_unet2d_modes return what a handle captured. */
int dpc_set_mode(const char* family, const char* mode);
const char* dpc_get_mode(const char* family);
/* Which algorithm the f16x3 3x3x3 convolution (Block.proj, video_diffusion_pytorch_conv3d.py:189-204) runs for the shapes that
 * qualify: "winograd_f43_frames" (default since r06: minimal filtering F(4,3) along the frame axis, interpolation points (0, 1, -1, 1/2,
 * -2, inf), 1/2 of the matrix products of the direct form, same 22-bit split operands and fp32 accumulation; csrc/conv3w4.hip),
 * "winograd_f23_frames" (F(2,3), 2/3 of the products; csrc/conv3w.hip; DPC_DEBUG=1 DPC_CONV3W_F43=0) or "direct" (DPC_CONV3W=0).
 * Reported next to the arithmetic modes by bench.py: roofline.achieved counts ALGORITHMIC (direct-form) FLOP in all cases,
 * roofline.mfma_issue_frac the issued ones. */
const char* dpc_conv3d_algorithm(void);

/* Opt-in timing of every kernel launch with HIP events on the launch stream, aggregated per kernel class
 * (bench.py's roofline leg; no counterpart in the reference, whose only stopwatch is commented out at
 * inference/inference_1d_burgers.py:287-291).  flops/bytes are ALGORITHMIC totals of the timed launches. */
typedef struct {
    const char* name;
    int64_t launches;
    double total_ms;
    double flops;
    double bytes;
} dpc_profile_row;
int dpc_profile_begin(void);
/* As dpc_profile_begin, but only launches of the named classes (comma-separated dpc_profile_row::name values) are
   bracketed by events; NULL or "" = all classes.  Two events per launch cost ~1.3 us of stream time each: bench.py times
   its K steps with only the dominant class instrumented (4 % -> 0.4 % overhead). */
int dpc_profile_begin_classes(const char* class_names);
int dpc_profile_end(dpc_profile_row* rows, int max_rows, int* n_rows);
/* Device self-test: the f16x3 operand split (hi = fp16(x), lo = fp16(x - hi)) of n floats exactly as the fused attention kernels
 * evaluate it, i.e. with MODE.FP16_OVFL set so that an overflowing conversion SATURATES at +-65504 instead of producing inf
 * (csrc/f16x3.h: hw_sat_enable -- those kernels dropped their software clamp in r04).  out[2 i] = hi, out[2 i + 1] = lo as floats. */
int dpc_selftest_fp16_clamp(const float* x, float* out, int n, dpc_stream_t stream);

/* ------------------------------------------------------------------ space-time U-Net denoiser
 * Replaces model/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py
 *   Unet3D_with_Conv3D.__init__ :357-471 (create/load), .forward :486-552 (forward).
 */
typedef struct dpc_unet3d_s* dpc_unet3d_t;

typedef struct {
    int32_t dim;            /* :359 */
    int32_t n_mults;        /* len(dim_mults) :362 */
    int32_t dim_mults[8];
    int32_t channels;       /* :363 */
    int32_t out_dim;        /* :361 (== channels when None) */
    int32_t attn_heads;     /* :364 */
    int32_t attn_dim_head;  /* :365  (only 32 is supported: one MFMA k-tile) */
    int32_t init_kernel;    /* :368 */
    int32_t groups;         /* :371 */
    int32_t micro_batch;    /* trajectories per internal pass (0 = whole batch) */
} dpc_unet3d_cfg;

int dpc_unet3d_create(const dpc_unet3d_cfg* cfg, dpc_unet3d_t* out);
void dpc_unet3d_destroy(dpc_unet3d_t h);

/* One call per state-dict entry, `name` as in the reference's state_dict() (e.g.
 * "downs.0.0.block1.proj.weight"); `w_d` is a device pointer in the reference's layout and is
 * re-packed into the kernels' layout on `stream`.  Unknown names -> DPC_ERR_ARG.
 * (checkpoint reader side: diffusion/diffusion_2d_smoke.py:956-985) */
int dpc_unet3d_load(dpc_unet3d_t h, const char* name, const float* w_d, const int64_t* shape, int ndim,
                    dpc_stream_t stream);

/* Small immutable tables computed once on the host in the reference's own arithmetic:
 *   relpos_bias_d [heads, F, F]  = RelativePositionBias.forward(F)        (…conv3d.py:106-112)
 *   rot_cos_d / rot_sin_d [F, attn_dim_head] interleaved-pair RoPE tables  (…conv3d.py:320-321, 380)
 *   sin_freqs_d [dim/2]           = exp(arange(dim/2) * -log(1e4)/(dim/2-1)) (…conv3d.py:146-148)
 */
int dpc_unet3d_set_tables(dpc_unet3d_t h, int frames, const float* relpos_bias_d, const float* rot_cos_d,
                          const float* rot_sin_d, const float* sin_freqs_d, dpc_stream_t stream);

/* Verifies every parameter has been loaded. */
int dpc_unet3d_finalize(dpc_unet3d_t h);

size_t dpc_unet3d_workspace_bytes(dpc_unet3d_t h, int B, int F, int H, int W);

/* x [B,F,C,H,W] fp32, t [B] int64 -> out [B,F,out_dim,H,W] fp32   (…conv3d.py:486-552).
 * x may be a channel slice of a wider tensor [B,F,x_channels_total,H,W] starting at x_channel_offset (the
 * prior model reads x[:, :, 3:5] of the joint state, diffusion_2d_smoke.py:612-613); pass 0,0 for a plain tensor. */
int dpc_unet3d_forward(dpc_unet3d_t h, const float* x, int x_channels_total, int x_channel_offset, const int64_t* t,
                       float* out, int B, int F, int H, int W, void* ws, size_t ws_bytes, dpc_stream_t stream);

/* Debug/test hook: copy a named internal activation of the LAST micro-batch of the last forward,
 * converted to the reference's channels-first layout [mb,C,F,H,W], into dst_d (caller-sized).
 * Only active when enabled before the forward. Names as in oracle taps ("init_conv", "downs.0.0", …). */
const char* dpc_unet3d_modes(dpc_unet3d_t h);
/* Opt-in f16x3 activation range check: with enable != 0 every f16x3 conv / implicit-GEMM / stem launch of a forward is
 * preceded by a pass over its input; dpc_unet3d_forward then returns DPC_ERR_STATE (naming the first such op) when an
 * activation lies outside the range the 2^4 pre-scale keeps exact (|x| <= 4094) instead of silently clamping it.
 * Costs one extra read of every conv input and one host sync per forward: a validation aid (e.g. the first run of a
 * new checkpoint), not for the timed path. */
int dpc_unet3d_set_range_check(dpc_unet3d_t h, int enable);
/* ALWAYS-ON range sentinel of the f16x3 mode (no extra pass, no per-forward sync): the kernels that PRODUCE the residual-stream
 * tensors -- the only tensors a later f16x3 kernel splits without a normalisation in front (ResnetBlock / res_conv / Downsample /
 * Upsample inputs, ...conv3d.py:206-230,159-163) -- OR a bit into a device word of the handle when an output has |x| > 4094 or is
 * not finite (GroupNorm-apply + residual pass, implicit-GEMM vector epilogue, fused temporal attention store).
 * dpc_unet3d_range_status reads that word (one host sync; reset != 0 clears it): DPC_OK, or DPC_ERR_STATE when any forward since
 * the last cleared status left the range.  The Python sampler calls it once at the end of sample().  Contract of the f16x3
 * mode, per kernel family: activations up to |x| <= 4094 are represented with 22 significant bits by EVERY kernel; beyond it the
 * direct convolutions / implicit GEMMs / stem clamp at 4094 and the Winograd convolution stays exact up to 4094 (plain input, pre-scale 2^3) /
 * 5676 (fused GroupNorm input) and then yields inf -> NaN; weights must satisfy |w| <= 15.99 (checked by dpc_unet3d_finalize). */
int dpc_unet3d_range_status(dpc_unet3d_t h, int reset, dpc_stream_t stream);
int dpc_unet3d_debug_taps(dpc_unet3d_t h, int enable);
int dpc_unet3d_get_tap(dpc_unet3d_t h, const char* name, float* dst_d, size_t dst_floats, dpc_stream_t stream);

/* ------------------------------------------------------------------ guided DDPM / DDIM update
 * Replaces diffusion/diffusion_2d_smoke.py model_predictions :610-656 (after the two denoiser
 * calls), p_mean_variance :659-666, p_sample :672-699, the in-paint of p_sample_loop :720 and
 * the ddim_sample body :759-775, with the analytic gradient of inference_2d_smoke.py:30-44.
 */
typedef struct {
    float sqrt_recip_ac;     /* extract(sqrt_recip_alphas_cumprod, t)   :578 */
    float sqrt_recipm1_ac;   /* extract(sqrt_recipm1_alphas_cumprod, t) :579 */
    float mean_coef1;        /* posterior_mean_coef1[t] :603  | DDIM: sqrt(alpha_next) :771 */
    float mean_coef2;        /* posterior_mean_coef2[t] :604  | DDIM: c :767 */
    float sigma;             /* exp(0.5*posterior_log_variance_clipped[t]) :685 (0 at t==0) | DDIM sigma :766 */
    float guide_scale;       /* standard_fixed_ratio :630 or eta(t)=coeff_ratio*betas.flip(0)[t] :632-635 */
    float w_scale;           /* (w_prob_exp - 1) :630 */
    float w_energy;          /* inference_2d_smoke.py:41 */
    int32_t mode;            /* 0 = DDPM p_sample, 1 = DDIM step, 2 = DDIM last step (return x0) */
    int32_t clip_x_start;    /* DDIM: clip both x0 (:616,641) */
} dpc_step_coef;

/* x, eps_j, z, x_next: [B,F,C,H,W]; eps_w: [B,F,2,H,W]; init: [B,H,W]; rescaler: [C] (device).
 * z may be NULL when sigma == 0.  x0_out may be NULL.  x_next may alias x.
 * Guidance gradient channels follow the reference: objective on [b,F-1,C-1], energy and the
 * prior-reweighting term on channels 3:5 (requires C >= 5). */
int dpc_ddpm_update_smoke(const float* x, const float* eps_j, const float* eps_w, const float* z, const float* init,
                          const float* rescaler, float* x_next, float* x0_out, const dpc_step_coef* coef, int B,
                          int F, int C, int H, int W, dpc_stream_t stream);

/* Counter-based N(0,1) noise keyed (seed, global trajectory index, draw index, element index):
 * a trajectory's noise does not depend on how the batch is sharded over GPUs.
 * out [B, per_traj] ; traj0 = global index of the first local trajectory. */
int dpc_philox_normal(float* out, int B, int64_t per_traj, uint64_t seed, int64_t traj0, int64_t draw,
                      dpc_stream_t stream);

/* ------------------------------------------------------------------ operator-level entry points
 * (used by the parity tests; the U-Net forward composes exactly these kernels)
 * Activations are channels-last [B,F,H,W,C] ("rows" = B*F*H*W points).
 */
/* Conv3d with reference-layout weight [Cout,Cin,kd,kh,kw] (nn.Conv3d, …conv3d.py:192,163,216,392,470) */
int dpc_conv3d_cl(const float* x_cl, const float* w_ref_d, const float* bias_d, float* out_cl, int B, int F, int H,
                  int W, int Cin, int Cout, int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                  void* ws, size_t ws_bytes, dpc_stream_t stream);
/* ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1), weight [Cin,Cout,1,4,4] (…conv3d.py:159-160) */
int dpc_convtranspose3d_144_cl(const float* x_cl, const float* w_ref_d, const float* bias_d, float* out_cl, int B,
                               int F, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes,
                               dpc_stream_t stream);
size_t dpc_conv_workspace_bytes(int Cin, int Cout, int ntaps);
/* GroupNorm(groups, eps 1e-5, affine) -> *(scale+1)+shift (optional, [B,2C] as chunked in :223-225) -> SiLU,
 * in place on x_cl (…conv3d.py:193-204) */
int dpc_groupnorm_silu_cl(float* x_cl, const float* gamma_d, const float* beta_d, const float* scale_shift_d, int B,
                          int64_t rows_per_sample, int C, int groups, void* ws, size_t ws_bytes, dpc_stream_t stream);
size_t dpc_groupnorm_workspace_bytes(int B, int C);
size_t dpc_linear_attention_workspace_bytes(int64_t images, int heads);
/* Attention core on a packed qkv tensor [rows, 3*heads*32]: sequences of L tokens, token stride and
 * sequence addressing given in rows; optional rotary tables [L,32] and bias [heads,L,L]
 * (…conv3d.py:311-351).  out [rows, heads*32]. */
int dpc_attention_core(const float* qkv, float* out, int heads, int L, int64_t n_seq, int64_t seq_inner,
                       int64_t seq_outer_stride_rows, int64_t seq_inner_stride_rows, int64_t token_stride_rows,
                       const float* rot_cos_d, const float* rot_sin_d, const float* bias_d, dpc_stream_t stream);
/* Spatial linear attention core (…conv3d.py:246-255) on qkv [images*N, 3*heads*32] -> out [images*N, heads*32] */
int dpc_linear_attention_core(const float* qkv, float* out, int heads, int64_t images, int N, void* ws,
                              size_t ws_bytes, dpc_stream_t stream);

/* ------------------------------------------------------------------ operator set of the jellyfish guidance surrogates
 * The two learned 2-D nets inside the design gradient -- diffusion/diffusion_2d_jellyfish.py `Unet` :276-403 (boundary
 * updater) and `ForceUnet` :406-481, differentiated by inference_2d_jellyfish.py force_fn :85-114 -- run forward AND
 * input-gradient backward on these entry points (graph in diffphycon_amd/model/surrogates_hip.py); all tensors are
 * channels-last fp32 [images * H * W, C] unless stated.
 *
 * dpc_conv_pack: weight [N][K][kh][kw] (reference Conv2d layout; the host passes flipped / transposed / sliced /
 * weight-standardised :107-120 copies for the backward-data and special cases), taps [tap_begin, tap_end) of the kernel
 * window (<= 32 per pack; tap_end <= 0 means all), K % 4 == 0; mode "" (process default) | "f32" | "x6" | "f16x3".
 * dpc_conv_run: out = conv(cat(a0[C0], a1[C1])) + bias + resid; ln_stats/ln_gamma: channel-LayerNorm prologue (:195-210)
 * for 1-tap ops; out_mode 0 [rows][N], 1 channels-first [images][N][Ho*Wo], 2 parity scatter into [images][2Ho][2Wo][N].
 * act_scale (f16x3 mode; 0 = the default 2^4): power-of-two scale applied to the input before the fp16 operand split and
 * undone in the epilogue -- the backward pass uses it to place gradients of any magnitude inside the fp16 window.
 * a0_stride (0 = C0): floats between consecutive pixels of a0 when the C0 "channels" of a pixel are a window over the
 * following pixels -- the 7x7 stems :296 run as 7 row taps over 7 * 4 contiguous floats of a width-padded image. */
typedef struct dpc_conv_s* dpc_conv_t;
int dpc_conv_pack(const float* w, int N, int K, int kh, int kw, int sh, int sw, int ph, int pw, int tap_begin, int tap_end,
                  const char* mode, dpc_conv_t* out, dpc_stream_t stream);
void dpc_conv_free(dpc_conv_t h);
int dpc_conv_run(dpc_conv_t h, const float* a0, const float* a1, int C0, int C1, const float* bias, const float* resid, float* out,
                 int images, int Hi, int Wi, int Ho, int Wo, const float* ln_stats, const float* ln_gamma, int out_mode, int par_a,
                 int par_b, float act_scale, int a0_stride, dpc_stream_t stream);
/* GroupNorm fused around the two 3x3 convolutions of a ResnetBlock (r05; diffusion_2d_jellyfish.py:122-148 Block / ResnetBlock, the same
 * fusion the 2-D denoiser uses internally): where dpc_conv_gn_fusable says the convolution runs on the halo-tile kernel (f16x3 mode, 3x3
 * stride 1, N % 64 == 0, H % 8 == 0, W % 8 == 0, C0 % 4 == 0, C1 % 4 == 0 -- shape only, never the batch; 0 = use the standalone GroupNorm passes), dpc_conv_run_gn
 *   - with gn_part != NULL also emits per-image partial sums of its OUTPUT, [images][dpc_conv_gn_entries(H, W)][N][2] floats, from which
 *     dpc_gn_finalize_fused forms stats [images][groups][2] = (mean, rstd) -- no statistics pass over the tensor -- and, with coef != NULL
 *     ([images * C * 7] floats), the per-channel coefficients of GN -> x (scale + 1) + shift (scale_shift [images][2C] or NULL);
 *   - with in_coef != NULL (that table) applies GroupNorm + (scale, shift) + SiLU to its INPUT on the fly: the activated tensor of
 *     block1 never exists in HBM.  Zero padding applies to the activated tensor, as in the reference. */
int dpc_conv_gn_fusable(dpc_conv_t h, int H, int W, int C0, int C1);    /* C0 + C1 = the input channels of the two (virtually concatenated) sources */
int64_t dpc_conv_gn_entries(int H, int W);
int dpc_conv_run_gn(dpc_conv_t h, const float* a0, const float* a1, int C0, int C1, const float* bias, float* out, int images, int H, int W,
                    float* gn_part, const float* in_coef, dpc_stream_t stream);
int dpc_gn_finalize_fused(const float* part, int images, int64_t entries, int C, int groups, int64_t rows_per_image, const float* gamma,
                          const float* beta, const float* scale_shift, float* stats, float* coef, dpc_stream_t stream);
/* GroupNorm (:140-157 Block): stats [B][groups][2] = (mean, rstd); apply: out = SiLU(GN(x) * (scale + 1) + shift) (+ resid),
 * scale_shift [B][2C] or NULL; backward: dx and (dss != NULL) d scale_shift [B][2C] given dy. */
size_t dpc_gn_workspace_bytes(int B, int C);
int dpc_gn_stats(const float* x, float* stats, int B, int64_t rows_per_sample, int C, int groups, void* ws, size_t ws_bytes,
                 dpc_stream_t stream);
int dpc_gn_apply(const float* x, float* out, const float* resid, const float* stats, const float* gamma, const float* beta,
                 const float* scale_shift, int B, int64_t rows_per_sample, int C, int groups, dpc_stream_t stream);
int dpc_gn_silu_bwd(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                    const float* scale_shift, float* dx, float* dss, int B, int64_t rows_per_sample, int C, int groups, void* ws,
                    size_t ws_bytes, dpc_stream_t stream);
/* channel LayerNorm (:195-204): stats [rows][2]; apply: out = resid + (x - mean) * rstd * g; backward: dx (accumulate != 0: +=). */
int dpc_ln_stats(const float* x, float* stats, int64_t rows, int C, dpc_stream_t stream);
int dpc_ln_apply(const float* x, const float* stats, const float* g, const float* resid, float* out, int64_t rows, int C,
                 dpc_stream_t stream);
int dpc_ln_bwd(const float* x, const float* stats, const float* g, const float* dy, float* dx, int64_t rows, int C, int accumulate,
               dpc_stream_t stream);
/* LinearAttention :232-251 (without the v / (h w), which the host folds into to_out) with a tape: the forward keeps the
 * per-(image, head) context and the statistics of the k softmax, the backward turns (qkv, dout, tape) into dqkv
 * [rows][3*heads*32].  dpc_attention_bwd: backward of dpc_attention_core for whole-image sequences (Attention :266-275,
 * L <= 256 tokens). */
size_t dpc_linear_attention_tape_bytes(int64_t images, int heads);
int dpc_linear_attention_fwd_save(const float* qkv, float* out, int heads, int64_t images, int N, void* tape, size_t tape_bytes,
                                  dpc_stream_t stream);
int dpc_linear_attention_bwd(const float* qkv, const float* dout, float* dqkv, int heads, int64_t images, int N, void* tape,
                             size_t tape_bytes, dpc_stream_t stream);
int dpc_attention_bwd(const float* qkv, const float* dout, float* dqkv, int heads, int64_t images, int L, dpc_stream_t stream);
/* nearest x2 up-sampling [images][H][W][C] -> [images][2H][2W][C] (Upsample :89-93) and its backward (2 x 2 block sums) */
int dpc_upsample2x_cl(const float* x, float* y, int images, int H, int W, int C, dpc_stream_t stream);
int dpc_downsum2x_cl(const float* dy, float* dx, int images, int H, int W, int C, dpc_stream_t stream);
/* layout glue at the boundary of the surrogates: channels-first [N][C][HW] <-> channels-last [N*HW][Cpad] */
int dpc_nchw_to_cl(const float* x, float* y, int64_t N, int C, int Cpad, int64_t HW, dpc_stream_t stream);
int dpc_cl_to_nchw(const float* x, float* y, int64_t N, int C, int Cpad, int csrc, int Ctot, int cdst, float mul, int64_t HW,
                   dpc_stream_t stream);
/* rows of W pixels -> rows of W + 2 wpad pixels (zero borders); dx[row][w][c] = sum_b x[row][w + taps/2 - b][b*C + c] */
int dpc_pad_w_cl(const float* x, float* y, int64_t rows, int W, int C, int wpad, dpc_stream_t stream);
int dpc_fold_w_cl(const float* x, float* dx, int64_t rows, int W, int C, int taps, dpc_stream_t stream);
/* y_cl[n][hw][cdst] = a * x[n][csrc][hw] + b (force_fn :96-101: un-normalised pressure into the ForceUnet input) */
int dpc_channel_affine_to_cl(const float* x, float* y, int64_t N, int Ctot, int csrc, int Cpad, int cdst, float a, float b,
                             int64_t HW, dpc_stream_t stream);
/* out[n] = mean_hw x[n][csrc][hw] (force_fn :94 theta) and its backward y[n][cdst][hw] = v[n] * mul */
int dpc_channel_mean(const float* x, float* out, int64_t N, int Ctot, int csrc, int64_t HW, dpc_stream_t stream);
int dpc_channel_fill(float* y, const float* v, int64_t N, int Ctot, int cdst, float mul, int64_t HW, dpc_stream_t stream);
/* out[n][c] = mean_r x[n][r][c] (ForceUnet head :478-480) and its backward y[n][r][c] = v[n][c] * mul;  y += x */
int dpc_mean_rows(const float* x, float* out, int64_t N, int64_t R, int C, dpc_stream_t stream);
int dpc_bcast_rows(const float* v, float* y, int64_t N, int64_t R, int C, float mul, dpc_stream_t stream);
int dpc_add_inplace(float* y, const float* x, int64_t n, dpc_stream_t stream);
/* out[0] = max(out[0], max_i |x[i]|) (the caller zeroes out; range calibration of the f16x3 backward pass) */
int dpc_absmax(const float* x, int64_t n, float* out, dpc_stream_t stream);
/* out[b][n] = out_act(bias[n] + sum_k in_act(in[b][k]) W[n][k]), act 0 none | 1 SiLU | 2 GELU (time MLPs :300-305, :128-131) */
int dpc_small_linear(const float* in, const float* W, const float* bias, float* out, int B, int K, int N, int in_act, int out_act,
                     dpc_stream_t stream);

/* ------------------------------------------------------------------ training step of the smoke denoiser
 * diffusion/diffusion_2d_smoke.py: q_sample :791-797, p_losses :809-831, Trainer.train :998-1054 (accelerator.backward :1025,
 * clip_grad_norm_(1.0) :1027, Adam :912/:1035, EMA :920/:1043); the network is Unet3D_with_Conv3D
 * (model/video_diffusion_pytorch/video_diffusion_pytorch_conv3d.py).  The backward pass is an explicit tape over these operators
 * (diffphycon_amd/model/video_diffusion_pytorch/unet3d_train.py) -- no autograd graph, no torch convolution.
 *
 * dpc_conv3_pack / dpc_conv3_run: the 3-D form of dpc_conv_pack / dpc_conv_run: any Conv3d of the net but the stem (3x3x3 on
 * the halo-tile kernels, 1x1x1, (1,4,4)/(1,2,2), the 2x2-tap parity classes of ConvTranspose3d with out_mode 2), reference weight
 * layout [N][K][kd][kh][kw], mode "" | "f32" | "x6" | "f16x3".  *inout != NULL re-packs in place (no allocation, no host sync:
 * the weights change every optimizer step); dpc_weight_range_check() reads the f16x3 weight-range flag once (DPC_ERR_STATE when a
 * packed weight left the fp16 window since the last check).  Backward-data convolutions are the same operator on flipped /
 * transposed weights with act_scale (power of two) placing the gradient inside the fp16 window, or in mode "x6".
 * dpc_stem_pack / dpc_stem_run: the 7x7x7 init_conv (:392) on the reference-layout state [B][F][ctot][H][W]. */
int dpc_conv3_pack(const float* w, int N, int K, int kd, int kh, int kw, int sh, int sw, int pd, int ph, int pw, const char* mode,
                   dpc_conv_t* inout, dpc_stream_t stream);
int dpc_conv3_run(dpc_conv_t h, const float* a0, const float* a1, int C0, int C1, const float* bias, const float* resid, float* out,
                  int B, int F, int Hi, int Wi, int Ho, int Wo, const float* ln_stats, const float* ln_gamma, int out_mode, int par_a,
                  int par_b, float act_scale, dpc_stream_t stream);
int dpc_weight_range_check(void);
typedef struct dpc_stem_s* dpc_stem_t;
int dpc_stem_pack(const float* w, int N, int C, int k, const char* mode, dpc_stem_t* inout, dpc_stream_t stream);
void dpc_stem_free(dpc_stem_t h);
int dpc_stem_run(dpc_stem_t h, const float* x, int x_channels_total, int x_channel_offset, const float* bias, float* out, int B, int F,
                 int H, int W, dpc_stream_t stream);
/* Convolution weight gradient (what accelerator.backward :1025 leaves in Conv3d.weight.grad): x channels-last [B,F,Hi,Wi,C],
 * dy channels-last [B,F,Ho,Wo,N], dw in the reference layout [N][dw_ctot][kf][kh][kw], channel slice [dw_coff, dw_coff + c_valid)
 * (a concatenated input = one call per source; c_valid < C: zero-padded input channels, 0 = C), dw = (accumulate ? dw : 0) +
 * scale * sum_p dy[p][n] x[p + tap][c].  C must divide or be a multiple of 32.  ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1)
 * (weight [Cin][Cout][1][4][4]): call with x = the transposed conv's OUTPUT gradient and dy = its input, geometry of the
 * (1,4,4)/(1,2,2)/(0,1,1) convolution.  Exact fp32 products (native fp32 MFMA), fixed summation order.
 * f16_dy_scale != 0 (a power of two): the 3x3x3 stride-1 pad-1 convolutions with W in {16, 32, 64}, C % 32 == 0, N % 64 == 0
 * run on the fp16 matrix cores instead (csrc/wgrad3.hip: LDS transpose reads, f16x3 = 22-bit split operands, 3 MFMAs per product,
 * fp32 accumulation): x is pre-scaled by 2^4 like every f16x3 activation operand, dy by f16_dy_scale (saturating at 65504), both
 * undone in the fixed-order reduction; other shapes ignore the flag.  rows of the workspace query: B * F * Ho.
 * dy_abs_limit > 0 (r04; the launches that stay on the fp32 MFMA): an element of the GRADIENT operand with |v| > dy_abs_limit (or not
 * finite) raises bit 1 of the same device word -- the gradient operand of a layer's weight gradient is the input of that layer's
 * backward-DATA convolution, which CLAMPS at 4094 in the f16x3 mode: with dy_abs_limit = 4094 on every layer no backward-data clamp
 * goes unseen (the f16x3 launches check theirs against 65504 / f16_dy_scale <= 4094); an element of the ACTIVATION operand beyond the
 * limit raises bit 0 (r05: it used to raise bit 1 as well, which a loss-scale halving cannot cure).  0 = no check.
 * accumulate is a bit set: 1 = add to dw; 2 = the gradient operand is x and the activation is dy (the ConvTranspose call form above).
 * The device word is per device (the current device of the calling thread).
 * Both operands SATURATE at the fp16 limit (|x| > 4094, |dy| > 65504 / f16_dy_scale) and a saturated (or non-finite) element raises
 * a device word; dpc_train_range_status reads it (ONE host sync; reset != 0 clears it): DPC_OK, or DPC_ERR_STATE when any
 * f16x3 weight-gradient launch since the last reset clamped an operand -- that step's gradients are then not exact.  The Trainer
 * (diffusion_2d_smoke.py Trainer.train :998-1054) asks where it already syncs: when it logs the loss and before it saves. */
int dpc_train_range_status(int reset, dpc_stream_t stream);
/* Dynamic loss scaling (what accelerate's GradScaler does for Trainer(fp16=True), diffusion_2d_smoke.py:871-874, 1025-1035: skip the
 * step and halve the scale when a gradient overflowed): enqueue BEFORE the gradient all-reduce.  If an output gradient was clamped /
 * not finite since the last call (bit 1 of the word above) g[0] (the flat gradient buffer) becomes +inf and the bit is cleared: the
 * all-reduce carries it to every rank, dpc_l2_norm's result is then not finite on all of them, and the host -- which reads that norm
 * once per step in this mode -- skips dpc_adam_ema_step everywhere.  No host sync, no extra collective.  Bit 0 (an ACTIVATION
 * outside |x| <= 4094: not a loss-scale matter) stays for dpc_train_range_status. */
int dpc_train_range_poison(float* g, dpc_stream_t stream);
size_t dpc_conv_wgrad_workspace_bytes(int C, int N, int kf, int kh, int kw, int64_t rows /* B * F * Ho */);
int dpc_conv_wgrad_cl(const float* x, const float* dy, float* dw, int B, int F, int Hi, int Wi, int C, int Ho, int Wo, int N, int kf,
                      int kh, int kw, int sh, int sw, int pf, int ph, int pw, int c_valid, int dw_ctot, int dw_coff, float scale,
                      float f16_dy_scale, float dy_abs_limit, int accumulate, void* ws, size_t ws_bytes, dpc_stream_t stream);
/* out[c] = (accumulate ? out[c] : 0) + scale * sum_r dy[r][c] (x NULL: bias gradients) or
 * scale * sum_r dy[r][c] (x[r][c] - mean_r) rstd_r (channel-LayerNorm gamma gradient :195-204, ln_stats [rows][2]); fp64 partials */
size_t dpc_colsum_workspace_bytes(int C);
int dpc_colsum(const float* dy, const float* x, const float* ln_stats, float* out, int64_t rows, int C, float scale, int accumulate,
               void* ws, size_t ws_bytes, dpc_stream_t stream);
/* dpc_gn_silu_bwd + the affine gradients d gamma, d beta [C] of the GroupNorm (Block.norm :193) */
int dpc_gn_silu_bwd_params(const float* x, const float* dy, const float* stats, const float* gamma, const float* beta,
                           const float* scale_shift, float* dx, float* dss, float* dgamma, float* dbeta, int B,
                           int64_t rows_per_sample, int C, int groups, void* ws, size_t ws_bytes, dpc_stream_t stream);
/* Backward of dpc_attention_core for sequences of L <= 64 tokens with the same addressing, rotary tables and bias
 * (Attention.forward :293-352: q scale, rotary on q and k, + pos_bias, softmax, PV): dqkv [rows][3*heads*32] given dout
 * [rows][heads*32]; dbias [heads][L][L] (NULL: skipped) = (accumulate_dbias ? dbias : 0) + sum over sequences of dS. */
size_t dpc_attention_bwd_seq_workspace_bytes(int heads, int L);
int dpc_attention_bwd_seq(const float* qkv, const float* dout, float* dqkv, float* dbias, int heads, int L, int64_t n_seq,
                          int64_t seq_inner, int64_t seq_outer_stride_rows, int64_t seq_inner_stride_rows, int64_t token_stride_rows,
                          const float* rot_cos, const float* rot_sin, const float* bias, int accumulate_dbias, void* ws, size_t ws_bytes,
                          dpc_stream_t stream);
/* Backward of dpc_small_linear (out = bias + in_act(x) W^T): dW [N][K], db [N] (NULL: skipped) and dx [B][K] (NULL: skipped;
 * gradient w.r.t. the PRE-activation input, accumulate_dx != 0: +=) given dy [B][N] */
int dpc_small_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int B, int K, int N,
                         int in_act, int accumulate_dx, dpc_stream_t stream);
/* p_losses :811-816: state = sqrt_ac[t_b] x0 + sqrt_1mac[t_b] noise with state[:, 0, 0] = x0[:, 0, 0]; target = noise with
 * [:, 0, 0] = 0.  x0 = channel slice of a [B][F][ctot][H][W] tensor (Trainer.train :1018-1019 trains the w model on [:, :, 3:5]). */
int dpc_q_sample_smoke(const float* x0, int x_channels_total, int x_channel_offset, const float* noise, const int64_t* t,
                       const float* sqrt_alphas_cumprod, const float* sqrt_one_minus_alphas_cumprod, float* state, float* target, int B,
                       int F, int C, int H, int W, dpc_stream_t stream);
/* loss[0] = mean (out - target)^2 (F.mse_loss :827); dout (NULL: skipped) = grad_scale * 2 (out - target) / n */
size_t dpc_reduce_workspace_bytes(void);
int dpc_mse_loss_grad(const float* out, const float* target, float* dout, float* loss, int64_t n, float grad_scale, void* ws,
                      size_t ws_bytes, dpc_stream_t stream);
/* out[0] = scale * ||x||_2 (clip_grad_norm_'s total norm :1027 over the flat gradient buffer), fp64 partials, fixed order */
int dpc_l2_norm(const float* x, int64_t n, float scale, float* out, void* ws, size_t ws_bytes, dpc_stream_t stream);
/* One optimizer step on flat fp32 buffers (Trainer.train :1027-1043): g <- g * grad_inv_scale * min(1, max_norm / (total_norm[0] +
 * 1e-6)) (total_norm NULL or max_norm <= 0: no clipping), torch.optim.Adam's update with its association (exp_avg.lerp_,
 * exp_avg_sq.mul_.addcmul_, param.addcdiv_(exp_avg, sqrt(v) / sqrt(1 - beta2^step) + eps, -lr / (1 - beta1^step))), then the EMA
 * of ema-pytorch 0.7.3 on `ema`: ema_mode 0 none | 1 copy | 2 lerp with ema_weight = 1 - decay | 3 copy then lerp.
 * lr, beta1, beta2, eps are doubles like torch's Python floats: 1 - beta, lr / (1 - beta1^step), sqrt(1 - beta2^step) are formed in
 * double and rounded to fp32 once (1.f - 0.99f differs from float(1 - 0.99) by 1e-5 relative). */
int dpc_adam_ema_step(float* w, const float* g, float* m, float* v, float* ema, int64_t n, const float* total_norm, float max_norm,
                      float grad_inv_scale, double lr, double beta1, double beta2, double eps, int step, int ema_mode, float ema_weight,
                      dpc_stream_t stream);

/* ------------------------------------------------------------------ Burgers finite-difference solver
 * Replaces dataset/apps/generate_burgers.py:207-299 burgers_numeric_solve_free (fp32, explicit Euler).
 * u0 [N,nx], f [N,num_t,nx] -> traj [N,num_t+1,nx]. */
int dpc_burgers_fd(const float* u0, const float* f, float* traj, int N, int nx, int num_t, double visc, double T,
                   double dt, dpc_stream_t stream);


/* ------------------------------------------------------------------ Burgers space-time U-Net denoiser
 * Replaces model/burgers_1d/unet.py Unet2D.__init__ :267-385 (create/load) and .forward :387-431, built by
 * train/train_1d_burgers.py:get_2d_ddpm :113-143 (channels 2, out_dim 2, attn 4 x 32, sinusoidal time embedding).
 * Parameter names are the reference's state_dict() keys (e.g. "downs.0.2.fn.fn.to_out.1.g").
 */
typedef struct dpc_unet2d_s* dpc_unet2d_t;
typedef struct {
    int32_t dim;            /* :274 */
    int32_t n_mults;        /* len(dim_mults) :277 */
    int32_t dim_mults[8];
    int32_t channels;       /* :278 */
    int32_t out_dim;        /* :276 */
    int32_t attn_heads;     /* :286 */
    int32_t attn_dim_head;  /* :285 (only 32) */
    int32_t groups;         /* resnet_block_groups :280 (scripts: 1) */
    int32_t micro_batch;    /* trajectories per internal pass (0 = whole batch) */
} dpc_unet2d_cfg;
int dpc_unet2d_create(const dpc_unet2d_cfg* cfg, dpc_unet2d_t* out);
void dpc_unet2d_destroy(dpc_unet2d_t h);
int dpc_unet2d_load(dpc_unet2d_t h, const char* name, const float* w_d, const int64_t* shape, int ndim,
                    dpc_stream_t stream);
/* sin_freqs_d [dim/2] = exp(arange(dim/2) * -log(theta)/(dim/2-1))  (unet.py:93-95) */
int dpc_unet2d_set_tables(dpc_unet2d_t h, const float* sin_freqs_d, dpc_stream_t stream);
int dpc_unet2d_finalize(dpc_unet2d_t h);
size_t dpc_unet2d_workspace_bytes(dpc_unet2d_t h, int B, int H, int W);
/* x [B,channels,H,W] fp32 (H = padded time rows 16, W = space cells 128), t [B] int64 -> out [B,out_dim,H,W] */
int dpc_unet2d_forward(dpc_unet2d_t h, const float* x, const int64_t* t, float* out, int B, int H, int W, void* ws,
                       size_t ws_bytes, dpc_stream_t stream);
const char* dpc_unet2d_modes(dpc_unet2d_t h);
int dpc_unet2d_debug_taps(dpc_unet2d_t h, int enable);
int dpc_unet2d_get_tap(dpc_unet2d_t h, const char* name, float* dst_d, size_t dst_floats, dpc_stream_t stream);

/* ------------------------------------------------------------------ Burgers guided DDPM update
 * Replaces diffusion/diffusion_1d_burgers.py: set_condition :500-522 + the zero-fill of p_sample_loop :539-553
 * (dpc_burgers_prepare, which also builds the prior model's input x_w :399-400), and model_predictions :402-441
 * (after the denoiser calls) + p_mean_variance :452-461 + p_sample :464-470 (dpc_ddpm_update_burgers), with the
 * autograd gradient of ddpm_guidance_loss (utils.py:1289-1328; inference_1d_burgers.py:129-165) in closed form.
 */
typedef struct {
    float sqrt_recip_ac;     /* extract(sqrt_recip_alphas_cumprod, t)   :364 */
    float sqrt_recipm1_ac;   /* extract(sqrt_recipm1_alphas_cumprod, t) :365 */
    float mean_coef1;        /* posterior_mean_coef1[t] :388 */
    float mean_coef2;        /* posterior_mean_coef2[t] :389 */
    float sigma;             /* exp(0.5*posterior_log_variance_clipped[t]) :469 (ignored when z == NULL) */
    float w_coef;            /* two models: (1-prior_beta)*eta_w(t) :409, or (1-prior_beta) when normalize_beta :407 */
    float prior_beta;        /* divisor of the normalize_beta branch :407 */
    float eta_J;             /* nablaJ_scheduler(t) :432 */
    float wu, wf, wreg;      /* guidance weights of ddpm_guidance_loss */
    int32_t two_models;      /* eval_two_models :397 */
    int32_t normalize_beta;  /* :406 */
    int32_t partially_observed; /* 1 = 'front_rear_quarter': centre half of loss_u zeroed, utils.py:1311-1314 */
    int32_t guidance_batch;  /* batch size the reference averages the loss over (the whole sample() batch) */
    int32_t clip_denoised;   /* :457 */
    int32_t cond_idx;        /* condition_idx :224 (10): last physical time row */
} dpc_burgers_coef;

/* img [B,2,nt,nx] in place: img[:,0,0,:] = u0, img[:,0,cond_idx,:] = uT (NULL = skip), centre half of channel 0 zeroed
 * when set_zero; x_w (NULL = skip) receives a copy with rows 1..cond_idx-1 of channel 0 zeroed. */
int dpc_burgers_prepare(float* img, float* x_w, const float* u0, const float* uT, int B, int nt, int nx, int cond_idx,
                        int set_zero, dpc_stream_t stream);
/* x, eps_uw, eps_w, z, x_next, x0_out, eps_out: [B,2,nt,nx]; u_target [B,2,nx] = rescaled target rows (t=0, t=T) or
 * NULL when wu == 0.  eps_w may be NULL when !two_models; z NULL = no noise (t == 0).  x_next may alias x. */
int dpc_ddpm_update_burgers(const float* x, const float* eps_uw, const float* eps_w, const float* z,
                            const float* u_target, float* x_next, float* x0_out, float* eps_out,
                            const dpc_burgers_coef* coef, int B, int nt, int nx, dpc_stream_t stream);

/* ------------------------------------------------------------------ jellyfish guided update
 * Replaces diffusion/diffusion_2d_jellyfish.py: model_predictions :703-757 (after the two denoiser calls),
 * p_mean_variance :759-771, the posterior sample of p_sample :788-790 and of ddim_sample :938-946
 * (dpc_ddpm_update_jelly), and the guidance step `pred = pred - (eta_J*g - eta_w*pred_noise_w)` :792-804 /
 * `pred_noise_joint += grad_final` :733-741 (dpc_jelly_apply_guidance).  The gradient g comes from the two learned
 * 2-D surrogates (autograd, PyTorch-ROCm) and is passed in as a tensor.
 * x [B,F,Cx,H,W]: Cx = 7 (state 3, boundary 3, theta map 1) or 5 (--only_vis_pressure: pressure, boundary 3, theta);
 * the diffused channels are x[:, :, :n_state] and x[:, :, Cx-1] -> Cd = n_state + 1 channels in eps / z / pred / x0. */
typedef struct {
    float sqrt_recip_ac;     /* :676 */
    float sqrt_recipm1_ac;   /* :677 */
    float mean_coef1;        /* posterior_mean_coef1[t] :695 | DDIM sqrt(alpha_next) :943 */
    float mean_coef2;        /* posterior_mean_coef2[t] :696 | DDIM c :940 */
    float sigma;             /* exp(0.5*posterior_log_variance_clipped[t]) :790 | DDIM sigma :939 */
    int32_t clip_denoised;   /* DDPM :763 */
    int32_t mode;            /* 0 DDPM, 1 DDIM step, 2 DDIM last step (x0) */
} dpc_jelly_coef;
/* eps_guided: DDIM only (noise after the guidance term, :741); NULL = eps.  z NULL = no noise.  x0_out may be NULL. */
int dpc_ddpm_update_jelly(const float* x, const float* eps, const float* eps_guided, const float* z, float* pred,
                          float* x0_out, const dpc_jelly_coef* coef, int B, int F, int Cx, int n_state, int H, int W,
                          dpc_stream_t stream);
/* io[B,F,Cd,H,W] += sign * (eta_J * g - eta_w * w) with w = eps_w [B,F,1,H,W] broadcast over the Cd channels
 * (pad_w = 0, the DDPM path :800) or zero-padded onto the last channel only (pad_w = 1, :727-732).  g may be NULL. */
int dpc_jelly_apply_guidance(float* io, const float* g, const float* eps_w, float eta_J, float eta_w, int pad_w,
                             float sign, int B, int F, int Cd, int H, int W, dpc_stream_t stream);

/* ------------------------------------------------------------------ smoke PDE evaluator (phi rollout)
 * Replaces dataset/apps/evaluate_solver.py `solver` :205-310 (with `get_envolve` :118-147 and the vendored phi it
 * drives: FluidSimulation.divergence_free phi/flow.py:318-327, StaggeredGrid.divergence/gradient/advect
 * phi/math/nd.py:367-427,602-614, SparseCGPressureSolver phi/solver/sparse.py:88-128 + conjugate_gradient
 * phi/solver/base.py:56-104, SciPyBackend.resample phi/math/scipy_backend.py:58-78).  One persistent workgroup per
 * trajectory; fp64 with the reference's operation order and numpy's pairwise summation tree => bit-identical output.
 */
typedef struct {
    int32_t n;                  /* cells per side: FluidSimulation([n]*2), evaluate_solver.py:95 (127) */
    int32_t rim;                /* width of the controlled rim in velocity cells, :132-140 (16) */
    int32_t n_buckets;          /* :151-152 (7) */
    int32_t target_bucket;      /* index whose share is the objective, :303 (1) */
    int32_t bucket_rect[8][4];  /* y, x, len_y, len_x on the padded (n+1)^2 array, :151-152 */
    const int8_t* fluid_d;      /* [n,n] device, 1 = fluid, 0 = obstacle (FluidSimulation._fluid_mask) */
    const int8_t* active_d;     /* [n,n] device (FluidSimulation._active_mask) */
} dpc_smoke_domain;

typedef struct {
    void* densitys;             /* [B, T', W', W'] or NULL; T' = ceil(num_t/frame_stride), W' = (n+1)/space_stride */
    void* zero_densitys;        /* same shape or NULL */
    double* velocitys;          /* [B, T', W', W', 2] or NULL */
    double* smoke_out;          /* [B, T'] (the reference tiles this scalar over 128 x 128, :307-308) or NULL */
    int32_t* cg_iters;          /* [B, num_t-1] diagnostic or NULL */
    int32_t density_f32;        /* 0: densitys/zero_densitys are fp64 (reference dtype), 1: fp32 (same values) */
    int32_t frame_stride;       /* 1 = every frame (reference); multi_evaluate consumes ::8, inference_2d_smoke.py:389 */
    int32_t space_stride;       /* 1 = full 128 x 128 (reference); multi_evaluate consumes ::2, :388 */
} dpc_smoke_out;

size_t dpc_smoke_workspace_bytes(int n, int B);

/* init_velocity f32 [n+1,n+1,2] per trajectory (velocity_batch_stride floats apart; 0 = shared), dens0 f32 [B,nx,nx],
 * c1/c2 f32 [B,nt,nx,nx]; controls and density are nearest-upsampled x(128/nx) in space and x(num_t/nt) in time as
 * :221-227 does with np.tile. */
int dpc_smoke_rollout(const dpc_smoke_domain* dom, const float* init_velocity, int64_t velocity_batch_stride,
                      const float* dens0, const float* c1, const float* c2, int B, int nx, int nt, int num_t, double dt,
                      double accuracy, int max_cg_iter, const dpc_smoke_out* out, void* ws, size_t ws_bytes,
                      dpc_stream_t stream);

/* Operator-level entries (parity tests): masked-Laplacian CG exactly as phi/solver/base.py:56-104 from x0 = 0, in place
 * on div_to_pressure f64 [B,n,n]; one semi-Lagrangian step; the integer tables derived from the masks
 * (cf: bits 0-3 = off-diagonals (y-1,x),(y,x-1),(y,x+1),(y+1,x), bits 4-6 = -diagonal; vmask: bit0 x, bit1 y;
 * bucket: 0 or bucket index + 1). */
int dpc_smoke_pressure_solve(const dpc_smoke_domain* dom, double* div_to_pressure, int B, double accuracy,
                             int max_cg_iter, int32_t* iterations, void* ws, size_t ws_bytes, dpc_stream_t stream);
int dpc_smoke_advect(const double* velocity, const float* density, float* out, int n, double dt, dpc_stream_t stream);
int dpc_smoke_domain_tables(const dpc_smoke_domain* dom, unsigned char* cf_out, unsigned char* vmask_out,
                            unsigned char* bucket_out, void* ws, size_t ws_bytes, dpc_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPC_H */
