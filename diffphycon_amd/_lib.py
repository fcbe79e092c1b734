"""ctypes binding of libdpc.so (include/dpc.h).  There is NO fallback: if the HIP library is missing or fails
to load, importing any compute entry raises — the product path never silently runs on something else."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPC_LIB") or os.path.join(_HERE, "lib", "libdpc.so")     # DPC_LIB: an instrumented build (tools/)
_lib = None


class StepCoef(C.Structure):
    """dpc_step_coef (include/dpc.h)."""
    _fields_ = [("sqrt_recip_ac", C.c_float), ("sqrt_recipm1_ac", C.c_float), ("mean_coef1", C.c_float),
                ("mean_coef2", C.c_float), ("sigma", C.c_float), ("guide_scale", C.c_float), ("w_scale", C.c_float),
                ("w_energy", C.c_float), ("mode", C.c_int32), ("clip_x_start", C.c_int32)]


class Unet3DCfg(C.Structure):
    """dpc_unet3d_cfg (include/dpc.h)."""
    _fields_ = [("dim", C.c_int32), ("n_mults", C.c_int32), ("dim_mults", C.c_int32 * 8), ("channels", C.c_int32),
                ("out_dim", C.c_int32), ("attn_heads", C.c_int32), ("attn_dim_head", C.c_int32),
                ("init_kernel", C.c_int32), ("groups", C.c_int32), ("micro_batch", C.c_int32)]


class Unet2DCfg(C.Structure):
    """dpc_unet2d_cfg (include/dpc.h)."""
    _fields_ = [("dim", C.c_int32), ("n_mults", C.c_int32), ("dim_mults", C.c_int32 * 8), ("channels", C.c_int32),
                ("out_dim", C.c_int32), ("attn_heads", C.c_int32), ("attn_dim_head", C.c_int32), ("groups", C.c_int32),
                ("micro_batch", C.c_int32)]


class BurgersCoef(C.Structure):
    """dpc_burgers_coef (include/dpc.h)."""
    _fields_ = [("sqrt_recip_ac", C.c_float), ("sqrt_recipm1_ac", C.c_float), ("mean_coef1", C.c_float),
                ("mean_coef2", C.c_float), ("sigma", C.c_float), ("w_coef", C.c_float), ("prior_beta", C.c_float),
                ("eta_J", C.c_float), ("wu", C.c_float), ("wf", C.c_float), ("wreg", C.c_float),
                ("two_models", C.c_int32), ("normalize_beta", C.c_int32), ("partially_observed", C.c_int32),
                ("guidance_batch", C.c_int32), ("clip_denoised", C.c_int32), ("cond_idx", C.c_int32)]


class JellyCoef(C.Structure):
    """dpc_jelly_coef (include/dpc.h)."""
    _fields_ = [("sqrt_recip_ac", C.c_float), ("sqrt_recipm1_ac", C.c_float), ("mean_coef1", C.c_float),
                ("mean_coef2", C.c_float), ("sigma", C.c_float), ("clip_denoised", C.c_int32), ("mode", C.c_int32)]


class SmokeDomain(C.Structure):
    """dpc_smoke_domain (include/dpc.h)."""
    _fields_ = [("n", C.c_int32), ("rim", C.c_int32), ("n_buckets", C.c_int32), ("target_bucket", C.c_int32),
                ("bucket_rect", (C.c_int32 * 4) * 8), ("fluid_d", C.c_void_p), ("active_d", C.c_void_p)]


class SmokeOut(C.Structure):
    """dpc_smoke_out (include/dpc.h)."""
    _fields_ = [("densitys", C.c_void_p), ("zero_densitys", C.c_void_p), ("velocitys", C.c_void_p),
                ("smoke_out", C.c_void_p), ("cg_iters", C.c_void_p), ("density_f32", C.c_int32),
                ("frame_stride", C.c_int32), ("space_stride", C.c_int32)]


class ProfileRow(C.Structure):
    """dpc_profile_row (include/dpc.h)."""
    _fields_ = [("name", C.c_char_p), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


def profile_begin(classes=None):
    """Start per-class event timing; `classes` (iterable of class names) restricts the instrumentation."""
    if classes:
        check(lib().dpc_profile_begin_classes(",".join(classes).encode()))
    else:
        check(lib().dpc_profile_begin())


def profile_end():
    """-> {class name: dict(launches, total_ms, flops, bytes)} for the launches since profile_begin()."""
    rows = (ProfileRow * 32)()
    n = C.c_int(0)
    check(lib().dpc_profile_end(C.cast(rows, C.c_void_p), 32, C.byref(n)))
    return {rows[i].name.decode(): dict(launches=rows[i].launches, total_ms=rows[i].total_ms, flops=rows[i].flops,
                                        bytes=rows[i].bytes) for i in range(n.value)}


_P, _I, _L, _Z, _D, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double, C.c_uint64

_SIGNATURES = {
    "dpc_version": (C.c_int, []),
    "dpc_build_info": (C.c_char_p, []),
    "dpc_set_cu_budget": (C.c_int, [C.c_int]),
    "dpc_last_error": (C.c_char_p, []),
    "dpc_set_mode": (C.c_int, [C.c_char_p, C.c_char_p]),
    "dpc_get_mode": (C.c_char_p, [C.c_char_p]),
    "dpc_conv3d_algorithm": (C.c_char_p, []),
    "dpc_unet3d_modes": (C.c_char_p, [_P]),
    "dpc_unet2d_modes": (C.c_char_p, [_P]),
    "dpc_unet3d_set_range_check": (C.c_int, [_P, _I]),
    "dpc_unet3d_range_status": (C.c_int, [_P, _I, _P]),
    "dpc_train_range_status": (C.c_int, [_I, _P]),
    "dpc_train_range_poison": (C.c_int, [_P, _P]),
    "dpc_selftest_fp16_clamp": (C.c_int, [_P, _P, _I, _P]),
    "dpc_profile_begin": (C.c_int, []),
    "dpc_profile_begin_classes": (C.c_int, [C.c_char_p]),
    "dpc_profile_end": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dpc_unet3d_create": (C.c_int, [C.POINTER(Unet3DCfg), C.POINTER(_P)]),
    "dpc_unet3d_destroy": (None, [_P]),
    "dpc_unet3d_load": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(_L), _I, _P]),
    "dpc_unet3d_set_tables": (C.c_int, [_P, _I, _P, _P, _P, _P, _P]),
    "dpc_unet3d_finalize": (C.c_int, [_P]),
    "dpc_unet3d_workspace_bytes": (_Z, [_P, _I, _I, _I, _I]),
    "dpc_unet3d_forward": (C.c_int, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "dpc_unet3d_debug_taps": (C.c_int, [_P, _I]),
    "dpc_unet3d_get_tap": (C.c_int, [_P, C.c_char_p, _P, _Z, _P]),
    "dpc_ddpm_update_smoke": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(StepCoef), _I, _I, _I, _I, _I, _P]),
    "dpc_philox_normal": (C.c_int, [_P, _I, _L, _U64, _L, _L, _P]),
    "dpc_conv_workspace_bytes": (_Z, [_I, _I, _I]),
    "dpc_conv3d_cl": (C.c_int, [_P, _P, _P, _P] + [_I] * 15 + [_P, _Z, _P]),
    "dpc_convtranspose3d_144_cl": (C.c_int, [_P, _P, _P, _P] + [_I] * 6 + [_P, _Z, _P]),
    "dpc_groupnorm_workspace_bytes": (_Z, [_I, _I]),
    "dpc_groupnorm_silu_cl": (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _I, _P, _Z, _P]),
    "dpc_attention_core": (C.c_int, [_P, _P, _I, _I, _L, _L, _L, _L, _L, _P, _P, _P, _P]),
    "dpc_linear_attention_workspace_bytes": (_Z, [_L, _I]),
    "dpc_linear_attention_core": (C.c_int, [_P, _P, _I, _L, _I, _P, _Z, _P]),
    "dpc_conv_pack": (C.c_int, [_P] + [_I] * 10 + [C.c_char_p, C.POINTER(_P), _P]),
    "dpc_conv_free": (None, [_P]),
    "dpc_conv_run": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P] + [_I] * 5 + [_P, _P, _I, _I, _I, C.c_float, _I, _P]),
    "dpc_gn_workspace_bytes": (_Z, [_I, _I]),
    "dpc_conv_gn_fusable": (C.c_int, [_P, _I, _I, _I, _I]),
    "dpc_conv_gn_entries": (_L, [_I, _I]),
    "dpc_conv_run_gn": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _I, _I, _I, _P, _P, _P]),
    "dpc_gn_finalize_fused": (C.c_int, [_P, _I, _L, _I, _I, _L, _P, _P, _P, _P, _P, _P]),
    "dpc_gn_stats": (C.c_int, [_P, _P, _I, _L, _I, _I, _P, _Z, _P]),
    "dpc_gn_apply": (C.c_int, [_P] * 7 + [_I, _L, _I, _I, _P]),
    "dpc_gn_silu_bwd": (C.c_int, [_P] * 8 + [_I, _L, _I, _I, _P, _Z, _P]),
    "dpc_ln_stats": (C.c_int, [_P, _P, _L, _I, _P]),
    "dpc_ln_apply": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P]),
    "dpc_ln_bwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "dpc_linear_attention_tape_bytes": (_Z, [_L, _I]),
    "dpc_linear_attention_fwd_save": (C.c_int, [_P, _P, _I, _L, _I, _P, _Z, _P]),
    "dpc_linear_attention_bwd": (C.c_int, [_P, _P, _P, _I, _L, _I, _P, _Z, _P]),
    "dpc_attention_bwd": (C.c_int, [_P, _P, _P, _I, _L, _I, _P]),
    "dpc_upsample2x_cl": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "dpc_downsum2x_cl": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "dpc_nchw_to_cl": (C.c_int, [_P, _P, _L, _I, _I, _L, _P]),
    "dpc_cl_to_nchw": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, _I, C.c_float, _L, _P]),
    "dpc_channel_affine_to_cl": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, C.c_float, C.c_float, _L, _P]),
    "dpc_channel_mean": (C.c_int, [_P, _P, _L, _I, _I, _L, _P]),
    "dpc_channel_fill": (C.c_int, [_P, _P, _L, _I, _I, C.c_float, _L, _P]),
    "dpc_mean_rows": (C.c_int, [_P, _P, _L, _L, _I, _P]),
    "dpc_bcast_rows": (C.c_int, [_P, _P, _L, _L, _I, C.c_float, _P]),
    "dpc_add_inplace": (C.c_int, [_P, _P, _L, _P]),
    "dpc_absmax": (C.c_int, [_P, _L, _P, _P]),
    "dpc_pad_w_cl": (C.c_int, [_P, _P, _L, _I, _I, _I, _P]),
    "dpc_fold_w_cl": (C.c_int, [_P, _P, _L, _I, _I, _I, _P]),
    "dpc_small_linear": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dpc_conv3_pack": (C.c_int, [_P] + [_I] * 10 + [C.c_char_p, C.POINTER(_P), _P]),
    "dpc_conv3_run": (C.c_int, [_P, _P, _P, _I, _I, _P, _P, _P] + [_I] * 6 + [_P, _P, _I, _I, _I, C.c_float, _P]),
    "dpc_weight_range_check": (C.c_int, []),
    "dpc_stem_pack": (C.c_int, [_P, _I, _I, _I, C.c_char_p, C.POINTER(_P), _P]),
    "dpc_stem_free": (None, [_P]),
    "dpc_stem_run": (C.c_int, [_P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "dpc_conv_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _L]),
    "dpc_conv_wgrad_cl": (C.c_int, [_P, _P, _P] + [_I] * 19 + [C.c_float, C.c_float, C.c_float, _I, _P, _Z, _P]),
    "dpc_colsum_workspace_bytes": (_Z, [_I]),
    "dpc_colsum": (C.c_int, [_P, _P, _P, _P, _L, _I, C.c_float, _I, _P, _Z, _P]),
    "dpc_gn_silu_bwd_params": (C.c_int, [_P] * 10 + [_I, _L, _I, _I, _P, _Z, _P]),
    "dpc_attention_bwd_seq_workspace_bytes": (_Z, [_I, _I]),
    "dpc_attention_bwd_seq": (C.c_int, [_P, _P, _P, _P, _I, _I, _L, _L, _L, _L, _L, _P, _P, _P, _I, _P, _Z, _P]),
    "dpc_small_linear_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dpc_q_sample_smoke": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dpc_reduce_workspace_bytes": (_Z, []),
    "dpc_mse_loss_grad": (C.c_int, [_P, _P, _P, _P, _L, C.c_float, _P, _Z, _P]),
    "dpc_l2_norm": (C.c_int, [_P, _L, C.c_float, _P, _P, _Z, _P]),
    "dpc_adam_ema_step": (C.c_int, [_P, _P, _P, _P, _P, _L, _P, C.c_float, C.c_float, _D, _D, _D, _D, _I, _I,
                                    C.c_float, _P]),
    "dpc_burgers_fd": (C.c_int, [_P, _P, _P, _I, _I, _I, _D, _D, _D, _P]),
    "dpc_unet2d_create": (C.c_int, [C.POINTER(Unet2DCfg), C.POINTER(_P)]),
    "dpc_unet2d_destroy": (None, [_P]),
    "dpc_unet2d_load": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(_L), _I, _P]),
    "dpc_unet2d_set_tables": (C.c_int, [_P, _P, _P]),
    "dpc_unet2d_finalize": (C.c_int, [_P]),
    "dpc_unet2d_workspace_bytes": (_Z, [_P, _I, _I, _I]),
    "dpc_unet2d_forward": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    "dpc_unet2d_debug_taps": (C.c_int, [_P, _I]),
    "dpc_unet2d_get_tap": (C.c_int, [_P, C.c_char_p, _P, _Z, _P]),
    "dpc_burgers_prepare": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dpc_ddpm_update_burgers": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(BurgersCoef), _I, _I, _I, _P]),
    "dpc_ddpm_update_jelly": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(JellyCoef), _I, _I, _I, _I, _I, _I, _P]),
    "dpc_jelly_apply_guidance": (C.c_int, [_P, _P, _P, C.c_float, C.c_float, _I, C.c_float, _I, _I, _I, _I, _I, _P]),
    "dpc_smoke_workspace_bytes": (_Z, [_I, _I]),
    "dpc_smoke_rollout": (C.c_int, [C.POINTER(SmokeDomain), _P, _L, _P, _P, _P, _I, _I, _I, _I, _D, _D, _I,
                                    C.POINTER(SmokeOut), _P, _Z, _P]),
    "dpc_smoke_pressure_solve": (C.c_int, [C.POINTER(SmokeDomain), _P, _I, _D, _I, _P, _P, _Z, _P]),
    "dpc_smoke_advect": (C.c_int, [_P, _P, _P, _I, _D, _P]),
    "dpc_smoke_domain_tables": (C.c_int, [C.POINTER(SmokeDomain), _P, _P, _P, _P, _Z, _P]),
}


def set_mode(family, mode):
    """Process-wide arithmetic mode for U-Net handles created AFTER this call (include/dpc.h: dpc_set_mode)."""
    check(lib().dpc_set_mode(family.encode(), mode.encode()))


def get_mode(family="all"):
    return lib().dpc_get_mode(family.encode()).decode()


def create_with_mode(arithmetic, create):
    """Run `create()` (a dpc_unet*_create call) with the process-wide arithmetic mode temporarily set to `arithmetic`
    ("f16x3" | "x6" | "f32", or a dict family -> mode); the handle captures it, the process-wide setting is restored."""
    if arithmetic is None:
        return create()
    fams = ("conv", "igemm", "attn", "stem")
    saved = {f: get_mode(f) for f in fams}
    want = arithmetic if isinstance(arithmetic, dict) else {f: arithmetic for f in fams}
    try:
        for f, m in want.items():
            set_mode(f, m)
        return create()
    finally:
        for f, m in saved.items():
            set_mode(f, m)


def exported_symbols():
    """Every symbol include/dpc.h declares (checked by the CPU test-suite against the built library)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m diffphycon_amd.build` "
                "(diffphycon_amd has no non-HIP fallback)")
        L = C.CDLL(LIB_PATH)
        # an older build (or a DPC_LIB override) that lacks an entry point of include/dpc.h is refused with the rebuild message, not with
        # a bare AttributeError from the loop below; the build stamp is the first thing asked for
        absent = [name for name in _SIGNATURES if not hasattr(L, name)]
        if absent:
            raise RuntimeError(f"{LIB_PATH} does not export {', '.join(absent[:4])}{' ...' if len(absent) > 4 else ''} "
                               f"({len(absent)} of {len(_SIGNATURES)} symbols of include/dpc.h): it was built from another version of the "
                               "sources; rebuild with `python -m diffphycon_amd.build --force`")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _check_build(L)
        _lib = L
    return _lib


def _check_build(L):
    """Correctness of every kernel on a SHARED GPU (a second stream / a second process) rests on one compile flag: no packed fp32 VALU
    instructions in the device code (build.py stamps what it MEASURED on the linked code objects).  A library -- the
    in-tree one or a DPC_LIB override -- whose stamp is missing or reports such instructions is refused; DPC_ALLOW_PACKED_FP32=1 loads it anyway
    (A/B experiments of tools/) and says so."""
    info = L.dpc_build_info().decode()
    if "packed_fp32_insts=0;" in info:
        return
    msg = (f"{LIB_PATH}: the build stamp says '{info[:60]}...': the device code contains packed fp32 VALU instructions "
           "(v_pk_{mul,add,fma}_f32), with which kernels return wrong lanes when a second kernel is resident (DESIGN.md 6.2). "
           "Rebuild with `python -m diffphycon_amd.build --force`")
    if os.environ.get("DPC_ALLOW_PACKED_FP32") == "1":
        import warnings
        warnings.warn(msg + " -- loaded because DPC_ALLOW_PACKED_FP32=1")
        return
    raise RuntimeError(msg + ", or set DPC_ALLOW_PACKED_FP32=1 to load it for an experiment")


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libdpc error {rc}: {lib().dpc_last_error().decode()}")


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libdpc needs device tensors (the hot path has no CPU implementation)")
    if t.dtype != dtype:
        raise TypeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def workspace(nbytes, device):
    return torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
