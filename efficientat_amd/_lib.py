"""ctypes binding of libeat_hip.so (the C ABI declared in include/eat_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an
exception is raised.  Build it with ``python -m efficientat_amd.build`` (or
``__graft_entry__.build()``).
"""
import ctypes
import os

import torch  # noqa: F401  -- must be imported BEFORE libeat_hip.so is loaded: the kernels have to run on
#                torch's own HIP runtime (libamdhip64 bundled with the wheel) to share its streams.
#                Loading our library first binds it to /opt/rocm's copy, which sees no device.

_HERE = os.path.dirname(os.path.abspath(__file__))
# EAT_LIB: another build of the same library (A/B of kernel changes; must export the same symbols)
LIB_PATH = os.environ.get("EAT_LIB") or os.path.join(_HERE, "libeat_hip.so")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_D = ctypes.c_double

# name -> argtypes (restype is always int unless noted); must mirror include/eat_hip.h
SIGNATURES = {
    "eat_mel_fwd": [_P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _P],
    "eat_stem_conv_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_pw_prepack": [_P, _P, _P, _I, _I, _P],
    "eat_pw_conv_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_linear_fwd": [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "eat_bn_stats": [_P, _I, _I, _I, _P, _P],
    "eat_bn_finalize": [_P, _P, _P, _P, _P, _F, _F, _D, _I, _P, _P, _P, _P, _P],
    "eat_bn_act_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_bn_act_bwd_reduce": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "eat_bn_act_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_plane_dot": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dw_conv_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_wgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_pw_conv_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_ctx_pool": [_P, _P, _I, _I, _I, _I, _P],
    "eat_dyn_aggregate": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dyn_pw_pack": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_pw_conv_dyn_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_dyn_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_dyn_act_fwd": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_fused_expand_dw_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_mbconv_fwd": [_P] * 9 + [_I] * 11 + [_P],
    "eat_front_fwd": [_P] * 8 + [_I] * 7 + [_P],
    "eat_block_fused_supported": [_I] * 9,
    "eat_dw_conv_fwd_tf": [_P, _P, _P, _I, _P, _P, _P] + [_I] * 8 + [_P],
    "eat_dw_conv_wgrad_tf": [_P, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_expand_dw_bf16_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_pw_prepack_bf16": [_P, _P, _P, _I, _I, _I, _P],
    "eat_pw_conv_bf16_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eat_ctx_pool_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dyrelu_ca_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dyrelu_ca_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dyn_bank_grad": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "eat_pw_conv_dyn_wgrad": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dw_conv_dyn_wgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_dyn_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "eat_mixup_fwd": [_P, _P, _P, _P, _I, _I, _P],
    "eat_dyn_heads_fwd": [_P, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P, _P],
    "eat_dyn_heads_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P],
    "eat_wave_i16_to_f32": [_P, _P, ctypes.c_longlong, _F, _P],
    "eat_col_sum": [_P, _P, _I, _I, _P],
    "eat_calib_copy": [_P, _P, ctypes.c_longlong, _I, _P],
    "eat_cast_b16": [_P, _P, ctypes.c_longlong, _P],
    "eat_pw_conv_kcat_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_dilated_fwd": [_P, _P, _P, _P, _P] + [_I] * 10 + [_P],
    "eat_dw_conv_dilated_dgrad": [_P, _P, _P] + [_I] * 9 + [_P],
    "eat_dw_conv_dilated_wgrad": [_P, _P, _P] + [_I] * 9 + [_P],
    "eat_kd_loss_fwd_bwd": [_P, _P, _P, _P, _P, _P, _I, _F, _I, _I, _P, _P, _P],
    "eat_pw_wgrad_slots": [_I] * 6,
    "eat_pw_wgrad_kernel_kind": [_I] * 8,
    "eat_pw_conv_wgrad_ws": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eat_pw_conv_tf_fwd": [_P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_pw_conv_cat_fwd": [_P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_pw_conv_wgrad_tf": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eat_pw_prepack_t": [_P, _P, _P, _I, _I, _P],
    "eat_pw_prepack_bf16_t": [_P, _P, _P, _I, _I, _I, _P],
    "eat_dw_partials_inner": [_I] * 7,
    "eat_dw_conv_fwd_stats": [_P, _P, _P, _I, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_bn_stats_partial": [_P, _I, _I, _I, _P, _P],
    "eat_bn_finalize_partials": [_P, _I, _I, _I, _P, _P, _P, _P, _F, _F, _D, _P, _P, _P, _P, _P, _P],
    "eat_bn_finalize_ws_doubles": [_I, _I, _I],
    "eat_pw_conv_gstats_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    "eat_bn_bwd_sums_from_tiles": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "eat_bn_bwd_sums_ws_doubles": [_I, _I],
    "eat_gram_bn_finalize": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _F, _D, _P, _P, _P, _P, _I, _P],
    "eat_gram_centered": [_P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_act_grad_sum": [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _P],
    "eat_dw_conv_dgrad_g": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_dw_bwd_partials_inner": [_I] * 6,
    "eat_dw_conv_bwd_g": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_se_bn_bwd_partials": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_se_bn_bwd_combine": [_P, _P, _P, _P, _I, _I, _P, _P],
    "eat_expand_bwd_coef": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _D, _I] + [_P] * 7 + [_I, _P, _P],
    "eat_expand_bwd_wcat": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "eat_expand_bwd_wcat_elems": [_I, _I, _I],
    "eat_stem_gram_blocks": [_I, _I],
    "eat_gram_bn_finalize_g": [_P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _F, _D, _P, _P, _P, _P, _P, _I, _P],
    "eat_pw_prepack_multi": [_P, _I, _I, _P],
    "eat_se_mlp_bwd": [_P] * 6 + [_F] + [_P] * 6 + [_I, _I, _I, _P],
    "eat_dw_bwd_merged_ok": [_I] * 8,
    "eat_dw_conv_bwd_bn_g": [_P] * 9 + [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_stem_gram": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_stem_bwd_blocks": [_I, _I],
    "eat_dyn_pw_pack_bf16": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eat_pw_conv_dyn_bf16_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_pw_dyn_wgrad_accumulates": [_I, _I, _I],
    "eat_pw_conv_stat_tiles": [_I, _I, _I],
    "eat_pw_conv_stats_fwd": [_P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dyn_pw_pack_t": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dyn_pw_pack_bf16_t": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eat_ctx_pool_cm": [_P, _P, _I, _I, _I, _I, _P],
    "eat_ctx_pool_cm_bwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "eat_ctx_split": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_ctx_split_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_dyrelu_ca_fwd2": [_P] * 7 + [_I, _I, _I, _I, _P],
    "eat_dyrelu_ca_bwd2": [_P] * 12 + [_I, _I, _I, _I, _P],
    "eat_bn_bwd_combine_partials": [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    "eat_dw_conv_dyn_fwd_stats": [_P, _P, _P, _I, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_dw_conv_dyn_bwd_bn_g": [_P] * 7 + [_I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_stem_bwd": [_P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_mlp_head_bwd": [_P] * 13 + [_I, _I, _I, _I, _P],
    "eat_se_mlp_dh_floats": [_I, _I, _I],
    "eat_mlp_head_dfeat_floats": [_I, _I, _I],
    # bf16 activation storage (BASELINE configs[2])
    "eat_pw_conv_b16_fwd": [_P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_b16_ok": [_I] * 8,
    "eat_dw_conv_fwd_stats_b16": [_P, _I, _P, _P, _I, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_bn_act_fwd_b16": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P],
    "eat_bn_act_bwd_reduce_b16": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P],
    "eat_bn_act_bwd_apply_b16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_se_bn_bwd_partials_b16": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "eat_dw_conv_bwd_bn_g_b16": [_P] * 9 + [_I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_pw_wgrad_b16_slots": [_I] * 5,
    "eat_pw_conv_wgrad_b16": [_P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    # bf16 activation storage of the DyMN blocks (BASELINE configs[3] on the SURVEY 8(d) byte contract)
    "eat_dyn_pw_pack_b16": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "eat_pw_conv_dyn_b16_fwd": [_P, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    "eat_dw_conv_dyn_fwd_stats_b16": [_P, _I, _P, _P, _I, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_dyrelu_ca_fwd2_b16": [_P] * 7 + [_I, _I, _I, _I, _P],
    "eat_dyrelu_ca_bwd2_b16": [_P] * 12 + [_I, _I, _I, _I, _P],
    "eat_dw_conv_dyn_bwd_bn_g_b16": [_P] * 7 + [_I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P] + [_I] * 8 + [_P],
    "eat_bn_bwd_apply_b16": [_P] * 8 + [_I, _I, _I, _I, _P],
    "eat_adam_multi": [_P, _I, _P, _D, _P, _F, _D, _D, _D, _D, _I, _D, _P],
    "eat_pw_dyn_wgrad_b16_slices": [_I] * 5,
    "eat_pw_conv_dyn_wgrad_b16": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P],
}

_lib = None


class EatHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EatHipError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU/PyTorch fallback). "
                "Build it with `python -m efficientat_amd.build`.")
        h = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = _I
        h.eat_version.restype = _I
        h.eat_pw_stream_mode.argtypes = [_I]
        h.eat_pw_stream_mode.restype = _I
        h.eat_last_error_string.restype = ctypes.c_char_p
        _lib = h
    return _lib


def call(name, *args):
    h = lib()
    rc = getattr(h, name)(*args)
    if rc != 0:
        raise EatHipError(f"{name} failed ({rc}): {h.eat_last_error_string().decode()}")


def call_rc(name, *args):
    """Like `call` for the entry points that answer 1 = "not applicable, nothing launched" (the caller takes another path)."""
    h = lib()
    rc = getattr(h, name)(*args)
    if rc not in (0, 1):
        raise EatHipError(f"{name} failed ({rc}): {h.eat_last_error_string().decode()}")
    return rc


def exported_symbols():
    return list(SIGNATURES) + ["eat_version", "eat_last_error_string", "eat_pw_stream_mode"]
