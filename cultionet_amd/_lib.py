"""ctypes binding of libcultionet_hip.so (the C ABI declared in include/cultionet_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent,
importing / calling raises immediately (build it with ``make -C cultionet_amd/csrc`` or
``python -c "import __graft_entry__ as g; g.build()"``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_long, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CN_LIB_PATH") or os.path.join(_HERE, "csrc", "libcultionet_hip.so")  # CN_LIB_PATH: diagnostic builds

P = c_void_p  # device pointers are passed as integers (tensor.data_ptr())
I, L, F, U64 = c_int, c_long, c_float, c_ulonglong

# name -> argtypes (restype int unless named in LONG_RESULT / DOUBLE_RESULT below); mirrors include/cultionet_hip.h one to one
SIGNATURES = {
    "cn_version": [],
    "cn_conv_kpad": [I],
    "cn_conv_npad": [I],
    "cn_pack_weights_f32": [P, P, I, I, I, L, L, L, P],
    "cn_pack_weights_batched_f32": [P, I, P],
    "cn_conv2d_fwd_f32": [P, L, P, P, P, L, I, I, I, I, I, I, I, I, I, I, I, P],
    "cn_conv2d_bwd_data_f32": [P, L, P, P, L, I, I, I, I, I, I, I, I, I, I, I, P],
    "cn_conv2d_bwd_weight_f32": [P, L, P, L, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "cn_conv_set_workspace": [P, P, L],
    "cn_conv_set_autotune": [I],
    "cn_conv2d_fwd_grouped_f32": [I, P, L, P, P, P, L, I, I, I, I, I, I, I, I, P, P, I, P],
    "cn_conv2d_bwd_data_grouped_f32": [I, P, L, P, P, L, I, I, I, I, I, P, P, I, P, P, I, P],
    "cn_conv2d_bwd_weight_grouped_f32": [I, P, L, P, L, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "cn_thin_conv3x3_fwd_f32": [P, L, P, P, P, L, I, I, I, I, I, I, I, I, P, P],
    "cn_thin_conv3x3_bwd_data_f32": [P, L, P, P, L, I, I, I, I, I, I, I, I, I, P, P],
    "cn_thin_conv3x3_bwd_weight_f32": [P, L, P, L, P, I, I, I, I, I, I, I, I, P],
    "cn_conv_transpose2d_fwd_f32": [P, L, P, P, P, L, I, I, I, I, I, I, I, I, I, I, I, P],
    "cn_conv_transpose2d_bwd_data_f32": [P, L, P, P, L, I, I, I, I, I, I, I, I, I, I, I, P],
    "cn_conv_transpose2d_bwd_weight_f32": [P, L, P, L, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "cn_channel_sum_f32": [P, L, I, I, I, P, I, P],
    "cn_bn_workspace_doubles": [I],
    "cn_bn_act_fwd_f32": [P, L, P, P, P, P, P, L, P, L, P, P, P, I, I, I, I, F, F, I, P],
    "cn_bn_act_bwd_f32": [P, L, P, L, P, P, P, P, P, L, P, P, P, P, I, I, I, I, I, I, I, P],
    "cn_bn_act_group_fwd_f32": [I, P, L, P, P, P, P, P, L, P, L, P, P, P, I, I, I, I, F, F, I, I, P],
    "cn_bn_act_group_bwd_f32": [I, P, L, P, L, P, P, P, P, P, L, P, P, P, P, I, I, I, I, I, I, P],
    "cn_layernorm_c_fwd_f32": [P, L, P, P, P, L, P, L, P, P, I, I, I, F, P],
    "cn_layernorm_c_workspace_floats": [I, I, I],
    "cn_layernorm_c_bwd_f32": [P, L, P, L, P, P, P, P, L, P, P, I, I, I, I, P, L, P],
    "cn_sca_pool_fwd_f32": [P, L, I, I, I, P, P, P, P, P, P],
    "cn_sca_pool_bwd_f32": [P, P, P, P, P, P, L, I, I, I, I, P],
    "cn_sca_mlp_fwd_f32": [P, P, P, P, P, P, P, P, P, I, I, I, P],
    "cn_sca_mlp_bwd_f32": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "cn_sca_apply_fwd_f32": [P, L, P, P, P, P, L, I, I, I, P],
    "cn_sca_apply_bwd_f32": [P, L, P, L, P, P, P, P, L, I, P, P, P, P, I, I, I, P],
    "cn_na2d_fwd_f32": [P, L, P, L, P, I, I, I, I, I, I, I, F, U64, P, P],
    "cn_na2d_bwd_f32": [P, L, P, L, P, P, P, L, I, I, I, I, I, I, I, F, U64, P, P],
    "cn_convt_taps_fwd_f32": [P, L, P, P, L, I, I, I, I, I, I, I, I, I, P],
    "cn_convt_taps_bwd_f32": [P, L, P, L, I, I, I, I, I, I, I, I, I, P],
    "cn_bilinear_fwd_f32": [P, L, P, L, I, I, I, I, I, I, I, I, P],
    "cn_bilinear_bwd_f32": [P, L, P, L, I, I, I, I, I, I, I, I, I, P],
    "cn_copy_f32": [P, L, P, L, I, L, I, P],
    "cn_add_f32": [P, L, P, L, P, L, I, L, P],
    "cn_fill_f32": [P, L, F, P],
    "cn_dropout_f32": [P, L, P, L, I, I, I, F, U64, P, I, I, P],
    "cn_rng_advance_u64": [P, U64, I, P],
    "cn_adaptive_maxpool_fwd_f32": [P, L, P, L, P, I, I, I, I, I, I, P],
    "cn_adaptive_maxpool_bwd_f32": [P, L, P, P, L, I, I, I, I, I, I, I, P],
    "cn_final_combine_fwd_f32": [P, P, P, P, P, P, P, I, I, F, P],
    "cn_final_combine_bwd_f32": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, F, P],
    "cn_tanimoto_fwd_f32": [P, L, P, P, P, I, I, I, I, I, L, I, F, I, P, P, P, F, P, P],
    "cn_tanimoto_bwd_f32": [P, L, P, P, P, I, I, I, I, I, L, P, F, P, L, I, P],
    "cn_tanimoto_multi_fwd_f32": [I, P, I, L, I, F, I, P, P, P, P, P],
    "cn_tanimoto_multi_bwd_f32": [I, P, I, L, P, P],
    "cn_eval_metrics_f32": [P, P, P, P, P, I, F, L, P, P, P, P],
    "cn_grad_sumsq_f32": [P, L, P, P],
    "cn_adamw_step_f32": [P, P, P, P, L, F, F, F, F, F, I, F, P, F, P],
    "cn_pack_timeconv_f32": [P, P, I, I, I, I, I, P],
    "cn_fold_timeconv_grad_f32": [P, P, I, I, I, I, P],
    "cn_prepare_chips_f32": [P, I, P, P, P, I, I, L, F, F, F, P],
    "cn_predictions_to_u16": [P, P, P, P, I, I, I, I, I, I, I, F, P],
    # ---- bf16 NHWC mixed-precision path ----
    "cn_bconv_packed_elems": [I, I, I],
    "cn_pack_weights_bf16": [P, P, I, I, I, L, L, L, P],
    "cn_pack_weights_batched_bf16": [P, I, P],
    "cn_conv2d_fwd_bf16": [P, L, P, P, P, L, L, I, I, I, I, I, I, I, I, I, I, I, I, P, P],
    "cn_conv2d_fwd_fused_bf16": [P, L, P, P, P, L, P, L, I, I, I, I, I, I, I, I, I, I, I, P],
    "cn_pack_weights_scaled_bf16": [P, P, P, I, I, I, L, L, L, P],
    "cn_bn_fold_f32": [P, P, P, P, P, F, I, P, P, P],
    "cn_conv2d_fwd_grouped_bf16": [I, P, L, P, P, P, L, I, I, I, I, I, I, I, I, P, P, I, P, P],
    "cn_conv2d_fwd_grouped_bnstats_bf16": [I, P, L, P, P, L, I, I, I, I, I, I, I, I, P, P, P, P, P, P, P, F, F, P, L, P, P],
    "cn_conv2d_bwd_data_bf16": [P, L, P, P, L, I, I, I, I, I, I, I, I, I, I, I, P],
    "cn_conv2d_bwd_data_grouped_bf16": [I, P, L, P, P, L, I, I, I, I, I, I, I, I, P, P, I, P],
    "cn_conv_transpose2d_fwd_bf16": [P, L, P, P, P, L, I, I, I, I, I, I, I, I, I, I, P],
    "cn_conv_transpose2d_bwd_data_bf16": [P, L, P, P, L, I, I, I, I, I, I, I, I, I, I, P],
    "cn_bwgrad_workspace_floats": [I, I, I, I, I, I, I, I, I, I, I],
    "cn_conv2d_bwd_weight_bf16": [P, L, P, L, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "cn_conv_transpose2d_bwd_weight_bf16": [P, L, P, L, P, I, I, I, I, I, I, I, I, I, P, L, P],
    "cn_bn_workspace_floats_bf16": [I],
    "cn_conv2d_stats_rows_bf16": [I, I, I, I, I, I, I, I, I],
    "cn_bn_act_fwd_bf16": [P, L, P, P, P, P, P, L, P, L, P, P, P, L, I, I, F, F, I, P, I, P],
    "cn_bn_act_bwd_bf16": [P, L, P, L, P, P, P, P, P, L, P, P, P, L, I, I, I, I, P],
    "cn_bn_group_workspace_floats_bf16": [I, I],
    "cn_bn_act_group_fwd_bf16": [I, P, L, P, P, P, P, P, L, P, L, P, P, P, L, I, I, F, F, I, I, P, I, P],
    "cn_bn_act_group_bwd_bf16": [I, P, L, P, L, P, P, P, P, P, L, P, P, P, P, L, I, I, I, P],
    "cn_channel_sum_bf16": [P, L, L, I, P, I, P, P],
    "cn_layernorm_c_fwd_bf16": [P, L, P, P, P, L, P, L, L, I, F, P],
    "cn_layernorm_c_bwd_bf16": [P, L, P, L, P, P, L, P, P, L, I, F, I, P],
    "cn_convert_f32nchw_to_bf16nhwc": [P, L, P, L, I, I, I, I, P],
    "cn_convert_bf16nhwc_to_f32nchw": [P, L, P, L, I, I, I, I, P],
    "cn_copy_bf16": [P, L, P, L, L, I, I, P],
    "cn_add_bf16": [P, L, P, L, P, L, L, I, P],
    "cn_zero_bf16": [P, L, L, I, P],
    "cn_bilinear_fwd_bf16": [P, L, P, L, I, I, I, I, I, I, P],
    "cn_bilinear_bwd_bf16": [P, L, P, L, I, I, I, I, I, I, I, P],
    "cn_na2d_fwd_bf16": [P, L, P, L, P, I, I, I, I, I, I, I, F, U64, P, P],
    "cn_na2d_bwd_bf16": [P, L, P, L, P, P, P, L, I, I, I, I, I, I, I, F, U64, P, P],
    "cn_dropout_bf16": [P, L, P, L, I, I, I, F, U64, P, I, I, P],
    "cn_pretime_workspace_floats": [I, I, I, I, I, I],
    "cn_pretime_fwd_f32": [P, L, P, P, P, L, I, I, I, I, I, I, I, P, F, P, L, P],
    "cn_pretime_bwd_f32": [P, L, P, P, P, L, I, P, I, I, I, I, I, I, P, F, P, L, P],
    "cn_window_chips_f32": [P, I, P, P, I, I, I, I, I, I, I, P, P, F, F, F, P],
    "cn_stitch_predictions_u16": [P, P, P, P, P, I, I, I, I, I, I, F, P],
    "cn_profile_begin": [],
    "cn_profile_end": [P],
    "cn_profile_top": [I, P, I, P],
    "cn_profile_top_bytes": [I],
    "cn_profile_set_filter": [P],
    "cn_launch_count": [I],
    "cn_slice_sums_begin": [P, I, P, L],
    "cn_slice_sums_count": [],
    "cn_slice_sums_end": [],
    "cn_slice_sums_run": [P, P, I, I, I, P],
    "cn_stream_priority_range": [P],
    "cn_stream_create": [I, P, I, P],
    "cn_stream_destroy": [P],
}

ERRORS = {-1: "CN_ERR_ARG (invalid argument / unsupported shape)", -2: "CN_ERR_LAUNCH (HIP launch failed)",
          -3: "CN_ERR_LDS (tile does not fit the LDS budget)"}


class HipLibraryMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library. Raises HipLibraryMissing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: the cultionet_amd HIP extension is not built. "
            "Run `make -C cultionet_amd/csrc` (needs hipcc); there is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise HipLibraryMissing(f"symbol {name} missing from {LIB_PATH}; rebuild the extension") from e
        fn.argtypes = argtypes
        fn.restype = c_long if name in LONG_RESULT else (c_double if name in DOUBLE_RESULT else c_int)
    _lib = lib
    return lib


def call(name: str, *args) -> int:
    """Call an entry point that returns a status code; raise on error."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise HipKernelError(f"{name} failed: {ERRORS.get(rc, rc)}")
    return rc


LONG_RESULT = {"cn_launch_count", "cn_bconv_packed_elems", "cn_bwgrad_workspace_floats", "cn_bn_workspace_floats_bf16",
               "cn_bn_group_workspace_floats_bf16", "cn_pretime_workspace_floats"}

DOUBLE_RESULT = {"cn_profile_top_bytes"}


def query(name: str, *args) -> int:
    """Call an entry point that returns a value (cn_version, cn_conv_kpad, ...)."""
    return getattr(load(), name)(*args)
