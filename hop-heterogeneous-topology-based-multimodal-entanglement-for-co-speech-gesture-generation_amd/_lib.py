"""ctypes binding of libhopmi.so (C ABI: include/hopmi.h).

There is no fallback: if the shared library is missing or a call fails, the op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("HOPMI_LIB") or os.path.join(_HERE, "libhopmi.so")      # (HOPMI_LIB: a probe build of the same ABI)
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.c_int
_VP = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/hopmi.h declares
SIGNATURES = {
    "hopmi_version": (ctypes.c_char_p, []),
    "hopmi_last_error": (ctypes.c_char_p, []),
    "hopmi_reload_env": (None, []),
    "hopmi_stream_capture_status": (_I, [_VP, ctypes.POINTER(ctypes.c_int)]),
    "hopmi_stream_capture_abandon": (_I, [_VP]),
    "hopmi_gcn_prep_floats": (ctypes.c_size_t, [_I]),
    "hopmi_gcn_prepare": (_I, [_VP, _VP, _VP, _I, _VP]),
    "hopmi_gcn_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hopmi_gcn_bwd_ws_floats": (ctypes.c_size_t, [_I, _I]),
    "hopmi_gcn_bwd": (_I, [_VP] * 10 + [_I, _I, _VP]),
    "hopmi_wn_layer_ws_floats": (ctypes.c_size_t, [_I, _I, _I, _I]),
    "hopmi_wn_weight_image_bytes": (ctypes.c_size_t, [_I]),
    "hopmi_wn_prepare_weights": (_I, [_VP, _VP, _VP, _I, _VP, _VP]),
    "hopmi_wn_layer_fwd": (_I, [_VP] * 10 + [_I] + [_VP] + [_I] * 5 + [_VP]),
    "hopmi_wn_bn_replay": (_I, [_VP] * 3 + [ctypes.c_float, _VP]),
    "hopmi_wn_bn_finalize": (_I, [_VP] * 5 + [ctypes.c_float, ctypes.c_float] + [_VP] * 2 + [_I] * 4 + [_VP]),
    "hopmi_wn_stack_grid": (_I, [_I, _I, _I, _VP, _I]),
    "hopmi_wn_stack_ws_bytes": (ctypes.c_size_t, [_I, _I, _I, _VP, _I]),
    "hopmi_wn_stack_fwd": (_I, [_VP] * 10 + [ctypes.c_float, ctypes.c_float] + [_VP] * 2 + [_I] + [_VP] * 3 + [_I] * 3 + [_VP, _I, _VP]),
    "hopmi_gcn_fwd_dt": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "hopmi_gcn_bwd_dt": (_I, [_VP] * 10 + [_I, _I, _I, _VP]),
    "hopmi_wn_layer_fwd_dt": (_I, [_VP] * 10 + [_I] + [_VP] + [_I] * 6 + [_VP]),
    "hopmi_wn_stack_fwd_dt": (_I, [_VP] * 10 + [ctypes.c_float, ctypes.c_float] + [_VP] * 2 + [_I] + [_VP] * 3 + [_I] * 3 + [_VP, _I, _I, _VP]),
    "hopmi_wn_layer_bwd_dt": (_I, [_VP] * 9 + [_I] + [_VP] * 3 + [_I] + [_VP] * 11 + [_I] + [_VP] * 4 + [_I] * 6 + [_VP]),
    "hopmi_wn_layer_bwd_ws_floats": (ctypes.c_size_t, [_I, _I, _I, _I]),
    "hopmi_wn_layer_bwd": (_I, [_VP] * 9 + [_I] + [_VP] * 3 + [_I] + [_VP] * 11 + [_I] + [_VP] * 4 + [_I] * 5 + [_VP]),
    "hopmi_hop_losses_ws_floats": (ctypes.c_size_t, [_I]),
    "hopmi_hop_losses_fwd": (_I, [_VP] * 7 + [_I] * 3 + [ctypes.c_float] * 3 + [_VP] * 3),
    "hopmi_hop_losses_bwd": (_I, [_VP] * 7 + [_I] * 3 + [ctypes.c_float] * 2 + [_VP] * 4),
    "hopmi_colsum_ws_floats": (ctypes.c_size_t, [_I, _I]),
    "hopmi_colsum": (_I, [_VP, _I, _I, _I, _VP, _VP, _VP]),
    "hopmi_reprog_attn_ws_bytes": (ctypes.c_size_t, [_I, _I, _I]),
    "hopmi_reprog_attn_fwd": (_I, [_VP] * 6 + [_I] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_reprog_attn_fwd_dt": (_I, [_VP] * 4 + [_I] + [_VP] * 2 + [_I] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_reprog_attn_bwd_splits": (_I, []),
    "hopmi_reprog_attn_bwd_ws_bytes": (ctypes.c_size_t, [_I, _I, _I, _I]),
    "hopmi_reprog_attn_bwd": (_I, [_VP] * 10 + [_I] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_reprog_attn_bwd_dt": (_I, [_VP] * 4 + [_I] + [_VP] * 6 + [_I] * 4 + [ctypes.c_float, ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_time_next_launch": (_I, [_VP, _VP]),
    "hopmi_noop_launch": (_I, [_VP]),
    "hopmi_bert_attn_fwd": (_I, [_VP] * 2 + [_I] * 3 + [ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_bert_attn_bwd": (_I, [_VP] * 3 + [_I] * 3 + [ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_bias_gelu_fwd": (_I, [_VP] * 3 + [_I, _I, _VP]),
    "hopmi_bias_gelu_bwd": (_I, [_VP] * 4 + [_I, _I, _VP]),
    "hopmi_bias_dropout_residual_layernorm_fwd": (_I, [_VP] * 3 + [_I] + [_VP] * 5 + [_I, _I, ctypes.c_float, ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_bias_dropout_residual_layernorm_bwd": (_I, [_VP] * 6 + [_I, _I, ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_logmel": (_I, [_VP, _I, _I, _I] + [_VP] * 7),
    "hopmi_bias_gelu_fwd_dt": (_I, [_VP, _VP, _VP, _I, _I, _I, _VP]),
    "hopmi_bias_gelu_bwd_dt": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "hopmi_bias_dropout_residual_layernorm_fwd_dt": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, ctypes.c_float,
                                                         ctypes.c_float, ctypes.c_uint, _VP, _I, _VP]),
    "hopmi_bias_dropout_residual_layernorm_bwd_dt": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, ctypes.c_float, ctypes.c_uint, _VP, _I, _VP]),
    "hopmi_bias_dropout_residual_layernorm_fwd_rs": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, ctypes.c_float,
                                                         ctypes.c_float, ctypes.c_uint, _VP, _I, _VP]),
    "hopmi_bias_dropout_residual_layernorm_bwd_rs": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, ctypes.c_float, ctypes.c_uint, _VP, _I, _VP]),
    "hopmi_bias_dropout_residual_layernorm_fwd_im": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, ctypes.c_float,
                                                         ctypes.c_float, ctypes.c_uint, _VP, _I, _VP]),
    "hopmi_bias_dropout_residual_layernorm_bwd_im": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, ctypes.c_float, ctypes.c_uint, _VP,
                                                         _I, _VP]),
    "hopmi_gemm_f16x2_ab_img": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP, _VP, ctypes.c_float, ctypes.c_float, _VP]),
    "hopmi_bert_attn_fwd_dt": (_I, [_VP, _VP, _I, _I, _I, ctypes.c_float, ctypes.c_uint, _VP, _I, _VP]),
    "hopmi_bert_attn_fwd_im": (_I, [_VP, _VP, _VP, _I, _I, _VP, _VP, _I, _I, _I, ctypes.c_float, ctypes.c_uint, _VP, _VP]),
    "hopmi_bert_attn_bwd_dt": (_I, [_VP, _VP, _VP, _I, _I, _I, ctypes.c_float, ctypes.c_uint, _VP, _I, _VP]),
    "hopmi_bn_cl_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, ctypes.c_float, ctypes.c_float, _I, _VP]),
    "hopmi_bn_cl_bwd": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "hopmi_gemm_split_image_bytes": (ctypes.c_size_t, [_I, _I, _I]),
    "hopmi_gemm_split_prepare": (_I, [_VP, _I, _I, _I, _VP, _VP]),
    "hopmi_gemm_split": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "hopmi_gemm_split_ab": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "hopmi_row_scales": (_I, [_VP, _I, _I, _VP, _VP]),
    "hopmi_gemm_f16x2_image_bytes": (ctypes.c_size_t, [_I, _I]),
    "hopmi_gemm_f16x2_prepare": (_I, [_VP, _I, _I, _VP, _VP]),
    "hopmi_rows_image_f16_bytes": (ctypes.c_size_t, [_I, _I]),
    "hopmi_rows_image_f16": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "hopmi_gemm_f16x2_ab": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "hopmi_gemm_f16x2_ab_ep": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "hopmi_gemm_f16x2_tiles_n": (_I, [_I]),
    "hopmi_gemm_f16x2_ab_splitk_ws_floats": (ctypes.c_size_t, [_I, _I, _I]),
    "hopmi_gemm_f16x2_ab_splitk": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP]),
    "hopmi_gemm_f16x2_tn_ws_floats": (ctypes.c_size_t, [_I, _I, _I, _I]),
    "hopmi_gemm_f16x2_tn": (_I, [_VP, _I, ctypes.c_longlong, _VP, _VP, _I, ctypes.c_longlong, _VP, _VP, _I, ctypes.c_longlong, _VP, _I, _I, _I, _I, _I,
                                _VP]),
    "hopmi_gemm_f16x2_tn_cs": (_I, [_VP, _I, ctypes.c_longlong, _VP, _VP, _I, ctypes.c_longlong, _VP, _VP, _I, ctypes.c_longlong, _VP, _I, _I, _I, _I, _I,
                                   _VP, _VP]),
    "hopmi_gemm_f16x2": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "hopmi_gemm_split_ep": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "hopmi_gru_ws_bytes": (ctypes.c_size_t, [_I, _I, _I]),
    "hopmi_gru_fwd": (_I, [_VP] * 6 + [_I, _I, _I, _VP]),
    "hopmi_gru_fwd_dt": (_I, [_VP, _I] + [_VP] * 5 + [_I, _I, _I, _VP]),
    "hopmi_gru_fwd_pair_dt": (_I, [_VP, _VP, _I, _I] + [_VP] * 5 + [_I, _I, _I, _VP]),
    "hopmi_gru_bwd_dt": (_I, [_VP] * 5 + [_I] + [_VP] * 3 + [_I, _I, _I, _VP]),
    "hopmi_gru_bwd_operands": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "hopmi_gru_bwd_ws_floats": (ctypes.c_size_t, [_I, _I]),
    "hopmi_gru_bwd": (_I, [_VP] * 8 + [_I, _I, _I, _VP]),
    "hopmi_adam_chunk": (_I, []),
    "hopmi_adam_multi": (_I, [_VP, _VP, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _VP, _VP]),
}


class HopmiError(RuntimeError):
    pass


def lib():
    """The loaded library; raises HopmiError (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HopmiError(
                f"hopmi: {_LIB_PATH} is missing -- the HIP kernels are not built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no CPU / eager fallback for the hot path.")
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise HopmiError(f"hopmi: {what} failed (code {rc}): {lib().hopmi_last_error().decode()}")


def version() -> str:
    return lib().hopmi_version().decode()
