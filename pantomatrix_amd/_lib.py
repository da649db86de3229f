"""ctypes binding of libemage_hip.so (include/emage_hip.h).  There is NO fallback: if the library is
missing or a symbol is absent, importing the ops fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libemage_hip.so")
TOOLS_LIB_PATH = os.path.join(_HERE, "csrc", "libemage_hip_tools.so")   # -DEMAGE_TOOLS twin: every tile configuration + emage_set_tuning

F32, BF16, F16X3, H2 = 0, 1, 2, 3
ABI_VERSION = 17

_p, _i, _f, _l = C.c_void_p, C.c_int, C.c_float, C.c_long

class GemmProblem(C.Structure):
    """`emage_gemm_problem` of include/emage_hip.h: one emage_gemm call's arguments (emage_gemm_grouped takes an array of them)."""
    _fields_ = ([(n, _p) for n in ("A", "W", "bias", "slope", "res", "out", "out_f32", "out_t")]
                + [(n, _i) for n in ("lda", "ldr", "res_is_f32", "res_first", "ldo", "n_store", "ldf", "t_col0", "t_rows", "t_ld",
                                     "M", "N", "Cp", "taps", "stride", "pad", "Lin", "Lout")]
                + [("a_scale", _f), ("w_scale", _f)]
                + [(n, _p) for n in ("ln_stats", "ln_c", "rs_stats", "rs_gamma", "rs_beta", "st_out")]       # the LayerNorm fold (round 6; all NULL = off)
                + [("ln_np", _i), ("rs_np", _i), ("ln_eps", _f)]
                + [("sk_ws", _p), ("sk_ws_bytes", _l), ("sk_count", _p), ("sk_tiles", _i)])             # split-K fix-up scratch (round 6; NULL = off)


class FinalizeEntry(C.Structure):
    """`emage_finalize_entry` of include/emage_hip.h (emage_col_sum_finalize_multi takes a host array of them)."""
    _fields_ = [("partial", _p), ("out", _p), ("chunks", _i), ("C", _i), ("accumulate", _i)]


# name -> argtypes, exactly the prototypes of include/emage_hip.h
TOOLS_SIGNATURES = {"emage_set_tuning": [_i, _i], "emage_h2_set_trace": [_p]}      # exported by the tools build only
SIGNATURES = {
    "emage_vq_argmin_f32": [_p, _i, _p, _p, _i, _l, _i, _i, _i, _p],
    "emage_argmax_logsoftmax_f32": [_p, _i, _p, _i, _l, _i, _i, _p],
    "emage_gather_rows": [_p, _p, _i, _l, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "emage_gemm": [_i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i,
                   _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p],
    "emage_gemm_ws": [_i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i,
                      _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p, C.c_size_t, _p],
    "emage_gemm_grouped": [_i, C.POINTER(GemmProblem), _i, _p],
    "emage_gemm_grouped_launches": [_i, C.POINTER(GemmProblem), _i],
    "emage_wav_conv_in": [_i, _p, _l, _i, _i, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "emage_conv_slab": [_i, _p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _f, _p],
    "emage_wav_block0": [_i, _p, _l, _i, _i, _l, _i, _p, _p, _f, _p, _p, _i, _i, _i, _p, _p, _p, _i, _i, _p, _i, _i, _i, _f, _f, _p],
    "emage_attention": [_i, _p, _i, _p, _i, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "emage_attention_dropout": [_i, _p, _i, _p, _i, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "emage_bn_stats_workspace_bytes": [_i, _i],
    "emage_bn_stats": [_p, _i, _i, _i, _p, _l, _p, _p, _p, _p, _f, _p],
    "emage_bn_apply": [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _f, _f, _p, _i, _i, _i, _p],
    "emage_mse_loss": [_p, _i, _p, _i, _i, _i, _f, _p, _p, _p],
    "emage_nll_loss": [_p, _i, _p, _i, _i, _f, _p, _p, _p],
    "emage_transpose_f32": [_p, _i, _p, _i, _i, _i, _p],
    "emage_col_sum": [_p, _i, _p, _i, _i, _i, _p, _i, _p, _l, _p],
    "emage_col_sum_chunks": [_i],
    "emage_col_sum_finalize_multi": [C.POINTER(FinalizeEntry), _i, _p],
    "emage_act_backward": [_p, _i, _p, _i, _f, _p, _i, _i, _i, _p],
    "emage_layernorm_backward_affine_workspace_bytes": [_i, _i, _i],
    "emage_layernorm_backward_affine": [_p, _i, _p, _p, _i, _f, _p, _i, _p, _p, _i, _i, _i, _i, _p, _l, _p],
    "emage_grad_prep": [_p, _i, _p, _i, _f, _i, _i, _f, _p, _i, _i, _p, _i, _i, _p, _i, _p, _l, _p],
    "emage_layernorm_backward": [_p, _i, _p, _p, _i, _f, _p, _i, _p, _i, _i, _i, _p],
    "emage_attention_backward": [_p, _i, _p, _i, _p, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "emage_mse_loss_grad": [_p, _i, _p, _i, _i, _i, _f, _p, _i, _p],
    "emage_nll_loss_grad": [_p, _i, _p, _i, _i, _f, _p, _i, _p],
    "emage_im2col_t": [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _l, _p],
    "emage_im2col_t_h2": [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _l, _p],
    "emage_col2im": [_p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    "emage_bn_backward": [_p, _i, _p, _p, _p, _f, _p, _i, _p, _i, _p, _p, _i, _i, _p, _l, _p],
    "emage_bn_backward_sums": [_p, _i, _p, _p, _f, _p, _i, _p, _p, _i, _i, _p, _l, _p],
    "emage_bn_backward_apply": [_p, _i, _p, _p, _p, _f, _p, _i, _p, _p, _l, _p, _i, _i, _i, _p],
    "emage_wav_conv_in_backward_workspace_bytes": [_i, _i, _i],
    "emage_wav_conv_in_backward": [_p, _i, _p, _l, _i, _i, _i, _i, _i, _i, _i, _p, _p, _l, _p],
    "emage_count_nonfinite": [_p, _l, _p, _p],
    "emage_count_nonfinite_multi": [C.POINTER(_p), C.POINTER(_l), _i, _p, _p],
    "emage_adam_step_dev": [_p, _p, _p, _p, _l, _p, _f, _f, _f, _f, _f, _p],
    "emage_adam_step": [_p, _p, _p, _p, _l, _i, _f, _f, _f, _f, _f, _p],
    "emage_adam_multi_chunk": [],
    "emage_adam_multi": [_p, _p, _p, _i, _p, _i, _f, _f, _f, _f, _f, _f, _i, _p, _p],
    "emage_dropout_mask": [_p, _l, _f, C.c_ulonglong, C.c_uint, _p, _i, _p],
    "emage_mul_add": [_p, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i, _p],
    "emage_mul_add_philox": [_p, _i, _f, C.c_ulonglong, C.c_uint, _p, _i, _i, _p, _i, _p, _i, _i, _i, _p],
    "emage_layernorm": [_i, _p, _i, _p, _p, _f, _p, _i, _p, _p, _i, _i, _i, _p],
    "emage_add": [_i, _p, _i, _p, _i, _i, _p, _i, _i, _i, _p, _p, _i, _i, _i, _p],
    "emage_pack_motion": [_i, _p, _p, _l, _p, _p, _l, _i, _p, _i, _i, _i, _i, _i, _p],
    "emage_cast_pad": [_i, _p, _i, _p, _i, _i, _i, _i, _p],
    "emage_h2_cast": [_p, _i, _p, _i, _i, _i, _i, _f, _i, _p],
    "emage_f16x3_pack_weights": [_p, _i, _p, _i, _i, _i, _f, _p],
    "emage_rot6d_to_axis_angle": [_p, _p, _i, _p],
    "emage_axis_angle_to_rot6d": [_p, _p, _i, _p],
    "emage_merge_parts": [_p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _p],
    "emage_velocity_to_position": [_p, _i, _i, _p, _i, _f, _p, _i, _i, _p],
    "emage_lstm_step": [_i, _p, _i, _p, _f, _f, _p, _i, _p, _i, _p, _i, _i, _i, _p],
    "emage_lstm_step_pair": [_i, _p, _p, _i, _i, _p, _p, _f, _f, _f, _p, _p, _i, _p, _p, _i, _p, _p, _i, _i, _i, _p],
    "emage_lstm_layer_sync_words": [_i, _i],
    "emage_lstm_layer_health": [_p, _i, _p, _p],
    "emage_lstm_layer": [_i, _p, _l, _i, _p, _p, _f, _f, _f, _p, _l, _i, _i, _i, _i, _p, _i, _p],
    "emage_softmax2_mix": [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _p],
    "emage_lstm_inputs": [_p, _p, _i, _p, _l, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    "emage_rot6d_scatter": [_p, _i, _p, _p, _i, _i, _p],
}
RESTYPES = {"emage_bn_stats_workspace_bytes": _l, "emage_wav_conv_in_backward_workspace_bytes": _l, "emage_layernorm_backward_affine_workspace_bytes": _l}

_lib = None
_tools = None
_use_tools = False


def _open(path, signatures):
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). The EMAGE path has no CPU fallback.")
    lib = C.CDLL(path)
    lib.emage_abi_version.restype = _i
    lib.emage_target_arch.restype = C.c_char_p
    if lib.emage_abi_version() != ABI_VERSION:
        raise ImportError(f"{os.path.basename(path)} ABI {lib.emage_abi_version()} != expected {ABI_VERSION}; rebuild")
    for name, args in signatures.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = RESTYPES.get(name, _i)
    return lib


def load():
    """The shared library every op launches into (loaded once, every entry point typed): the product library, or — after
    `use_tools()` — its tools twin."""
    global _lib, _tools
    if _use_tools:
        if _tools is None:
            _tools = _open(TOOLS_LIB_PATH, {**SIGNATURES, **TOOLS_SIGNATURES})
        return _tools
    if _lib is None:
        _lib = _open(LIB_PATH, SIGNATURES)
    return _lib


def use_tools(on: bool = True):
    """tools/ and the tile-configuration tests: route every launch of this process through libemage_hip_tools.so (all tile
    configurations, `emage_set_tuning`, ablation branches, phase tracer).  The product path never calls this."""
    global _use_tools
    _use_tools = bool(on)
    return load()


class EmageKernelError(RuntimeError):
    pass


def check(code: int, what: str):
    if code != 0:
        kind = "unsupported argument (EMAGE_EINVAL)" if code < 0 else f"hipError {code}"
        raise EmageKernelError(f"{what}: {kind}")
