"""ctypes binding of liblcrec_hip.so -- the C-ABI declared in include/lcrec.h.

There is no CPU fallback: if the library has not been built (or cannot be
loaded) every entry point raises.  Build it with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C lc-rec_amd/csrc``.
"""
import ctypes
import os

import torch  # noqa: F401  (first, so the HIP runtime torch ships is the one this library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LCREC_LIB_PATH") or os.path.join(_HERE, "csrc", "liblcrec_hip.so")   # override: diagnostic builds

_f32p = ctypes.c_void_p
_vp = ctypes.c_void_p


class LcrecError(RuntimeError):
    pass


_SIGNATURES = {
    "lcrec_version": (ctypes.c_int, []),
    "lcrec_last_error": (ctypes.c_char_p, []),
    "lcrec_linear_forward": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int,
                                            ctypes.c_int, _vp, _vp]),
    "lcrec_linear_backward_splits": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "lcrec_linear_backward_workspace": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.c_int]),
    "lcrec_linear_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp,
                                             ctypes.c_size_t, _vp]),
    "lcrec_linear_backward_weights_workspace": (ctypes.c_size_t, [_vp, ctypes.c_int]),
    "lcrec_linear_backward_weights": (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_size_t, _vp]),
    "lcrec_rq_assign_workspace": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                    ctypes.c_int]),
    "lcrec_context_create": (ctypes.c_int, [ctypes.POINTER(_vp)]),
    "lcrec_context_destroy": (ctypes.c_int, [_vp]),
    "lcrec_context_set_pipelines": (ctypes.c_int, [_vp, ctypes.c_int]),
    "lcrec_rq_assign": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_int, _vp, ctypes.c_int64, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_float,
                                       _vp, ctypes.c_size_t, _vp, _vp]),
    "lcrec_encode_assign_workspace": (ctypes.c_size_t, [ctypes.c_int64, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                                        ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "lcrec_encode_assign_chunk_rows": (ctypes.c_int64, []),
    "lcrec_encode_assign": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                           ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                           ctypes.POINTER(_vp), _vp, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                           _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, ctypes.c_size_t, _vp,
                                           _vp]),
    "lcrec_sinkhorn_assign_workspace": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int,
                                                          ctypes.POINTER(ctypes.c_int64), ctypes.c_int]),
    "lcrec_sinkhorn_assign": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_double,
                                             ctypes.c_int, _vp, ctypes.c_int64, _vp, ctypes.c_size_t, _vp, _vp, _vp]),
    "lcrec_rq_apply_level": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, ctypes.c_int, _vp,
                                            ctypes.c_int64, _vp, ctypes.c_int, _vp, _vp, _vp, ctypes.c_size_t, _vp, _vp]),
    "lcrec_code_stats": (ctypes.c_int, [_vp, ctypes.c_int64, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                        _vp, _vp, _vp]),
    "lcrec_code_stats_levels": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                               ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                               ctypes.POINTER(_vp), ctypes.c_float, ctypes.c_float, _vp]),
    "lcrec_ema_update": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                        ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "lcrec_bn_relu_forward": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp,
                                             _vp, _vp, _vp, _vp, ctypes.c_int, _vp]),
    "lcrec_bn_relu_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, _vp,
                                              _vp, _vp, _vp, _vp, _vp, _vp]),
    "lcrec_linear_bn_forward_workspace": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "lcrec_linear_bn_forward": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp,
                                               ctypes.c_int, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _vp, ctypes.c_size_t, _vp, _vp]),
    "lcrec_bn_stats": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp]),
    "lcrec_bn_relu_apply": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp]),
    "lcrec_bn_backward_reduce": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp,
                                                _vp]),
    "lcrec_bn_merge_stats": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp]),
    "lcrec_bn_backward_apply": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, ctypes.c_int, _vp, _vp,
                                               ctypes.c_float, _vp, _vp, _vp]),
    "lcrec_relu_bias_backward": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp]),
    "lcrec_train_reduce_workspace": (ctypes.c_size_t, []),
    "lcrec_recon_loss_grad": (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, ctypes.c_size_t, _vp,
                                             _vp]),
    "lcrec_grad_norm_clip": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_float, _vp, _vp, ctypes.c_size_t, _vp, _vp]),
    "lcrec_codebook_grad": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "lcrec_step_losses": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp]),
    "lcrec_quantizer_input_grad": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                                  ctypes.c_float, _vp, _vp, _vp]),
    "lcrec_quantizer_input_grad_bias": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                                       ctypes.c_float, _vp, _vp, _vp, _vp]),
    "lcrec_adamw_step": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp, _vp]),
    "lcrec_collision_groups_workspace": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int]),
    "lcrec_collision_groups": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _vp,
                                              _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "lcrec_index_json_bound": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int]),
    "lcrec_index_json_format": (ctypes.c_int64, [_vp, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, _vp, ctypes.c_int64]),
    "lcrec_trace_enable": (ctypes.c_int, [ctypes.c_int]),
    "lcrec_trace_collect": (ctypes.c_int, [_vp, ctypes.c_int]),
}


class DwProblem(ctypes.Structure):
    """lcrec_dw_problem of include/lcrec.h"""
    _fields_ = [("gy", ctypes.c_void_p), ("x", ctypes.c_void_p), ("gw", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("in_dim", ctypes.c_int), ("out_dim", ctypes.c_int), ("x_scale", ctypes.c_void_p), ("x_shift", ctypes.c_void_p),
                ("x_relu", ctypes.c_int), ("splits", ctypes.c_int)]


class TraceEntry(ctypes.Structure):
    _fields_ = [("kernel", ctypes.c_char_p), ("launches", ctypes.c_int64), ("total_ms", ctypes.c_double)]


EXPORTS = tuple(_SIGNATURES)
ABI_VERSION = 3                      # LCREC_ABI_VERSION of include/lcrec.h this binding was written against

_lib = None


def load():
    """Return the loaded library; raise LcrecError (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LcrecError(
            f"{LIB_PATH} is not built: lcrec_amd has no CPU fallback. "
            "Run `make -C lc-rec_amd/csrc` (hipcc, --offload-arch=gfx950).")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:
        raise LcrecError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.lcrec_version() != ABI_VERSION:
        raise LcrecError(f"ABI version mismatch: library {lib.lcrec_version()}, binding {ABI_VERSION} "
                         "(rebuild: make -C lc-rec_amd/csrc)")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().lcrec_last_error().decode("utf-8", "replace")
        raise LcrecError(f"{what} failed ({rc}): {msg}")
