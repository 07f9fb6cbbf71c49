"""ctypes binding of libfrlw_evd.so (the C-ABI in include/frlw_evd.h).

There is no fallback: if the HIP library is missing or fails to load, every product entry point
raises.  ``__graft_entry__.build()`` (or ``python frlw-evd_amd/_build.py``) produces the library.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

FRLW_OK = 0
FRLW_ERR_ARG = -1
FRLW_ERR_INDEX = -2
FRLW_ERR_WORKSPACE = -3
FRLW_ERR_HIP = -4
FRLW_ERR_POLARITY = -5
FRLW_ERR_UNSUPPORTED = -6
FRLW_ERR_SPAN = -7
MAX_SEQUENCES = 64

LAYOUT_XYTP_F64 = 0
LAYOUT_DAT8 = 1
TAF_U8_FLIP_K = 1

_ERR_TEXT = {
    FRLW_ERR_ARG: "bad argument",
    FRLW_ERR_INDEX: "event coordinate out of range",
    FRLW_ERR_WORKSPACE: "workspace too small",
    FRLW_ERR_HIP: "HIP runtime error",
    FRLW_ERR_POLARITY: "polarity outside {0, 1}",
    FRLW_ERR_UNSUPPORTED: "unsupported size",
    FRLW_ERR_SPAN: "event outside the encode span",
}


class FrlwTuning(C.Structure):
    """frlw_tuning_t: per-call overrides of the launch heuristics, every field < 0 = the library's choice."""
    _fields_ = [("struct_size", C.c_int32),
                ("tile_width_log2", C.c_int32), ("batches_per_wave", C.c_int32), ("hot_tile_records", C.c_int32),
                ("staged_scatter", C.c_int32), ("quarter_below", C.c_int32), ("no_value_table", C.c_int32),
                ("taf_tile_walk", C.c_int32), ("direct_bins", C.c_int32), ("chunk_major", C.c_int32),
                ("ev_lds_float_atomics", C.c_int32), ("walk_window_table", C.c_int32)]

    def __init__(self, **kw):
        super().__init__(*[int(kw.pop(name, -1)) for name, _ in self._fields_])
        self.struct_size = C.sizeof(type(self))  # the library reads only what lies inside (frlw_evd.h)
        if kw:
            raise TypeError(f"unknown tuning fields {sorted(kw)}")


class FrlwBaseconvFuse(C.Structure):
    """frlw_baseconv_fuse_t: what the blocks around a training-mode BaseConv fold into its launches (residual added to y, y
    written into a channel slice, another consumer's gradient added to dx)."""
    _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32), ("residual", C.c_void_p), ("residual_row_stride", C.c_int64),
                ("y_row_stride", C.c_int64), ("dx_add", C.c_void_p), ("dx_add_row_stride", C.c_int64),
                ("split", C.c_int32), ("reserved2", C.c_int32), ("w2", C.c_void_p), ("gamma2", C.c_void_p), ("beta2", C.c_void_p),
                ("running_mean2", C.c_void_p), ("running_var2", C.c_void_p), ("num_batches_tracked2", C.c_void_p),
                ("y2", C.c_void_p), ("y2_row_stride", C.c_int64), ("dy2", C.c_void_p), ("dy2_row_stride", C.c_int64)]

    def __init__(self, **kw):
        super().__init__()
        self.struct_size = C.sizeof(type(self))
        for k, v in kw.items():
            if k not in dict(self._fields_) or k == "struct_size":
                raise TypeError(f"unknown field {k}")
            setattr(self, k, v if (v is None or dict(self._fields_)[k] is C.c_void_p) else int(v))


class FrlwEvents(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n", C.c_int64), ("layout", C.c_int32),
                ("row_stride", C.c_int32), ("xmap", C.c_void_p), ("ymap", C.c_void_p),
                ("map_w", C.c_int32), ("map_h", C.c_int32), ("tuning", C.POINTER(FrlwTuning))]


# every symbol include/frlw_evd.h declares: name -> (restype, argtypes)
_P, _I, _I64, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_EV = C.POINTER(FrlwEvents)
SYMBOLS = {
    "frlw_version": (C.c_char_p, []),
    "frlw_encoder_workspace_bytes": (_SZ, [_I64, _I, _I]),
    "frlw_encoder_path_counts": (_I, [C.POINTER(C.c_uint64)]),
    "frlw_encoder_status": (_I, [_P, _P, C.POINTER(C.c_int)]),
    "frlw_workspace_init": (_I, [_P, _SZ, _P]),
    "frlw_encoder_deferred_status": (_I, [_P, _P, C.POINTER(C.c_int)]),
    "frlw_eci_encode": (_I, [_EV, _I, _I, _P, _P, _P, _SZ, _P]),
    "frlw_ev_encode": (_I, [_EV, _I, _I, _I, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "frlw_sae_encode": (_I, [_EV, _I, _I, C.POINTER(C.c_double), _I, _P, _P, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "frlw_taf_encode": (_I, [_EV, _I, _I, _I, _I64, _I64, _I, _P, _P, _P, _I, _P, _SZ, _P]),
    "frlw_selftest_mfma_f32_rate": (_I, [_I, _I, _P, _P, _P]),
    "frlw_selftest_lds_atomic_order": (_I, [_I, _I, _P, _P]),
    "frlw_fast_path_verdict": (_I, [_P, _SZ, _P, C.POINTER(C.c_int)]),
    "frlw_taf_batch_workspace_bytes": (_SZ, [_I64, _I, _I, _I, _I64]),
    "frlw_taf_encode_batch": (_I, [_EV, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I, _I, _I, _I, _I64, _I, _P, _P, _P, _I,
                                  _P, _SZ, _P]),
    "frlw_taf_stripe_partition": (_I, [_EV, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I, _I, _I, _I, _I, _I, _I64, _I, _P, _SZ, _P]),
    "frlw_taf_stripe_window_masks": (_P, [_P]),
    "frlw_taf_stripe_finish": (_I, [_EV, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I, _I, _I, _I, _I, _I, _I64, _I, _P, _P, _P, _I,
                                   _P, _SZ, _P]),
    "frlw_ev_batch_workspace_bytes": (_SZ, [_I64, _I, _I, _I, _I64]),
    "frlw_ev_encode_batch": (_I, [_EV, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _I, _I, _I, _I, _I64, _P, _P, _P, _SZ, _P]),
    "frlw_leaky_transform": (_I, [_P, _I64, _P, _P, _P]),
    "frlw_resize_nearest_f32": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "frlw_resize_nearest_u8": (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "frlw_quantize_u8": (_I, [_P, _I64, _I, _P, _P]),
    "frlw_det_create": (_P, []),
    "frlw_det_destroy": (None, [_P]),
    "frlw_det_num_ops": (_I, [_P]),
    "frlw_det_set_scratch": (_I, [_P, _I, _I64]),
    "frlw_det_set_lane": (_I, [_P, _I]),
    "frlw_det_set_precision": (_I, [_P, _I]),
    "frlw_conv_split_operand_bytes": (_SZ, [_I, _I]),
    "frlw_conv_split_operand": (_I, [_P, _I, _I, _P, _P]),
    "frlw_det_add_fork": (_I, [_P]),
    "frlw_det_add_join": (_I, [_P]),
    "frlw_det_add_focus": (_I, [_P, _I, _I, _I, _I, _I]),
    "frlw_det_add_upsample": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I]),
    "frlw_det_add_spp_pool": (_I, [_P, _I, _I, _I, _I, _I]),
    "frlw_focus_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "frlw_det_add_focus_stem": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I]),
    "frlw_det_add_pred": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I64]),
    "frlw_det_add_conv": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I64, _I, _I, _I, _I, _I, _I]),
    "frlw_det_add_decode_nms": (_I, [_P, _I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                     C.c_float, C.c_float, _I, _I, _I, _I]),
    "frlw_det_nms_workspace_floats": (C.c_longlong, [_I]),
    "frlw_det_run": (_I, [_P, _I, C.POINTER(C.c_void_p), _I, _I, _I, _P]),
    "frlw_det_bfm_weight_count": (_I, [_I]),
    "frlw_det_add_bfm_stem": (_I, [_P, _I, _I, _I, _I, _P, _I, _I]),
    "frlw_sample_transform_u8": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "frlw_eval_transform_dt": (_I, [_P, _P, _P, _I64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                   _P, _P, _P]),
    "frlw_conv2d_dgrad_parity": (_I, [_I, _I, _I, _I]),
    "frlw_conv_operand_floats": (_I64, [_I, _I, _I]),
    "frlw_conv_weight_layouts": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "frlw_conv_weight_layouts_batch": (_I, [_P, _I, C.c_int64, _P]),
    "frlw_conv2d_fwd": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _I64, _I, _P]),
    "frlw_conv2d_dgrad": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _I64, _I, _P]),
    "frlw_conv2d_wgrad_scratch_floats": (_I64, [_I, _I, _I, _I, _I, _I]),
    "frlw_conv2d_wgrad": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _I64, _I, _P]),
    "frlw_bn_scratch_doubles": (_I64, [_I64, _I]),
    "frlw_bn_stats": (_I, [_P, _I64, _I, C.c_float, _P, _P, _P, _P, _P]),
    "frlw_bn_silu_fwd": (_I, [_P, _I64, _I, _P, _P, _P, _P, _P, _P]),
    "frlw_bn_silu_bwd": (_I, [_P, _I64, _P, _I64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frlw_baseconv_weight_cache_floats": (_I64, [_I, _I, _I, _I]),
    "frlw_baseconv_train_scratch_bytes": (_I64, [_I, _I, _I, _I, _I, _I, _I]),
    "frlw_baseconv_train_fwd": (_I, [_P, _P, _P, _P, C.c_float, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P,
                                    C.c_float, _P, _P, _P, _I64, _P, _P, _I, _P]),
    "frlw_baseconv_train_bwd": (_I, [_P, _I64, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P,
                                    _P, _I64, _P, _P, _I, _P]),
    "frlw_spp_train_fwd": (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "frlw_spp_train_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "frlw_pred_bwd_scratch_floats": (_I64, [_I64, _I, _I]),
    "frlw_pred_fwd": (_I, [_P, _P, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "frlw_pred_bwd": (_I, [_P, _P, _P, _I64, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "frlw_simota_workspace_bytes": (_SZ, [_I, _I, _I]),
    "frlw_simota_assign": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, C.c_float, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "frlw_yolox_loss_workspace_bytes": (_SZ, [_I, _I, _I]),
    "frlw_yolox_loss_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I, C.c_float, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "frlw_yolox_loss_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


# symbols of the developer build only (-DFRLW_DEV_BUILD, libfrlw_evd_dev.so): bound when present
DEV_SYMBOLS = {
    "frlw_debug_force_lds_order": (_I, [_I]),
}


def library_path() -> str:
    # FRLW_LIB_PATH: developer knob to A/B a differently-built libfrlw_evd (tools/variants.sh)
    return os.environ.get("FRLW_LIB_PATH") or _build.LIB


def load():
    """Load libfrlw_evd.so (never builds: call __graft_entry__.build() first)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension is not built and there is no fallback path. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'`.")
        # torch ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import it first so that this
        # library binds to the HIP runtime that owns torch's device memory and streams; loading in the
        # other order puts two HIP runtimes in the process and ours sees "no ROCm-capable device".
        import torch  # noqa: F401
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in DEV_SYMBOLS.items():
            fn = getattr(lib, name, None)
            if fn is not None:
                fn.restype = res
                fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str = "frlw"):
    if rc == FRLW_OK:
        return
    msg = f"{what}: {_ERR_TEXT.get(rc, 'error')} ({rc})"
    if rc == FRLW_ERR_INDEX:
        raise IndexError(msg)
    if rc in (FRLW_ERR_ARG, FRLW_ERR_POLARITY, FRLW_ERR_UNSUPPORTED, FRLW_ERR_SPAN):
        raise ValueError(msg)
    raise RuntimeError(msg)
