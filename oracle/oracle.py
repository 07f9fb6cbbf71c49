"""ctypes front-end of ``oracle/frlw_oracle.c`` (CPU restatement, test infrastructure only).

Parity pinned against the reference's own Python run on CPU: see ``tests/golden/make_golden.py``
and ``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfrlw_oracle.so")
_lib = None


class OracleIndexError(IndexError):
    pass


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "frlw_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libfrlw_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _check(rc):
    if rc == -1:
        raise OracleIndexError("index out of range (torch would raise IndexError)")
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}")


def _ev(events):
    ev = np.ascontiguousarray(events, dtype=np.float64)
    assert ev.ndim == 2 and ev.shape[1] >= 4
    return ev


def eventframe(events, shape):
    H, W = (int(v) for v in shape)
    ev = _ev(events)
    out = np.empty((2, H, W), np.float32)
    _check(lib().orc_eventframe(_p(ev, C.c_double), C.c_int64(len(ev)), ev.shape[1], H, W,
                                _p(out, C.c_float)))
    return out


def event_volume(events, shape, volume_bins=5):
    H, W = (int(v) for v in shape)
    ev = _ev(events)
    out = np.empty((2 * volume_bins, H, W), np.float32)
    _check(lib().orc_event_volume(_p(ev, C.c_double), C.c_int64(len(ev)), ev.shape[1], H, W,
                                  int(volume_bins), _p(out, C.c_float)))
    return out


def sae(events, shape, lamdas, memory, now):
    H, W = (int(v) for v in shape)
    ev = _ev(events)
    lam = np.ascontiguousarray(lamdas, np.float64)
    out = np.empty((2 * len(lam), H, W), np.float32)
    mem_out = np.empty((2, H, W), np.float32)
    mem_in = None if memory is None else np.ascontiguousarray(memory, np.float32)
    _check(lib().orc_sae(_p(ev, C.c_double), C.c_int64(len(ev)), ev.shape[1], H, W,
                         _p(lam, C.c_double), len(lam), _p(mem_in, C.c_float), C.c_int64(int(now)),
                         _p(out, C.c_float), _p(mem_out, C.c_float)))
    return out, mem_out


def taf_window(events, shape, state, volume_bins):
    """One window.  ``state`` (H,W,2,K) f32 is NOT mutated; returns (view (2K,H,W), new state)."""
    H, W = (int(v) for v in shape)
    ev = _ev(events)
    st = np.array(state, dtype=np.float32, copy=True, order="C")
    assert st.shape == (H, W, 2, volume_bins)
    view = np.empty((2 * volume_bins, H, W), np.float32)
    _check(lib().orc_taf_window(_p(ev, C.c_double), C.c_int64(len(ev)), ev.shape[1], H, W,
                                volume_bins, _p(st, C.c_float), _p(view, C.c_float)))
    return view, st


def leaky_transform(ecd):
    a = np.ascontiguousarray(ecd, np.float32)
    out = np.empty_like(a)
    lib().orc_leaky_transform(_p(a, C.c_float), C.c_int64(a.size), _p(out, C.c_float))
    return out


def resize_nearest(vol, target_shape):
    a = np.ascontiguousarray(vol, np.float32)
    Cn, H, W = a.shape
    Ho, Wo = (int(v) for v in target_shape)
    out = np.empty((Cn, Ho, Wo), np.float32)
    lib().orc_resize_nearest(_p(a, C.c_float), Cn, H, W, Ho, Wo, _p(out, C.c_float))
    return out


def quantize_u8(vol, clip255=False):
    a = np.ascontiguousarray(vol, np.float32)
    out = np.empty(a.shape, np.uint8)
    lib().orc_quantize_u8(_p(a, C.c_float), C.c_int64(a.size), int(clip255), _p(out, C.c_uint8))
    return out


def _rec(dat):
    r = np.ascontiguousarray(dat)
    assert r.dtype.itemsize == 8
    return r


def taf_stream_dat8(dat, sensor_shape, shape, volume_bins, t_start, window_us, n_windows, state):
    Hs, Ws = (int(v) for v in sensor_shape)
    H, W = (int(v) for v in shape)
    r = _rec(dat)
    st = np.array(state, dtype=np.float32, copy=True, order="C")
    view = np.empty((2 * volume_bins, H, W), np.float32)
    _check(lib().orc_taf_stream_dat8(r.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(r)),
                                     Hs, Ws, H, W, volume_bins, C.c_int64(int(t_start)),
                                     C.c_int64(int(window_us)), int(n_windows),
                                     _p(st, C.c_float), _p(view, C.c_float)))
    return view, st


def ev_stream_dat8(dat, sensor_shape, shape, volume_bins, t_end, window_us):
    Hs, Ws = (int(v) for v in sensor_shape)
    H, W = (int(v) for v in shape)
    r = _rec(dat)
    out = np.empty((2 * volume_bins, H, W), np.float32)
    _check(lib().orc_ev_stream_dat8(r.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(r)),
                                    Hs, Ws, H, W, volume_bins, C.c_int64(int(t_end)),
                                    C.c_int64(int(window_us)), _p(out, C.c_float)))
    return out


def eci_stream_dat8(dat, sensor_shape, shape):
    Hs, Ws = (int(v) for v in sensor_shape)
    H, W = (int(v) for v in shape)
    r = _rec(dat)
    out = np.empty((2, H, W), np.float32)
    _check(lib().orc_eci_stream_dat8(r.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(r)),
                                     Hs, Ws, H, W, _p(out, C.c_float)))
    return out


def sae_stream_dat8(dat, sensor_shape, shape, lamdas, memory, now, window_us):
    Hs, Ws = (int(v) for v in sensor_shape)
    H, W = (int(v) for v in shape)
    r = _rec(dat)
    lam = np.ascontiguousarray(lamdas, np.float64)
    out = np.empty((2 * len(lam), H, W), np.float32)
    mem_out = np.empty((2, H, W), np.float32)
    mem_in = None if memory is None else np.ascontiguousarray(memory, np.float32)
    _check(lib().orc_sae_stream_dat8(r.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(len(r)),
                                     Hs, Ws, H, W, _p(lam, C.c_double), len(lam),
                                     _p(mem_in, C.c_float), C.c_int64(int(now)),
                                     C.c_int64(int(window_us)), _p(out, C.c_float),
                                     _p(mem_out, C.c_float)))
    return out, mem_out


def sample_transform(vol_u8, hr, wr, y0, x0, flip):
    """Image half of the training loader's sample transform (data/dataset.py:217-231), numpy restatement:
    nearest resize (torch: src = min(floor(dst * float32(in / out)), in - 1)) to (hr, wr), / 255, crop at (y0, x0)
    to the input size, horizontal flip.  ``vol_u8`` (C, H, W) uint8 -> (C, H, W, 1, 1) float32."""
    Cc, H, W = vol_u8.shape
    sh = np.float32(H) / np.float32(hr)
    sw = np.float32(W) / np.float32(wr)
    ys = np.minimum(np.floor((np.arange(H) + y0).astype(np.float32) * sh).astype(np.int64), H - 1)
    xs = np.minimum(np.floor((np.arange(W) + x0).astype(np.float32) * sw).astype(np.int64), W - 1)
    out = vol_u8[:, ys][:, :, xs].astype(np.float32) / np.float32(255)
    if flip:
        out = out[:, :, ::-1]
    return np.ascontiguousarray(out)[:, :, :, None, None]
