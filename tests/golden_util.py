"""Helpers shared by the golden-vector tests (oracle on CPU, HIP path on GPU)."""
import hashlib

import numpy as np

GEN1 = ((240, 304), (256, 320))
MPX = ((720, 1280), (512, 640))
LAMDAS = [0.00001, 0.0000025, 0.000001]  # generate_surfaceofactiveevents.py:103


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bitexact(got, want, what=""):
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    assert got.dtype == want.dtype, f"{what}: dtype {got.dtype} != {want.dtype}"
    if got.tobytes() != want.tobytes():
        bad = np.flatnonzero(bits(got).reshape(-1) != bits(want).reshape(-1))
        raise AssertionError(f"{what}: {bad.size} of {got.size} elements differ, first at {bad[:5]}: "
                             f"{got.reshape(-1)[bad[:5]]} vs {want.reshape(-1)[bad[:5]]}")


def assert_big(got, g, prefix, what=""):
    """Check a big f32 buffer against its golden sha256 + sampled positions."""
    got = np.ascontiguousarray(got)
    assert tuple(g[prefix + "_shape"]) == got.shape, f"{what}: shape"
    flat = got.reshape(-1)
    assert_bitexact(flat[g[prefix + "_idx"]], g[prefix + "_val"], what + " (sample)")
    assert sha(flat) == str(g[prefix + "_sha"]), f"{what}: sha256 differs although the sample matches"


def assert_u8_budget(got, want, frac, what=""):
    """uint8 artefacts after a transcendental: mismatches only by 1 LSB and at most `frac`."""
    got = np.ascontiguousarray(got).astype(np.int16)
    want = np.ascontiguousarray(want).astype(np.int16)
    assert got.shape == want.shape, f"{what}: shape"
    d = np.abs(got - want)
    assert d.max(initial=0) <= 1, f"{what}: uint8 differs by {d.max()} LSB"
    n = int((d != 0).sum())
    assert n <= frac * got.size, f"{what}: {n} of {got.size} uint8 values differ (budget {frac:g})"
    return n


def downscale_maps(sensor_shape, shape):
    """The harness' `x * rw`, `y * rh` in f64 then .long() truncation (generate_taf.py:216-219)."""
    Hs, Ws = sensor_shape
    H, W = shape
    rw, rh = W / Ws, H / Hs
    xmap = (np.arange(Ws, dtype=np.float64) * rw).astype(np.int64).astype(np.uint16)
    ymap = (np.arange(Hs, dtype=np.float64) * rh).astype(np.int64).astype(np.uint16)
    return xmap, ymap
