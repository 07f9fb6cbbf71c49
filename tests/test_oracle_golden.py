"""The CPU oracle (oracle/frlw_oracle.c) against golden vectors produced by the reference's own
Python functions (tests/golden/make_golden.py).  Runs on CPU: `-m "not gpu"`.

Bars: bit-exact f32 everywhere except directly after expf / log1pf (torch-CPU = SLEEF, oracle =
libm), where the quantised uint8 artefact may differ by 1 LSB in at most 1e-5 of the elements
(level L2) and the f32 values agree to 2 ulp.
"""
import os

import numpy as np
import pytest

from frlw_evd_amd import synth
from oracle import oracle as orc
from golden_util import (GEN1, LAMDAS, MPX, assert_big, assert_bitexact, assert_u8_budget,
                         downscale_maps, sha)

U8_BUDGET = 1e-5


@pytest.fixture(scope="module")
def tiny(golden_dir):
    return np.load(os.path.join(golden_dir, "tiny.npz"))


def assert_ulp(got, want, ulps, what):
    a = np.ascontiguousarray(got).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(want).view(np.int32).astype(np.int64)
    assert np.abs(a - b).max() <= ulps, f"{what}: {np.abs(a - b).max()} ulp"


def test_tiny_eci_ev(tiny):
    H, W = tiny["shape"]
    ev = tiny["events"]
    assert_bitexact(orc.eventframe(ev, (H, W)), tiny["eci"], "eci")
    assert_bitexact(orc.event_volume(ev, (H, W), 5), tiny["ev"], "ev")
    assert_bitexact(orc.event_volume(ev, (H, W), 3), tiny["ev_bins3"], "ev bins=3")
    # hot pixel (>= 25 events) and the 20-event cell saturate at exactly 255
    assert tiny["eci"][1, 2, 3] == 255.0 and tiny["eci"][0, 0, 0] == 255.0


def test_tiny_sae(tiny):
    H, W = tiny["shape"]
    se, half, now = tiny["sae_events"], int(tiny["sae_half"]), tiny["sae_now"]
    o1, m1 = orc.sae(se[:half], (H, W), LAMDAS, None, now[0])
    assert_bitexact(m1, tiny["sae_mem1"], "sae memory 1")
    assert_ulp(o1, tiny["sae_out1"], 2, "sae out 1")
    o2, m2 = orc.sae(se[half:], (H, W), LAMDAS, m1, now[1])
    assert_bitexact(m2, tiny["sae_mem2"], "sae memory 2")
    assert_ulp(o2, tiny["sae_out2"], 2, "sae out 2")


def test_tiny_taf(tiny):
    H, W = tiny["shape"]
    ev, sp = tiny["events"], tiny["taf_splits"]
    st = np.full((H, W, 2, 8), -6000, np.float32)
    for i in range(4):
        v, st = orc.taf_window(ev[sp[i]:sp[i + 1]], (H, W), st, 8)
        assert_bitexact(v, tiny[f"taf_view{i}"], f"taf view {i}")
        assert_bitexact(st, tiny[f"taf_state{i}"], f"taf state {i}")
    # the empty window (i = 1) leaves the state untouched (generate_taf.py:40-41)
    assert_bitexact(tiny["taf_state1"], tiny["taf_state0"], "empty window")
    assert_ulp(orc.leaky_transform(v), tiny["taf_leaky"], 2, "leaky_transform")
    st = np.full((H, W, 2, 4), -6000, np.float32)
    for i in range(4):
        v, st = orc.taf_window(ev[sp[i]:sp[i + 1]], (H, W), st, 4)
    assert_bitexact(st, tiny["taf_k4_state"], "taf K=4 state")
    assert_bitexact(v, tiny["taf_k4_view"], "taf K=4 view")


def test_taf_growing_branch_is_a_padded_step(golden_dir):
    """generate_taf.py:50-53 (past_volume with volume_bins - 1 slots): the reference's result equals an ordinary K-slot
    step on [-5999, old...] -- the identity the product's ``generate_taf_cuda`` uses for that branch."""
    g = np.load(os.path.join(golden_dir, "tiny_taf_grow.npz"))
    tiny = np.load(os.path.join(golden_dir, "tiny.npz"))
    H, W = (int(v) for v in tiny["shape"])
    for K in (8, 4):
        past = g[f"k{K}_past"]
        padded = np.concatenate([np.full((H, W, 2, 1), -5999, np.float32), past], axis=3)
        v, st = orc.taf_window(tiny["events"][:150], (H, W), padded, K)
        assert_bitexact(st, g[f"k{K}_state"], f"K={K} grown state")
        assert_bitexact(v, g[f"k{K}_view"], f"K={K} grown view")
        assert np.any(st[..., 0] == -6000) and np.any(st[..., 0] != -6000)


def test_out_of_range_raises():
    ev = np.array([[0.0, 8.0, 0.5, 1.0]])  # flat index past the end -> IndexError in torch
    for fn in (lambda: orc.eventframe(ev, (8, 12)), lambda: orc.event_volume(ev, (8, 12), 5),
               lambda: orc.taf_window(ev, (8, 12), np.zeros((8, 12, 2, 8), np.float32), 8)):
        with pytest.raises(IndexError):
            fn()
    # SAE filters instead of raising (generate_surfaceofactiveevents.py:72)
    # x >= W alone only aliases into the next row (flat index x + W*y), like the reference
    alias = orc.eventframe(np.array([[13.0, 1.0, 0.5, 1.0]]), (8, 12))
    assert alias[1, 2, 1] > 0 and np.count_nonzero(alias) == 1
    out, mem = orc.sae(ev, (8, 12), LAMDAS, None, 100)
    assert np.all(mem == np.float32(100) - np.float32(5000000))


def test_empty_stream():
    ev = np.zeros((0, 4))
    assert np.all(orc.eventframe(ev, (8, 12)) == 0)
    assert np.all(orc.event_volume(ev, (8, 12), 5) == 0)
    st = np.full((8, 12, 2, 8), -6000, np.float32)
    v, st2 = orc.taf_window(ev, (8, 12), st, 8)
    assert_bitexact(st2, st, "empty stream state")


# ------------------------------------------------------------------------------------------
# GEN1-shaped goldens (SURVEY.md section 8d cfg 1, 2, 5), through the harness restatement
# ------------------------------------------------------------------------------------------
def test_gen1_eci(golden_dir):
    g = np.load(os.path.join(golden_dir, "gen1_eci.npz"))
    shape, tshape = GEN1
    ev = synth.synth_events(1001, 100_000, shape[1], shape[0], 50_000)
    nat = orc.eventframe(synth.to_xytp_f64(ev), shape)
    assert_bitexact(nat, g["eci_native"], "eci native")
    assert_bitexact(orc.quantize_u8(orc.resize_nearest(nat, tshape)), g["eci_u8"], "eci u8")
    # DAT path = same numbers
    assert_bitexact(orc.eci_stream_dat8(synth.to_dat8(ev), shape, shape), g["eci_native"], "eci dat8")
    ev = synth.synth_events(1001, 100_000, shape[1], shape[0], 50_000, hotspot=True)
    nat = orc.eci_stream_dat8(synth.to_dat8(ev), shape, shape)
    assert sha(nat) == str(g["eci_hot_native_sha"])
    assert_bitexact(orc.quantize_u8(orc.resize_nearest(nat, tshape)), g["eci_hot_u8"], "eci hot u8")


@pytest.mark.parametrize("tag,hot", [("", False), ("hot_", True)])
def test_gen1_ev(golden_dir, tag, hot):
    g = np.load(os.path.join(golden_dir, "gen1_ev.npz"))
    shape, tshape = GEN1
    ev = synth.synth_events(1002, 1_000_000, shape[1], shape[0], 250_000, hotspot=hot)
    nat = orc.ev_stream_dat8(synth.to_dat8(ev), shape, shape, 5, 250_000, 250_000)
    assert_big(nat, g, tag + "native", "ev native")
    assert_bitexact(orc.quantize_u8(orc.resize_nearest(nat, tshape), clip255=True), g[tag + "u8"], "ev u8")
    if not hot:  # the f64-tensor path, harness normalisation done with numpy like the reference
        keep = ev["t"] > 0
        t = (ev["t"][keep].astype(np.float64) - 0) / 250_000
        e = synth.to_xytp_f64({k: v[keep] for k, v in ev.items()}, t)
        assert_big(orc.event_volume(e, shape, 5), g, "native", "ev native f64")


def test_gen1_sae(golden_dir):
    g = np.load(os.path.join(golden_dir, "gen1_sae.npz"))
    shape, tshape = GEN1
    ev = synth.synth_events(1006, 1_000_000, shape[1], shape[0], 5_000_000, t_offset=30_000_000)
    dat = synth.to_dat8(ev)
    cut = int(g["cut"])
    now = g["now"]
    o1, m1 = orc.sae_stream_dat8(dat[:cut], shape, shape, LAMDAS, None, now[0], 5541263)
    assert_big(m1, g, "mem1", "sae mem1")
    n = assert_u8_budget(orc.quantize_u8(orc.resize_nearest(o1, tshape)), g["u8_1"], U8_BUDGET, "sae u8 1")
    o2, m2 = orc.sae_stream_dat8(dat[cut:], shape, shape, LAMDAS, m1, now[1], 5541263)
    assert_big(m2, g, "mem2", "sae mem2")
    n += assert_u8_budget(orc.quantize_u8(orc.resize_nearest(o2, tshape)), g["u8_2"], U8_BUDGET, "sae u8 2")
    flat = o1.reshape(-1)
    a = flat[g["native1_idx"]].view(np.int32).astype(np.int64)
    b = g["native1_val"].view(np.int32).astype(np.int64)
    assert np.abs(a - b).max() <= 2


def _taf_u8(view, K, tshape):
    v = orc.resize_nearest(view, tshape).reshape(K, 2, *tshape)
    lk = orc.leaky_transform(v)
    return orc.quantize_u8(np.ascontiguousarray(lk[::-1]))


@pytest.mark.parametrize("tag,hot", [("", False), ("hot_", True)])
def test_gen1_taf(golden_dir, tag, hot):
    g = np.load(os.path.join(golden_dir, "gen1_taf.npz"))
    shape, tshape = GEN1
    K = 8
    ev = synth.synth_events(1005, 1_000_000, shape[1], shape[0], 80_000, hotspot=hot)
    st0 = np.full((*shape, 2, K), -6000, np.float32)
    view, st = orc.taf_stream_dat8(synth.to_dat8(ev), shape, shape, K, 0, 10_000, 8, st0)
    assert_big(st, g, tag + "state", "taf state")
    assert_big(view, g, tag + "native", "taf view")
    assert_u8_budget(_taf_u8(view, K, tshape), g[tag + "u8"], U8_BUDGET, "taf u8")
    if not hot:  # second label: state carried over, 3 more windows (generate_taf.py:180-186)
        ev2 = synth.synth_events(2005, 300_000, shape[1], shape[0], 30_000, t_offset=80_000)
        view2, st2 = orc.taf_stream_dat8(synth.to_dat8(ev2), shape, shape, K, 80_000, 10_000, 3, st)
        assert_big(st2, g, "carry_state", "taf carry state")
        assert_u8_budget(_taf_u8(view2, K, tshape), g["carry_u8"], U8_BUDGET, "taf carry u8")


# ------------------------------------------------------------------------------------------
# 1 Mpx-shaped goldens (SURVEY.md section 8d cfg 3) -- sha256 + samples
# ------------------------------------------------------------------------------------------
def test_mpx_taf_native(golden_dir):
    g = np.load(os.path.join(golden_dir, "mpx_taf_native.npz"))
    shape = MPX[0]
    K = 8
    ev = synth.synth_events(1003, 10_000_000, shape[1], shape[0], 80_000)
    st0 = np.full((*shape, 2, K), -6000, np.float32)
    view, st = orc.taf_stream_dat8(synth.to_dat8(ev), shape, shape, K, 0, 10_000, 8, st0)
    assert_big(st, g, "state", "mpx taf state")
    u8 = orc.quantize_u8(np.ascontiguousarray(orc.leaky_transform(view.reshape(K, 2, *shape))[::-1]))
    assert_u8_budget(u8.reshape(-1)[g["u8_idx"]], g["u8_val"], 1e-4, "mpx taf u8 sample")


def test_mpx_downscale(golden_dir):
    g = np.load(os.path.join(golden_dir, "mpx_downscale.npz"))
    shape, tshape = MPX
    K = 8
    ev = synth.synth_events(1013, 2_000_000, shape[1], shape[0], 80_000, hotspot=True)
    dat = synth.to_dat8(ev)
    st0 = np.full((*tshape, 2, K), -6000, np.float32)
    view, st = orc.taf_stream_dat8(dat, shape, tshape, K, 0, 10_000, 8, st0)
    assert_big(st, g, "state", "downscale taf state")
    assert_big(orc.ev_stream_dat8(dat, shape, tshape, 5, 80_000, 80_000), g, "ev_native", "downscale ev")
    assert_big(orc.eci_stream_dat8(dat[-200_000:], shape, tshape), g, "eci_native", "downscale eci")
    # the integer maps the HIP path uses are the same truncation
    xmap, ymap = downscale_maps(shape, tshape)
    assert np.array_equal(xmap, np.arange(shape[1]) // 2)
    assert np.array_equal(ymap, (np.arange(shape[0]) * tshape[0]) // shape[0])
