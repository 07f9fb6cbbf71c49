"""The fast / batched TAF path (csrc/taf_fast.hip, ``frlw_taf_encode_batch``) against the CPU oracle (itself pinned to
the reference's goldens, tests/test_oracle_golden.py) and against the general path, bit for bit.

Covers what the reference's per-file harness loop implies (generate_taf.py:143-235): independent sequences with their
own "window without events leaves the state untouched" rule (:40-41), own start time, state carried from call to call.
"""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from frlw_evd_amd import synth  # noqa: E402
from golden_util import assert_bitexact, assert_u8_budget  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def er():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import event_representation
    return event_representation


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def to_dev(rec):
    return torch.from_numpy(np.ascontiguousarray(rec).view(np.uint8).reshape(-1, 8).copy()).cuda()


def host(t):
    return t.cpu().numpy()


def oracle_taf(orc, rec, shape, K, t_start, win, n_win, state0, flip=True):
    H, W = shape
    view, st = orc.taf_stream_dat8(rec, shape, shape, K, t_start, win, n_win, state0)
    u8 = orc.quantize_u8(orc.leaky_transform(view.reshape(K, 2, H, W)))
    return view, st, (np.ascontiguousarray(u8[::-1]) if flip else u8)


def test_lds_atomic_lane_order():
    """The hardware property the stable ranks rest on (see csrc/taf_fast.hip): zero mismatches over many conflicts."""
    from frlw_evd_amd import _lib
    lib = _lib.load()
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    for n_addr in (1, 3, 16, 40, 450, 512):
        _lib.check(lib.frlw_selftest_lds_atomic_order(n_addr, 50, C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        bad, conflicts, fbad = (int(v) for v in out.cpu())
        assert conflicts > 10_000, (n_addr, conflicts)
        assert bad == 0, f"{bad} rank mismatches with {n_addr} addresses"
        assert fbad == 0, f"{fbad} ds_add_f32 sums differ from the sequential f32 sum with {n_addr} addresses"


@pytest.mark.parametrize("hotspot", [False, True])
def test_gen1_vs_oracle(er, orc, hotspot):
    H, W, K = 240, 304, 8
    ev = synth.synth_events(5150, 1_000_000, W, H, 80_000, hotspot=hotspot)
    rec = synth.to_dat8(ev)
    st = torch.full((H, W, 2, K), -6000.0, device="cuda")
    u8, view = er.encode_taf_dat(to_dev(rec), (H, W), st, 0, 10_000, 8, K, want_view=True, fast=True)
    oview, ost, ou8 = oracle_taf(orc, rec, (H, W), K, 0, 10_000, 8, np.full((H, W, 2, K), -6000, np.float32))
    assert_bitexact(host(st), ost, "state")
    assert_bitexact(host(view), oview, "view")
    assert_u8_budget(host(u8), ou8, 1e-5, "uint8")


CASES = [
    # H, W, n, K, n_windows, window_us, hotspot
    (17, 33, 5_000, 8, 8, 10_000, False),
    (17, 33, 60_000, 4, 3, 977, True),
    (64, 64, 200_000, 5, 16, 5_000, False),
    (240, 304, 300_000, 8, 1, 10_000, False),
    (100, 1000, 400_000, 8, 8, 10_000, True),
    (720, 1280, 2_000_000, 8, 8, 10_000, False),
    (720, 1280, 1_500_000, 8, 8, 10_000, True),
    (31, 2047, 100_000, 2, 64, 250, False),
]


@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}-n{c[2]}-K{c[3]}-w{c[4]}" for c in CASES])
def test_fast_equals_general(er, case):
    H, W, n, K, n_win, win, hotspot = case
    ev = synth.synth_events(H * 7 + n_win, n, W, H, n_win * win, hotspot=hotspot, t_offset=123_456)
    ev["t"][-3:] = 123_456 + n_win * win  # exactly the end of the last window (generate_taf.py:197-203)
    dat = to_dev(synth.to_dat8(ev))
    init = torch.from_numpy(np.random.default_rng(3).uniform(-50, 0, (H, W, 2, K)).astype(np.float32)).cuda()
    sa, sb = init.clone(), init.clone()
    ua, va = er.encode_taf_dat(dat, (H, W), sa, 123_456, win, n_win, K, want_view=True, fast=True)
    ub, vb = er.encode_taf_dat(dat, (H, W), sb, 123_456, win, n_win, K, want_view=True, fast=False)
    assert not torch.equal(sa, init)
    assert torch.equal(sa, sb), "state"
    assert torch.equal(va, vb), "view"
    assert torch.equal(ua, ub), "uint8"


def test_batch_sparse_samples_vs_oracle(er, orc):
    """Sample 1 has no event in window 3, sample 2 is entirely empty, sample 3 starts late (windows 0-1 empty): each
    follows its OWN all(forward) rule (generate_taf.py:40-41), exactly like four separate files."""
    H, W, K, win, n_win = 240, 304, 8, 10_000, 8
    recs, starts = [], [0, 1_000_000, 5_000, 70_000]
    for j, t0 in enumerate(starts):
        ev = synth.synth_events(900 + j, 150_000, W, H, n_win * win, t_offset=t0)
        keep = np.ones(len(ev["t"]), bool)
        if j == 1:
            keep = (ev["t"] - t0) // win != 3
        if j == 2:
            keep[:] = False
        if j == 3:
            keep = (ev["t"] - t0) >= 2 * win + 17
        recs.append(synth.to_dat8({k: v[keep] for k, v in ev.items()}))
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = to_dev(np.concatenate(recs))
    state0 = np.random.default_rng(5).uniform(-30, 0, (4, H, W, 2, K)).astype(np.float32)
    st = torch.from_numpy(state0).cuda()
    u8, view = er.encode_taf_batch(dat, offs, (H, W), st, starts, win, n_win, K, want_view=True)
    for j in range(4):
        oview, ost, ou8 = oracle_taf(orc, recs[j], (H, W), K, starts[j], win, n_win, state0[j])
        assert_bitexact(host(st[j]), ost, f"state of sample {j}")
        assert_bitexact(host(view[j]), oview, f"view of sample {j}")
        assert_u8_budget(host(u8[j]), ou8, 1e-4, f"uint8 of sample {j}")
        # and the per-sample general path agrees bit for bit
        sj = torch.from_numpy(state0[j]).cuda()
        uj, _ = er.encode_taf_dat(to_dev(recs[j]) if len(recs[j]) else torch.empty((0, 8), dtype=torch.uint8, device="cuda"),
                                  (H, W), sj, starts[j], win, n_win, K, fast=False)
        assert torch.equal(sj, st[j]) and torch.equal(uj, u8[j]), f"general path, sample {j}"
    assert_bitexact(host(st[2]), state0[2], "an empty sequence leaves its state untouched")


def test_unsorted_stream(er, orc):
    H, W, K, win, n_win = 64, 96, 8, 10_000, 8
    ev = synth.synth_events(77, 120_000, W, H, n_win * win, hotspot=True)
    perm = np.random.default_rng(1).permutation(len(ev["t"]))
    ev = {k: v[perm] for k, v in ev.items()}
    rec = synth.to_dat8(ev)
    st = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    u8, view = er.encode_taf_batch(to_dev(rec), [0, len(rec)], (H, W), st, 0, win, n_win, K, want_view=True)
    oview, ost, ou8 = oracle_taf(orc, rec, (H, W), K, 0, win, n_win, np.full((H, W, 2, K), -6000, np.float32))
    assert_bitexact(host(st[0]), ost, "state")
    assert_bitexact(host(view[0]), oview, "view")


def test_span_violation_writes_nothing_and_falls_back(er):
    H, W, K, win, n_win = 48, 80, 8, 10_000, 4
    ev = synth.synth_events(31, 50_000, W, H, n_win * win, t_offset=100_000)
    ev["t"][10] = 99_000       # before t_start
    ev["t"][-1] = 100_000 + n_win * win + 5  # after the last window
    dat = to_dev(synth.to_dat8(ev))
    init = torch.full((H, W, 2, K), -7.0, device="cuda")
    st = init.clone()
    with pytest.raises(ValueError):
        er.encode_taf_batch(dat, [0, len(ev["t"])], (H, W), st.view(1, H, W, 2, K), 100_000, win, n_win, K)
    assert torch.equal(st, init), "nothing may be written after a span violation"
    sa, sb = init.clone(), init.clone()
    ua, _ = er.encode_taf_dat(dat, (H, W), sa, 100_000, win, n_win, K, fast=True)   # falls back by itself
    ub, _ = er.encode_taf_dat(dat, (H, W), sb, 100_000, win, n_win, K, fast=False)
    assert torch.equal(sa, sb) and torch.equal(ua, ub) and not torch.equal(sa, init)
    with pytest.raises(IndexError):  # out of frame: the general path reports it like torch's index_add_
        oob = dict(ev)
        oob["y"] = ev["y"].copy()
        oob["y"][0] = H + 3
        oob["t"] = np.clip(ev["t"], 100_000, 100_000 + n_win * win)
        er.encode_taf_dat(to_dev(synth.to_dat8(oob)), (H, W), init.clone(), 100_000, win, n_win, K, fast=True)


def test_state_carry_and_downscale_maps(er, orc):
    """Two consecutive labels of one sequence (state carried, generate_taf.py:175-186) on the down-scaled 1 Mpx geometry
    (coordinate maps = x * rw, y * rh + truncation, :216-219)."""
    Hs, Ws, H, W, K, win, n_win = 720, 1280, 512, 640, 8, 10_000, 4
    xmap, ymap = er.coordinate_maps((Hs, Ws), (H, W), "cuda")
    st = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    ost = np.full((H, W, 2, K), -6000, np.float32)
    for part in range(2):
        ev = synth.synth_events(400 + part, 600_000, Ws, Hs, n_win * win, t_offset=part * n_win * win)
        rec = synth.to_dat8(ev)
        er.encode_taf_batch(to_dev(rec), [0, len(rec)], (H, W), st, part * n_win * win, win, n_win, K, xmap=xmap, ymap=ymap)
        _, ost = orc.taf_stream_dat8(rec, (Hs, Ws), (H, W), K, part * n_win * win, win, n_win, ost)
        assert_bitexact(host(st[0]), ost, f"state after part {part}")


def test_gen1_batch64_equals_per_sample(er):
    """BASELINE.json configs[4] encode shape: 64 GEN1 streams of 8 x 125 000 events in one call == 64 single calls."""
    H, W, K, win, n_win, B = 240, 304, 8, 10_000, 8, 64
    recs = [synth.to_dat8(synth.synth_events(2000 + j, 1_000_000 if j < 2 else 60_000, W, H, n_win * win)) for j in range(B)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    st = torch.full((B, H, W, 2, K), -6000.0, device="cuda")
    u8, _ = er.encode_taf_batch(to_dev(np.concatenate(recs)), offs, (H, W), st, 0, win, n_win, K)
    for j in (0, 1, 2, 31, 63):
        sj = torch.full((H, W, 2, K), -6000.0, device="cuda")
        uj, _ = er.encode_taf_dat(to_dev(recs[j]), (H, W), sj, 0, win, n_win, K, fast=False)
        assert torch.equal(sj, st[j]) and torch.equal(uj, u8[j]), f"sample {j}"


def test_unchecked_calls_leave_a_deferred_status(er, orc):
    """An unchecked fast-path call whose events leave the span writes nothing -- and must not go unnoticed: the status
    accumulates in the workspace and ``raise_deferred`` surfaces it with ONE sync, then is clean again.  ``fast="auto"``
    keeps unchecked calls on the general path, which places such events like the reference does (z = 0)."""
    H, W, K = 64, 96, 8
    ev = synth.synth_events(77, 40_000, W, H, 80_000)
    rec = synth.to_dat8(ev)
    bad = rec.copy()
    bad["t"][-5:] += 1_000_000  # five events far behind the last window
    st0 = np.full((H, W, 2, K), -6000, np.float32)
    er.raise_deferred()  # whatever earlier tests left behind
    st = torch.from_numpy(st0.copy()).cuda()
    er.encode_taf_batch(to_dev(bad), [0, len(bad)], (H, W), st.view(1, H, W, 2, K), 0, 10_000, 8, K, check=False)
    assert_bitexact(host(st), st0, "nothing was written")
    er.encode_taf_batch(to_dev(rec), [0, len(rec)], (H, W), st.view(1, H, W, 2, K), 0, 10_000, 8, K, check=False)  # a clean call after it
    with pytest.raises(ValueError):
        er.raise_deferred()
    er.raise_deferred()  # cleared by the read
    # unchecked + fast="auto": the general path, same bits as the oracle (which places out-of-span events in window 0)
    st = torch.from_numpy(st0.copy()).cuda()
    big = synth.to_dat8(synth.synth_events(78, er.FAST_MIN_EVENTS + 10, W, H, 80_000))
    big["t"][-3:] += 1_000_000
    er.encode_taf_dat(to_dev(big), (H, W), st, 0, 10_000, 8, K, check=False)
    er.raise_deferred()
    _, ost, _ = oracle_taf(orc, big, (H, W), K, 0, 10_000, 8, st0)
    assert_bitexact(host(st), ost, "general path on the out-of-span stream")
    # an out-of-frame coordinate through the unchecked general path: IndexError at the deferred check
    oob = rec[:100].copy()
    oob["_"][7] = (oob["_"][7] & np.uint32(~(16383 << 14) & 0xFFFFFFFF)) | np.uint32((H + 3) << 14)  # y = H + 3: flat index beyond the frame
    er.encode_taf_dat(to_dev(oob), (H, W), torch.from_numpy(st0.copy()).cuda(), 0, 10_000, 8, K, check=False, fast=False)
    with pytest.raises(IndexError):
        er.raise_deferred()


def test_an_earlier_unchecked_error_is_not_mistaken_for_this_calls(er, orc):
    """the status word is sticky.  An unchecked call leaves an error; a later CLEAN checked
    ``encode_taf_dat(fast=True)`` must raise that earlier error BEFORE it touches the state -- not step the state with the fast
    path, read the old error, take it for its own and step the state a second time through the general path."""
    H, W, K = 64, 96, 8
    rec = synth.to_dat8(synth.synth_events(81, 40_000, W, H, 80_000))
    bad = rec.copy()
    bad["t"][-5:] += 1_000_000
    st0 = np.full((H, W, 2, K), -6000, np.float32)
    er.raise_deferred()
    junk = torch.from_numpy(st0.copy()).cuda()
    er.encode_taf_batch(to_dev(bad), [0, len(bad)], (H, W), junk.view(1, H, W, 2, K), 0, 10_000, 8, K, check=False)
    st = torch.from_numpy(st0.copy()).cuda()
    with pytest.raises(ValueError):
        er.encode_taf_dat(to_dev(rec), (H, W), st, 0, 10_000, 8, K, fast=True)
    assert_bitexact(host(st), st0, "the earlier call's error surfaced before this call ran")
    er.encode_taf_dat(to_dev(rec), (H, W), st, 0, 10_000, 8, K, fast=True)  # the word is clean now: ONE step
    _, ost, _ = oracle_taf(orc, rec, (H, W), K, 0, 10_000, 8, st0)
    assert_bitexact(host(st), ost, "stepped exactly once")


_FORCED_FAILURE_CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from frlw_evd_amd import _lib, synth, event_representation as er
from oracle import oracle as orc
lib = _lib.load()
assert hasattr(lib, "frlw_debug_force_lds_order"), "the developer library must carry the hook"
H, W, K = 64, 96, 8
rec = synth.to_dat8(synth.synth_events(79, 30_000, W, H, 80_000))
st0 = np.full((H, W, 2, K), -6000, np.float32)
_, ost = orc.taf_stream_dat8(rec, (H, W), (H, W), K, 0, 10_000, 8, st0)
dev = lambda r: torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).reshape(-1, 8).copy()).cuda()
try:
    _lib.check(lib.frlw_debug_force_lds_order(0))  # "the property does not hold on this device"
    st = torch.from_numpy(st0.copy()).cuda()
    try:
        er.encode_taf_batch(dev(rec), [0, len(rec)], (H, W), st.view(1, H, W, 2, K), 0, 10_000, 8, K)
        raise SystemExit("the fast path did not refuse")
    except NotImplementedError:
        pass
    assert st.cpu().numpy().tobytes() == st0.tobytes(), "refused before anything ran"
    er.encode_taf_dat(dev(rec), (H, W), st, 0, 10_000, 8, K, fast=True)  # falls back to the general path
    assert st.cpu().numpy().tobytes() == ost.tobytes(), "general path after the refusal"
finally:
    _lib.check(lib.frlw_debug_force_lds_order(-1))  # forget: the next call runs the real self-test again
st = torch.from_numpy(st0.copy()).cuda()
er.encode_taf_batch(dev(rec), [0, len(rec)], (H, W), st.view(1, H, W, 2, K), 0, 10_000, 8, K)
assert st.cpu().numpy().tobytes() == ost.tobytes(), "fast path after the real self-test passed"
print("forced-failure ok")
"""


def test_failed_lane_order_selftest_disables_the_fast_path():
    """The library checks the LDS lane-order property on its first fast-path call per device and caches the verdict; if it
    does not hold the fast path refuses and ``encode_taf_dat`` runs the general path -- same bits.  The hook that forces the
    verdict exists only in the developer build (libfrlw_evd_dev.so, -DFRLW_DEV_BUILD): a fresh child process loads that."""
    import os
    import subprocess
    import sys
    from frlw_evd_amd import _build
    dev_lib = _build.DEV_LIB
    if not os.path.exists(dev_lib):
        dev_lib = _build.build_dev()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _FORCED_FAILURE_CHILD, root], env=dict(os.environ, FRLW_LIB_PATH=dev_lib),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "forced-failure ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


# ---- the tile walk (kf_taf_tile, frlw_tuning_t::taf_tile_walk = 1): same bits as the default split + sub-tile kernels ----
def _batch(seed, B, n, H, W, span, hotspot=False):
    recs = [synth.to_dat8(synth.synth_events(seed + j, n if j != 2 else n // 40, W, H, span, hotspot=hotspot)) for j in range(B)]
    return recs, np.concatenate([[0], np.cumsum([len(r) for r in recs])])


@pytest.mark.parametrize("K,n_win,win", [(8, 8, 10_000), (4, 3, 977), (5, 16, 5_000)])
def test_tile_walk_equals_default_path(er, monkeypatch, K, n_win, win):
    """A batch with >= 256 (sequence, tile) pairs, one sparse sequence, one sequence with an empty window and one whose
    stream is NOT time-sorted (the tile walk needs window-sorted lists: kf_scatter flags it and it takes the default
    kernels inside the same call), plus a hot spot that pushes tiles over the segment limit."""
    from frlw_evd_amd import _lib
    H, W, B = 240, 304, 8  # 40 tiles per sequence -> 320 pairs
    recs, _ = _batch(4400 + K, B, 260_000, H, W, n_win * win, hotspot=True)
    r4 = recs[4]
    recs[4] = r4[(r4["t"] // win) != 1]                       # no event in window 1 of sequence 4
    recs[5] = recs[5][np.random.default_rng(9).permutation(len(recs[5]))]  # unsorted sequence
    recs[6] = np.concatenate([recs[6]] * 3)                   # time runs backwards twice, 3x the events
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = to_dev(np.concatenate(recs))
    init = torch.from_numpy(np.random.default_rng(3).uniform(-50, 0, (B, H, W, 2, K)).astype(np.float32)).cuda()
    sa, sb = init.clone(), init.clone()
    ua, va = er.encode_taf_batch(dat, offs, (H, W), sa, 0, win, n_win, K, want_view=True)
    monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(taf_tile_walk=1))
    ub, vb = er.encode_taf_batch(dat, offs, (H, W), sb, 0, win, n_win, K, want_view=True)
    assert not torch.equal(sa, init)
    assert torch.equal(sa, sb), "state"
    assert torch.equal(va, vb), "view"
    assert torch.equal(ua, ub), "uint8"


def test_tile_walk_mpx_golden(er, monkeypatch, golden_dir):
    """The reference's own 10 M-event 1280x720 state (sha256 golden) through the tile walk."""
    import hashlib
    import os
    from frlw_evd_amd import _lib
    g = np.load(os.path.join(golden_dir, "mpx_taf_native.npz"))
    H, W, K = 720, 1280, 8
    rec = synth.to_dat8(synth.synth_events(1003, 10_000_000, W, H, 80_000))
    st = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(taf_tile_walk=1))
    er.encode_taf_batch(to_dev(rec), [0, len(rec)], (H, W), st, 0, 10_000, 8, K)
    assert hashlib.sha256(host(st[0]).tobytes()).hexdigest() == str(g["state_sha"])


# ---- row-stripe sharding of one frame: stripes of several "ranks" == the whole-frame encode --------------
def _stripe_case(seed, H, W, n, n_win, win, hotspot):
    ev = synth.synth_events(seed, n, W, H, n_win * win, hotspot=hotspot)
    # window 2: events ONLY in the upper stripe rows; window 5: no event anywhere (the per-frame rule, generate_taf.py:40-41)
    w_idx = np.minimum(ev["t"] // win, n_win - 1)
    keep = ~((w_idx == 2) & (ev["y"] >= H // 3)) & (w_idx != 5)
    ev = {k: v[keep] for k, v in ev.items()}
    ev["x"][:7] = W + 3   # x >= W aliases into the next row of the WHOLE frame (flat index, generate_taf.py:23): may change stripe
    ev["y"][:7] = np.minimum(ev["y"][:7], H - 2)
    return synth.to_dat8(ev)


@pytest.mark.parametrize("cuts", [(0, 100, 240), (0, 33, 170, 240)])
def test_row_stripes_equal_whole_frame(er, cuts):
    """Two / three stripes encoded one after the other in this process -- each from the whole stream, each with its own
    workspace, the window masks OR-ed between the two halves of the encodes -- equal one encode of the whole frame."""
    H, W, K, win, n_win = 240, 304, 8, 10_000, 8
    recs = [_stripe_case(610, H, W, 700_000, n_win, win, False), _stripe_case(611, H, W, 90_000, n_win, win, True)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = to_dev(np.concatenate(recs))
    B = len(recs)
    init = torch.from_numpy(np.random.default_rng(8).uniform(-40, 0, (B, H, W, 2, K)).astype(np.float32)).cuda()
    full = init.clone()
    u_full, v_full = er.encode_taf_batch(dat, offs, (H, W), full, 0, win, n_win, K, want_view=True)
    stripes = list(zip(cuts[:-1], cuts[1:]))
    # pass 1: every stripe's own masks (what each rank has before the exchange); pass 2: the real encodes with the OR of all
    own = []
    for (lo, hi) in stripes:
        st = init[:, lo:hi].contiguous()
        er.encode_taf_stripe(dat, offs, (H, W), (lo, hi), st, 0, win, n_win, K, exchange=lambda m: own.append(m.clone()))
    assert len({tuple(m.tolist()) for m in own}) > 1, "the stripes should differ in their window masks (window 2)"
    total = own[0].clone()
    for m in own[1:]:
        total |= m
    assert all(((int(v) >> 5) & 1) == 0 for v in total.tolist()), "window 5 is empty in the whole frame"
    state = init.clone()
    for (lo, hi) in stripes:
        st = init[:, lo:hi].contiguous()
        u8, view = er.encode_taf_stripe(dat, offs, (H, W), (lo, hi), st, 0, win, n_win, K, want_view=True,
                                        exchange=lambda m: m.copy_(total))
        state[:, lo:hi] = st
        assert torch.equal(u8, u_full[:, :, :, lo:hi]), f"uint8 of stripe {lo}:{hi}"
        assert torch.equal(view, v_full[:, :, lo:hi]), f"view of stripe {lo}:{hi}"
    assert torch.equal(state, full), "stripes put together == whole frame"
    # without the exchange the upper stripe would age window 2 alone: the rule really is global
    st = init[:, stripes[-1][0]:stripes[-1][1]].contiguous()
    er.encode_taf_stripe(dat, offs, (H, W), stripes[-1], st, 0, win, n_win, K)
    assert not torch.equal(st, full[:, stripes[-1][0]:stripes[-1][1]])


def test_negative_start_time_takes_the_general_decode():
    """A sequence whose t_start is negative (or beyond 32 bits) is outside the SIMPLE decode of kf_hist / kf_scatter (32-bit time
    arithmetic, csrc/taf_fast.hip): the call falls back to the general instantiation -- same bits as the general path."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import event_representation as er, synth
    H, W, K, win, n_win = 120, 160, 8, 10_000, 8
    recs = [synth.to_dat8(synth.synth_events(900 + j, 150_000, W, H, n_win * win - 5_000)) for j in range(3)]
    starts = [-5_000, 0, -1]  # events at t in [0, 75 000): inside [t_start, t_start + 80 000] for every sequence
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])

    def dev(r):
        return torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).reshape(-1, 8).copy()).cuda()
    st = torch.full((3, H, W, 2, K), -6000.0, device="cuda")
    u8, view = er.encode_taf_batch(dev(np.concatenate(recs)), offs, (H, W), st, starts, win, n_win, K, want_view=True)
    for j in range(3):
        sj = torch.full((H, W, 2, K), -6000.0, device="cuda")
        uj, vj = er.encode_taf_dat(dev(recs[j]), (H, W), sj, starts[j], win, n_win, K, want_view=True, fast=False)
        assert torch.equal(sj, st[j]) and torch.equal(vj, view[j]) and torch.equal(uj, u8[j]), j


@pytest.mark.parametrize("n_seq", [1, 40])
def test_direct_bins_equal_tile_bins(n_seq, monkeypatch):
    """The direct partition mode (bins = 256-cell sub-tiles, no split pass; the default for calls with few (sequence, tile)
    pairs) against tile bins + split pass, both forced through frlw_tuning_t: same bits.  40 sequences x 576 bins = 23 040
    (sequence, bin) pairs: the tile scan runs in three rounds."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import _lib, event_representation as er, synth
    H, W, K, win, n_win = 240, 304, 8, 10_000, 8
    recs = []
    for j in range(n_seq):
        n = 0 if j == 5 else (400_000 if n_seq == 1 else 30_000 + 1_000 * j)
        recs.append(synth.to_dat8(synth.synth_events(1200 + j, n, W, H, n_win * win, hotspot=(j % 7 == 3))))
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = torch.from_numpy(np.ascontiguousarray(np.concatenate(recs)).view(np.uint8).reshape(-1, 8).copy()).cuda()
    outs = []
    # (direct, chunk_major): sub-tile / tile bins x the chunk-major partition (no histogram pass; the direct-mode walk gathers
    # its own list) / histogram + scans + bin-major scatter -- four ways to the same bits
    for direct, cmaj in ((1, 1), (1, 0), (0, 1), (0, 0)):
        monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(direct_bins=direct, chunk_major=cmaj))
        st = torch.full((n_seq, H, W, 2, K), -6000.0, device="cuda")
        u8, view = er.encode_taf_batch(dat, offs, (H, W), st, 0, win, n_win, K, want_view=True)
        outs.append((st, u8, view))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("shape,n_seq,n,hot", [((720, 1280), 1, 3_000_000, True), ((240, 304), 3, 900_000, True),
                                                ((97, 131), 5, 70_000, False), ((240, 304), 2, 0, False)])
def test_chunk_major_equals_histogram_partition(shape, n_seq, n, hot, monkeypatch):
    """frlw_tuning_t::chunk_major 1 against 0 at the library's own choice of bins: a 1280x720 stream with a hot spot (tile bins:
    whole-tile split through the directory columns AND the segment kernels for the skewed tiles), GEN1 sequences above and
    below the walk's LDS list (a hot sub-tile's list goes through rec2[]), tiny frames, empty sequences; state, view and uint8
    volume bit for bit, with a state carried over from a first call."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import _lib, event_representation as er, synth
    H, W = shape
    K, win, n_win = 8, 10_000, 8
    recs = [synth.to_dat8(synth.synth_events(1400 + j, n // (1 + 3 * (j == 1)), W, H, n_win * win, hotspot=hot)) for j in range(n_seq)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = torch.from_numpy(np.ascontiguousarray(np.concatenate(recs)).view(np.uint8).reshape(-1, 8).copy()).cuda()
    outs = []
    for cmaj in (1, 0):
        monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(chunk_major=cmaj))
        st = torch.full((n_seq, H, W, 2, K), -6000.0, device="cuda")
        for _ in range(2):  # second call: the FIFO state of the first carried over
            u8, view = er.encode_taf_batch(dat, offs, (H, W), st, 0, win, n_win, K, want_view=True)
        outs.append((st, u8, view))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape,n_seq,n,hot,bpw", [((720, 1280), 1, 700_000, True, 20), ((720, 1280), 2, 333_337, False, 12),
                                                    ((240, 304), 3, 250_001, True, 20), ((97, 131), 2, 9_000, False, 20)])
def test_big_chunks_equal_histogram_partition(shape, n_seq, n, hot, bpw, monkeypatch):
    """The one-workgroup-per-CU scatter (chunks of up to 20 batches per wavefront: what a 10 M-event call takes by itself) forced
    onto small and ragged calls with frlw_tuning_t::batches_per_wave -- last chunks of a few events, wavefronts without any,
    sequences of different lengths -- against the histogram partition: state, view and uint8 volume bit for bit, with a state
    carried over from a first call.  (Its event loads are requested four batches ahead of the ranks)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import _lib, event_representation as er, synth
    H, W = shape
    K, win, n_win = 8, 10_000, 8
    recs = [synth.to_dat8(synth.synth_events(1500 + j, n // (1 + 2 * (j == 1)), W, H, n_win * win, hotspot=hot)) for j in range(n_seq)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = torch.from_numpy(np.ascontiguousarray(np.concatenate(recs)).view(np.uint8).reshape(-1, 8).copy()).cuda()
    outs = []
    for tuning in (_lib.FrlwTuning(chunk_major=1, direct_bins=0, batches_per_wave=bpw), _lib.FrlwTuning(chunk_major=0, direct_bins=0)):
        monkeypatch.setattr(er, "TUNING", tuning)
        st = torch.full((n_seq, H, W, 2, K), -6000.0, device="cuda")
        for _ in range(2):
            u8, view = er.encode_taf_batch(dat, offs, (H, W), st, 0, win, n_win, K, want_view=True)
        outs.append((st, u8, view))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_chunk_major_error_leaves_everything_untouched(monkeypatch):
    """The chunk-major scatter is also the validator (there is no histogram pass in front of it): an event behind the span sets
    the status, nothing is written, the deferred word carries it, and the next clean call on the workspace starts clean."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import _lib, event_representation as er, synth
    H, W, K = 240, 304, 8
    rec = synth.to_dat8(synth.synth_events(1500, 300_000, W, H, 80_000))
    bad = rec.copy()
    bad["t"][-3:] += 1_000_000
    dev = lambda r: torch.from_numpy(np.ascontiguousarray(r).view(np.uint8).reshape(-1, 8).copy()).cuda()
    er.raise_deferred()
    for cmaj in (1, 0):
        monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(chunk_major=cmaj))
        st = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
        with pytest.raises(ValueError):
            er.encode_taf_batch(dev(bad), [0, len(bad)], (H, W), st, 0, 10_000, 8, K)
        assert float((st != -6000.0).sum()) == 0.0
        er.encode_taf_batch(dev(bad), [0, len(bad)], (H, W), st, 0, 10_000, 8, K, check=False)
        with pytest.raises(ValueError):
            er.raise_deferred()
        er.encode_taf_batch(dev(rec), [0, len(rec)], (H, W), st, 0, 10_000, 8, K)  # clean: the header was reset by the call itself
        assert float((st != -6000.0).sum()) > 0.0


@pytest.mark.parametrize("shape,n_seq,n", [((240, 304), 2, 400_000), ((720, 1280), 1, 2_000_000)])
def test_captured_encode_replays_against_the_oracle(er, orc, shape, n_seq, n):
    """include/frlw_evd.h: calls can be captured into a hipGraph.  The chunk-major scatter resets the workspace header itself, keyed
    by a host-made epoch; frozen into a graph node that epoch would already be the published one on every replay after the first
    (stale window masks / flags).  A captured call therefore takes a memset node and epoch 0.  Here: capture ONE
    frlw_taf_encode_batch, replay it three times with DIFFERENT records in the same buffer -- data whose empty window moves from
    replay to replay, so a window mask that survived a replay would age cells the reference leaves alone -- and hold the carried
    state and the uint8 volume to the oracle after every replay (direct mode and tile bins; generate_taf.py:40-41, :193-235)."""
    H, W = shape
    K, win, n_win = 8, 10_000, 8

    def records(seed, empty_window):
        out = []
        for j in range(n_seq):
            ev = synth.synth_events(seed + j, n // n_seq, W, H, n_win * win)
            rec = synth.to_dat8(ev)
            kept = np.flatnonzero((rec["t"] // win) != empty_window)  # one window without a single event in the whole sequence
            # thinned to the same record count for every variant (the graph's offsets are frozen), stream order kept
            pick = np.sort(np.random.default_rng(seed + 50 + j).choice(kept, size=(n // n_seq) * 3 // 4, replace=False))
            out.append(rec[pick])
        return out

    variants = [records(9100, 3), records(9200, 5), records(9300, 3)]
    counts = [len(r) for r in variants[0]]
    for v in variants:
        assert [len(r) for r in v] == counts
    offs = np.concatenate([[0], np.cumsum(counts)])
    cat = lambda v: torch.from_numpy(np.ascontiguousarray(np.concatenate(v)).view(np.uint8).reshape(-1, 8).copy())
    side = torch.cuda.Stream()
    dat = cat(variants[0]).cuda()
    state = torch.full((n_seq, H, W, 2, K), -6000.0, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        warm = state.clone()  # eager call on the capture stream first: workspace, self-test and threshold table exist before the capture
        er.encode_taf_batch(dat, offs, (H, W), warm, 0, win, n_win, K)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            u8, _ = er.encode_taf_batch(dat, offs, (H, W), state, 0, win, n_win, K, check=False)
    want = [np.full((H, W, 2, K), -6000, np.float32) for _ in range(n_seq)]
    for v in variants:
        dat.copy_(cat(v).cuda())
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        for j in range(n_seq):
            _view, want[j], ou8 = oracle_taf(orc, v[j], (H, W), K, 0, win, n_win, want[j])
            assert_bitexact(host(state[j]), want[j], f"state of sequence {j}")
            assert_u8_budget(host(u8[j]), ou8, 1e-5, f"uint8 of sequence {j}")
    with torch.cuda.stream(side):
        er.raise_deferred()


@pytest.mark.parametrize("shape,n_seq,n,K,n_win,win,hot", [((720, 1280), 1, 3_000_000, 8, 8, 10_000, False),
                                                           ((240, 304), 12, 700_000, 8, 8, 10_000, True),    # tiles of three chunks, tiles over the segment limit
                                                           ((240, 304), 9, 300_000, 4, 64, 1_000, False),    # 64 windows
                                                           ((240, 304), 9, 200_000, 5, 1, 40_000, False),    # ONE window (no window bits in the record)
                                                           ((97, 131), 40, 30_000, 3, 5, 7_001, True)])
def test_walk_window_table_equals_the_walks_own_scan(shape, n_seq, n, K, n_win, win, hot, monkeypatch):
    """frlw_tuning_t::walk_window_table 1 (the default: kf_split_whole<true> leaves every sub-tile list's window starts in a table,
    kf_taf_walk reads them) against 0 (the walk scans its list), forced onto tile bins + the chunk-major partition: sequences
    with windows without any event (at the front, in the middle, at the end), an unsorted sequence (the tile is flagged and the
    walk filters per window), a sequence whose time runs backwards twice, an empty sequence, tiles that span several staging
    chunks and tiles the segment kernels place (no table: the walk scans) -- state, view and uint8 volume bit for bit, with a
    state carried over from a first call."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import _lib, event_representation as er, synth
    H, W = shape
    span = n_win * win
    recs = [synth.to_dat8(synth.synth_events(1700 + j, n // (1 + 3 * (j == 1)), W, H, span, hotspot=hot and j % 3 == 0)) for j in range(n_seq)]
    if n_seq >= 5:
        if n_win >= 3:
            r = recs[2]
            recs[2] = r[((r["t"] // win) != 1) & ((r["t"] // win) != 0)]        # no event in windows 0 and 1
            r = recs[3]
            recs[3] = r[(r["t"] // win) < n_win - 2]                             # none in the last two
        recs[4] = recs[4][np.random.default_rng(11).permutation(len(recs[4]))]   # unsorted
        recs[0] = np.concatenate([recs[0]] * 3)                                  # time runs backwards twice
        recs[-1] = recs[-1][:0]                                                  # empty
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = torch.from_numpy(np.ascontiguousarray(np.concatenate(recs)).view(np.uint8).reshape(-1, 8).copy()).cuda()
    outs = []
    for wt in (1, 0):
        monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(chunk_major=1, direct_bins=0, walk_window_table=wt))
        st = torch.full((n_seq, H, W, 2, K), -6000.0, device="cuda")
        for _ in range(2):
            u8, view = er.encode_taf_batch(dat, offs, (H, W), st, 0, win, n_win, K, want_view=True)
        outs.append((st, u8, view))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    # ... and against the histogram partition (no table anywhere)
    monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(chunk_major=0, direct_bins=0))
    st = torch.full((n_seq, H, W, 2, K), -6000.0, device="cuda")
    for _ in range(2):
        u8, view = er.encode_taf_batch(dat, offs, (H, W), st, 0, win, n_win, K, want_view=True)
    for a, b in zip(outs[0], (st, u8, view)):
        assert torch.equal(a, b)


def test_stall_verdict_survives_the_header_reset_and_is_cleared_by_the_next_call(er):
    """a workgroup that gives up its bounded wait for the header reset used to OR ST_STALL into hdr->status, which
    a workgroup 0 that starts later zeroes.  The verdict now lives in a word outside the reset range (8 bytes in front of the
    sticky word, keyed by the call's epoch): frlw_encoder_status reports FRLW_ERR_HIP while it is set, and workgroup 0 of a LATER
    call clears an EARLIER call's verdict (a stall cannot be provoked from outside: the word is planted by hand here)."""
    import ctypes as C
    from frlw_evd_amd import _lib
    lib = _lib.load()
    H, W, K = 240, 304, 8
    rec = synth.to_dat8(synth.synth_events(1900, 300_000, W, H, 80_000))
    dat = to_dev(rec)
    st = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    er.encode_taf_batch(dat, [0, len(rec)], (H, W), st, 0, 10_000, 8, K)  # allocates / initialises the batch workspace
    ws = er._WORKSPACES[("batch", dat.device.index, torch.cuda.current_stream().cuda_stream)]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    status = C.c_int(0)
    ws[1024 - 16:1024 - 12].view(torch.int32).fill_(0x7fffff01)  # "the call with epoch 0x7fffff01 stalled"
    _lib.check(lib.frlw_encoder_status(C.c_void_p(ws.data_ptr()), stream, C.byref(status)), "status")
    assert status.value == _lib.FRLW_ERR_HIP
    st2 = torch.full((1, H, W, 2, K), -6000.0, device="cuda")
    er.encode_taf_batch(dat, [0, len(rec)], (H, W), st2, 0, 10_000, 8, K)  # a later call: its workgroup 0 drops the old verdict
    _lib.check(lib.frlw_encoder_status(C.c_void_p(ws.data_ptr()), stream, C.byref(status)), "status")
    assert status.value == 0
    assert int(ws[1024 - 16:1024 - 12].view(torch.int32).item()) == 0
    assert torch.equal(st, st2)
