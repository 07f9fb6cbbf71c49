"""HIP encoders (through the C-ABI, via the reference-signature shims) against the CPU oracle, the
reference-generated golden vectors, and size-independent properties at BASELINE sizes.

Bars: bit-exact f32 for everything that is add / sub / mul / div / max in the
reference; directly after expf / log1pf the f32 values agree within 2 ulp and the uint8 artefact may
differ by exactly 1 LSB in at most 1e-5 of the elements (1e-4 on the small samples).
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from frlw_evd_amd import synth  # noqa: E402
from golden_util import (GEN1, LAMDAS, MPX, assert_big, assert_bitexact, assert_u8_budget,  # noqa: E402
                         downscale_maps, sha)

pytestmark = pytest.mark.gpu

U8_BUDGET = 1e-5


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


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dat_dev(ev):
    return torch.from_numpy(synth.to_dat8(ev).view(np.uint8).reshape(-1, 8).copy()).cuda()


def host(t):
    return t.cpu().numpy()


def path_counts():
    """frlw_encoder_path_counts: [SAE two-launch, SAE general, ECI two-launch, ECI scan / general] launches of this process."""
    import ctypes as C
    from frlw_evd_amd import _lib
    c = (C.c_uint64 * 4)()
    _lib.check(_lib.load().frlw_encoder_path_counts(c), "frlw_encoder_path_counts")
    return np.array(list(c), dtype=np.int64)


def assert_ulp(got, want, ulps, what):
    a = np.ascontiguousarray(got).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(want).view(np.int32).astype(np.int64)
    assert np.abs(a - b).max() <= ulps, f"{what}: {np.abs(a - b).max()} ulp"


# ------------------------------------------------------------------------------------------
# tiny hand-checkable cases through the reference signatures, against the reference's goldens
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tiny(golden_dir):
    return np.load(os.path.join(golden_dir, "tiny.npz"))


def test_tiny_eci_ev(er, tiny):
    H, W = (int(v) for v in tiny["shape"])
    ev = dev(tiny["events"])
    out, dt = er.generate_eventframe(ev, (H, W))
    assert_bitexact(host(out), tiny["eci"], "eci")
    assert dt >= 0
    out, _ = er.generate_agile_event_volume_cuda(ev, (H, W), 50000, 5)
    assert_bitexact(host(out), tiny["ev"], "ev")
    out, _ = er.generate_agile_event_volume_cuda(ev, (H, W), 50000, 3)
    assert_bitexact(host(out), tiny["ev_bins3"], "ev bins=3")


def test_tiny_sae(er, tiny):
    H, W = (int(v) for v in tiny["shape"])
    se, half, now = tiny["sae_events"], int(tiny["sae_half"]), tiny["sae_now"]
    o1, m1, _ = er.generate_leaky_cuda(dev(se[:half]), (H, W), LAMDAS, None, now[0])
    assert_bitexact(host(m1), tiny["sae_mem1"], "sae memory 1")
    assert_ulp(host(o1), tiny["sae_out1"], 2, "sae out 1")
    o2, m2, _ = er.generate_leaky_cuda(dev(se[half:]), (H, W), LAMDAS, m1, now[1])
    assert_bitexact(host(m2), tiny["sae_mem2"], "sae memory 2")
    assert_ulp(host(o2), tiny["sae_out2"], 2, "sae out 2")


@pytest.mark.parametrize("K", [8, 4])
def test_tiny_taf(er, tiny, K):
    H, W = (int(v) for v in tiny["shape"])
    ev, sp = tiny["events"], tiny["taf_splits"]
    st = torch.full((H, W, 2, K), -6000.0, device="cuda")
    for i in range(4):
        w = ev[sp[i]:sp[i + 1]]
        w5 = np.concatenate([w, np.zeros((len(w), 1))], axis=1)  # the z column, generate_taf.py:203
        before = st.clone()
        view, st2, _ = er.generate_taf_cuda(dev(w5), (H, W), st, K)
        assert torch.equal(st, before), "past_volume must not be mutated"
        st = st2
        if K == 8:
            assert_bitexact(host(view), tiny[f"taf_view{i}"], f"taf view {i}")
            assert_bitexact(host(st), tiny[f"taf_state{i}"], f"taf state {i}")
    if K == 8:
        assert_ulp(host(er.leaky_transform(view)), tiny["taf_leaky"], 2, "leaky_transform")
    else:
        assert_bitexact(host(st), tiny["taf_k4_state"], "taf K=4 state")
        assert_bitexact(host(view), tiny["taf_k4_view"], "taf K=4 view")


def test_taf_growing_branch(er, golden_dir):
    """generate_taf.py:50-53: ``past_volume`` with volume_bins - 1 slots grows to volume_bins; goldens from the reference."""
    g = np.load(os.path.join(golden_dir, "tiny_taf_grow.npz"))
    tiny = np.load(os.path.join(golden_dir, "tiny.npz"))
    H, W = (int(v) for v in tiny["shape"])
    w = tiny["events"][:150]
    w5 = np.concatenate([w, np.zeros((len(w), 1))], axis=1)
    for K in (8, 4):
        past = torch.from_numpy(g[f"k{K}_past"]).cuda()
        view, st, _ = er.generate_taf_cuda(dev(w5), (H, W), past, K)
        assert st.shape == (H, W, 2, K)
        assert_bitexact(host(st), g[f"k{K}_state"], f"K={K} grown state")
        assert_bitexact(host(view), g[f"k{K}_view"], f"K={K} grown view")
        with pytest.raises(RuntimeError):  # an empty window keeps the short volume: the reference's .view raises
            er.generate_taf_cuda(dev(w5[:0]), (H, W), past, K)
        with pytest.raises(RuntimeError):  # any other slot count fails that .view as well
            er.generate_taf_cuda(dev(w5), (H, W), past[..., :K - 2], K)


def test_errors_like_torch(er):
    oob = dev(np.array([[0.0, 8.0, 0.5, 1.0]]))  # flat index past the end
    with pytest.raises(IndexError):
        er.generate_eventframe(oob, (8, 12))
    with pytest.raises(IndexError):
        er.generate_agile_event_volume_cuda(oob, (8, 12))
    with pytest.raises(IndexError):
        er.generate_taf_cuda(oob, (8, 12), torch.zeros((8, 12, 2, 8), device="cuda"), 8)
    # x >= W alone aliases into the next row, like the flat index of the reference
    out, _ = er.generate_eventframe(dev(np.array([[13.0, 1.0, 0.5, 1.0]])), (8, 12))
    out = host(out)
    assert out[1, 2, 1] > 0 and np.count_nonzero(out) == 1
    # SAE filters (generate_surfaceofactiveevents.py:72)
    o, mem, _ = er.generate_leaky_cuda(oob, (8, 12), LAMDAS, None, 100)
    assert np.all(host(mem) == np.float32(100) - np.float32(5000000))
    with pytest.raises(RuntimeError):
        er.generate_eventframe(torch.zeros((1, 4), dtype=torch.float64), (8, 12))  # CPU tensor: no fallback


def test_empty_stream(er):
    ev = torch.zeros((0, 4), dtype=torch.float64, device="cuda")
    out, _ = er.generate_eventframe(ev, (8, 12))
    assert torch.count_nonzero(out) == 0
    out, _ = er.generate_agile_event_volume_cuda(ev, (8, 12))
    assert torch.count_nonzero(out) == 0
    st = torch.full((8, 12, 2, 8), -6000.0, device="cuda")
    view, st2, _ = er.generate_taf_cuda(ev, (8, 12), st, 8)
    assert torch.equal(st2, st)  # all(forward) -> unchanged (generate_taf.py:40-41)
    assert torch.equal(view, st.permute(3, 2, 0, 1).reshape(16, 8, 12))


# ------------------------------------------------------------------------------------------
# random ragged shapes: HIP vs oracle on the same seeded inputs (f64-tensor path)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,W,n,seed", [(8, 12, 500, 1), (37, 70, 20_000, 2), (240, 304, 200_000, 3),
                                         (9, 33, 5_000, 4), (64, 64, 100_000, 5)])
def test_random_vs_oracle_f64(er, orc, H, W, n, seed):
    ev = synth.synth_events(seed, n, W, H, 100_000, hotspot=bool(seed & 1))
    t = ev["t"] / 100_000.0
    e = synth.to_xytp_f64(ev, t)
    ed = dev(e)
    assert_bitexact(host(er.generate_eventframe(ed, (H, W))[0]), orc.eventframe(e, (H, W)), "eci")
    for bins in (5, 2):
        assert_bitexact(host(er.generate_agile_event_volume_cuda(ed, (H, W), 0, bins)[0]),
                        orc.event_volume(e, (H, W), bins), f"ev bins={bins}")
    st0 = np.random.default_rng(seed).uniform(-50, 0, size=(H, W, 2, 6)).astype(np.float32)
    view, st, _ = er.generate_taf_cuda(ed, (H, W), dev(st0), 6)
    oview, ost = orc.taf_window(e, (H, W), st0, 6)
    assert_bitexact(host(st), ost, "taf state")
    assert_bitexact(host(view), oview, "taf view")
    e_abs = synth.to_xytp_f64(ev, ev["t"] + 40_000_000)
    o, mem, _ = er.generate_leaky_cuda(dev(e_abs), (H, W), LAMDAS, None, 40_100_000)
    oo, omem = orc.sae(e_abs, (H, W), LAMDAS, None, 40_100_000)
    assert_bitexact(host(mem), omem, "sae memory")
    assert_ulp(host(o), oo, 2, "sae out")
    assert_u8_budget(host(er.quantize_u8(o)), orc.quantize_u8(oo), 1e-4, "sae u8")


def test_taf_unsorted_stream(er, orc):
    """Stream order, not time order, defines the result: shuffle the records (windows interleave)."""
    H, W, K = 40, 70, 8
    ev = synth.synth_events(21, 60_000, W, H, 80_000, hotspot=True)
    perm = np.random.default_rng(5).permutation(60_000)
    ev = {k: v[perm] for k, v in ev.items()}
    dat = synth.to_dat8(ev)
    st0 = np.full((H, W, 2, K), -6000, np.float32)
    oview, ost = orc.taf_stream_dat8(dat, (H, W), (H, W), K, 0, 10_000, 8, st0)
    st = dev(st0)
    u8, view = er.encode_taf_dat(dat_dev(ev), (H, W), st, 0, 10_000, 8, K, want_view=True)
    assert_bitexact(host(st), ost, "unsorted taf state")
    assert_bitexact(host(view), oview, "unsorted taf view")


@pytest.mark.parametrize("thr", [0, 300, 3000])
def test_hot_tile_sharing_forced(er, orc, monkeypatch, thr):
    """Skew path: with frlw_tuning_t::hot_tile_records forced low, (up to 32) tiles are shared by several workgroups,
    each summing only its own 128 cells; more than 32 hot tiles -> the rest stay whole.  Same bits either way, EV and
    TAF (general path), sorted and shuffled streams."""
    from frlw_evd_amd import _lib
    monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(hot_tile_records=thr))
    H, W, K = 100, 300, 8
    ev = synth.synth_events(31 + thr, 300_000, W, H, 80_000, hotspot=True)
    dat = synth.to_dat8(ev)
    st0 = np.random.default_rng(thr).uniform(-50, 0, size=(H, W, 2, K)).astype(np.float32)
    oview, ost = orc.taf_stream_dat8(dat, (H, W), (H, W), K, 0, 10_000, 8, st0)
    st = dev(st0)
    _, view = er.encode_taf_dat(dat_dev(ev), (H, W), st, 0, 10_000, 8, K, want_view=True)
    assert_bitexact(host(st), ost, "shared taf state")
    assert_bitexact(host(view), oview, "shared taf view")
    f32 = er.encode_ev_dat(dat_dev(ev), (H, W), 80_000, 80_000, volume_bins=5)[0]
    assert_bitexact(host(f32), orc.ev_stream_dat8(dat, (H, W), (H, W), 5, 80_000, 80_000), "shared ev")
    perm = np.random.default_rng(6).permutation(len(ev["t"]))
    evs = {k: v[perm] for k, v in ev.items()}
    dats = synth.to_dat8(evs)
    oview, ost = orc.taf_stream_dat8(dats, (H, W), (H, W), K, 0, 10_000, 8, st0)
    st = dev(st0)
    er.encode_taf_dat(dat_dev(evs), (H, W), st, 0, 10_000, 8, K)
    assert_bitexact(host(st), ost, "shared taf state, shuffled stream")


def test_taf_outside_window_span(er, orc):
    """Events before t_start / after the last window fall into window 0 (generate_taf.py:197-203)."""
    H, W, K = 16, 40, 8
    ev = synth.synth_events(22, 20_000, W, H, 120_000)
    dat = synth.to_dat8(ev)
    st0 = np.full((H, W, 2, K), -6000, np.float32)
    oview, ost = orc.taf_stream_dat8(dat, (H, W), (H, W), K, 20_000, 10_000, 8, st0)
    st = dev(st0)
    er.encode_taf_dat(dat_dev(ev), (H, W), st, 20_000, 10_000, 8, K)
    assert_bitexact(host(st), ost, "taf with outsiders")


# ------------------------------------------------------------------------------------------
# GEN1-shaped goldens through the fused DAT path (SURVEY.md section 8d cfg 1, 2, 5)
# ------------------------------------------------------------------------------------------
def test_gen1_eci(er, golden_dir):
    g = np.load(os.path.join(golden_dir, "gen1_eci.npz"))
    shape, tshape = GEN1
    for hot, nat_key, u8_key in ((False, "eci_native", "eci_u8"), (True, None, "eci_hot_u8")):
        ev = synth.synth_events(1001, 100_000, shape[1], shape[0], 50_000, hotspot=hot)
        out, u8 = er.encode_eci_dat(dat_dev(ev), shape, want_u8=True)
        if nat_key:
            assert_bitexact(host(out), g[nat_key], "eci native")
        else:
            assert sha(host(out)) == str(g["eci_hot_native_sha"])
        assert_bitexact(host(er.resize_nearest(u8, tshape)), g[u8_key], "eci u8")
        assert_bitexact(host(er.quantize_u8(er.resize_nearest(out, tshape))), g[u8_key], "eci u8 via f32")


@pytest.mark.parametrize("tag,hot", [("", False), ("hot_", True)])
def test_gen1_ev(er, golden_dir, tag, hot):
    g = np.load(os.path.join(golden_dir, "gen1_ev.npz"))
    shape, tshape = GEN1
    ev = synth.synth_events(1002, 1_000_000, shape[1], shape[0], 250_000, hotspot=hot)
    out, u8 = er.encode_ev_dat(dat_dev(ev), shape, 250_000, 250_000, 5, want_u8=True)
    assert_big(host(out), g, tag + "native", "ev native")
    assert_bitexact(host(er.resize_nearest(u8, tshape)), g[tag + "u8"], "ev u8")


def test_gen1_sae(er, golden_dir):
    g = np.load(os.path.join(golden_dir, "gen1_sae.npz"))
    shape, tshape = GEN1
    ev = synth.synth_events(1006, 1_000_000, shape[1], shape[0], 5_000_000, t_offset=30_000_000)
    dat = dat_dev(ev)
    cut, now = int(g["cut"]), g["now"]
    o1, u1, m1 = er.encode_sae_dat(dat[:cut], shape, LAMDAS, None, now[0], 5541263, want_u8=True)
    assert_big(host(m1), g, "mem1", "sae mem1")
    assert_u8_budget(host(er.resize_nearest(u1, tshape)), g["u8_1"], U8_BUDGET, "sae u8 1")
    o2, u2, m2 = er.encode_sae_dat(dat[cut:], shape, LAMDAS, m1, now[1], 5541263, want_u8=True)
    assert_big(host(m2), g, "mem2", "sae mem2")
    assert_u8_budget(host(er.resize_nearest(u2, tshape)), g["u8_2"], U8_BUDGET, "sae u8 2")


@pytest.mark.parametrize("tag,hot", [("", False), ("hot_", True)])
def test_gen1_taf(er, golden_dir, tag, hot):
    g = np.load(os.path.join(golden_dir, "gen1_taf.npz"))
    shape, tshape = GEN1
    K = 8
    ev = synth.synth_events(1005, 1_000_000, shape[1], shape[0], 80_000, hotspot=hot)
    st = torch.full((*shape, 2, K), -6000.0, device="cuda")
    u8, view = er.encode_taf_dat(dat_dev(ev), shape, st, 0, 10_000, 8, K, want_view=True)
    assert_big(host(st), g, tag + "state", "taf state")
    assert_big(host(view), g, tag + "native", "taf view")
    u8r = er.resize_nearest(u8.reshape(2 * K, *shape), tshape).reshape(K, 2, *tshape)
    assert_u8_budget(host(u8r), g[tag + "u8"], U8_BUDGET, "taf u8")
    if not hot:  # second label with the state carried over (generate_taf.py:180-186)
        ev2 = synth.synth_events(2005, 300_000, shape[1], shape[0], 30_000, t_offset=80_000)
        u8, _ = er.encode_taf_dat(dat_dev(ev2), shape, st, 80_000, 10_000, 3, K)
        assert_big(host(st), g, "carry_state", "taf carry state")
        u8r = er.resize_nearest(u8.reshape(2 * K, *shape), tshape).reshape(K, 2, *tshape)
        assert_u8_budget(host(u8r), g["carry_u8"], U8_BUDGET, "taf carry u8")


# ------------------------------------------------------------------------------------------
# BASELINE.json full sizes: 1 Mpx goldens + size-independent properties
# ------------------------------------------------------------------------------------------
def test_mpx_taf_native_golden(er, golden_dir):
    """cfg 3: TAF K=8, 10 M events, 1280x720, 8 windows -- state sha256 equals the reference's."""
    g = np.load(os.path.join(golden_dir, "mpx_taf_native.npz"))
    shape = MPX[0]
    K = 8
    ev = synth.synth_events(1003, 10_000_000, shape[1], shape[0], 80_000)
    st = torch.full((*shape, 2, K), -6000.0, device="cuda")
    u8, _ = er.encode_taf_dat(dat_dev(ev), shape, st, 0, 10_000, 8, K)
    assert_big(host(st), g, "state", "mpx taf state")
    assert_u8_budget(host(u8).reshape(-1)[g["u8_idx"]], g["u8_val"], 1e-4, "mpx taf u8 sample")


def test_mpx_downscale_golden(er, golden_dir):
    g = np.load(os.path.join(golden_dir, "mpx_downscale.npz"))
    shape, tshape = MPX
    K = 8
    ev = synth.synth_events(1013, 2_000_000, shape[1], shape[0], 80_000, hotspot=True)
    dat = dat_dev(ev)
    xm, ym = er.coordinate_maps(shape, tshape, "cuda")
    xmap, ymap = downscale_maps(shape, tshape)
    assert np.array_equal(host(xm).astype(np.uint16), xmap) and np.array_equal(host(ym).astype(np.uint16), ymap)
    st = torch.full((*tshape, 2, K), -6000.0, device="cuda")
    er.encode_taf_dat(dat, tshape, st, 0, 10_000, 8, K, xmap=xm, ymap=ym)
    assert_big(host(st), g, "state", "downscale taf state")
    out, _ = er.encode_ev_dat(dat, tshape, 80_000, 80_000, 5, xmap=xm, ymap=ym)
    assert_big(host(out), g, "ev_native", "downscale ev")
    out, _ = er.encode_eci_dat(dat[-200_000:], tshape, xmap=xm, ymap=ym)
    assert_big(host(out), g, "eci_native", "downscale eci")


def test_mpx_properties(er):
    """Size-independent properties at the cfg-3 size (no oracle run needed)."""
    shape = MPX[0]
    K = 8
    ev = synth.synth_events(77, 10_000_000, shape[1], shape[0], 80_000, hotspot=True)
    dat = dat_dev(ev)
    # (1) one 8-window call == a 5-window call followed by a 3-window call on the carried state
    st_a = torch.full((*shape, 2, K), -6000.0, device="cuda")
    er.encode_taf_dat(dat, shape, st_a, 0, 10_000, 8, K, want_u8=False)
    cut = int(np.searchsorted(ev["t"], 50_000, side="left"))
    st_b = torch.full((*shape, 2, K), -6000.0, device="cuda")
    er.encode_taf_dat(dat[:cut], shape, st_b, 0, 10_000, 5, K, want_u8=False)
    er.encode_taf_dat(dat[cut:], shape, st_b, 50_000, 10_000, 3, K, want_u8=False)
    assert torch.equal(st_a, st_b)
    # (2) deterministic run to run
    st_c = torch.full((*shape, 2, K), -6000.0, device="cuda")
    er.encode_taf_dat(dat, shape, st_c, 0, 10_000, 8, K, want_u8=False)
    assert torch.equal(st_a, st_c)
    # (3) ECI is a pure count: invariant under any permutation of the stream, and sums to the event count
    out, _ = er.encode_eci_dat(dat[:2_000_000], shape)
    perm = torch.randperm(2_000_000, device="cuda")
    out_p, _ = er.encode_eci_dat(dat[:2_000_000][perm], shape)
    assert torch.equal(out, out_p)
    # (4) SAE memory is idempotent: feeding the memory back with no events returns it unchanged
    o, _, mem = er.encode_sae_dat(dat[:2_000_000], shape, LAMDAS, None, 80_000, 0)
    _, _, mem2 = er.encode_sae_dat(dat[:0], shape, LAMDAS, mem, 80_000, 0)
    assert torch.equal(mem, mem2)


# ------------------------------------------------------------------------------------------
# parameter sweeps through the fused DAT path: HIP vs oracle, bit-exact
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,n_windows,window_us,H,W", [(4, 1, 10_000, 64, 96), (5, 13, 7_000, 40, 200),
                                                      (8, 64, 1_250, 72, 130), (1, 3, 50_000, 16, 300),
                                                      (7, 8, 10_000, 240, 304)])
def test_taf_dat_parameter_sweep(er, orc, K, n_windows, window_us, H, W):
    span = n_windows * window_us
    ev = synth.synth_events(100 + K + n_windows, 150_000, W, H, span + window_us // 2, hotspot=True, t_offset=5_000)
    dat = synth.to_dat8(ev)
    st0 = np.random.default_rng(K).uniform(-100, 0, size=(H, W, 2, K)).astype(np.float32)
    oview, ost = orc.taf_stream_dat8(dat, (H, W), (H, W), K, 4_000, window_us, n_windows, st0)
    st = dev(st0)
    u8, view = er.encode_taf_dat(dat_dev(ev), (H, W), st, 4_000, window_us, n_windows, K, want_view=True, flip_k=False)
    assert_bitexact(host(st), ost, "taf state")
    assert_bitexact(host(view), oview, "taf view")
    want_u8 = orc.quantize_u8(orc.leaky_transform(oview.reshape(K, 2, H, W)))
    assert_u8_budget(host(u8), want_u8, 1e-4, "taf u8")


@pytest.mark.parametrize("bins", [1, 2, 5, 8])
def test_ev_dat_bins_and_time_filter(er, orc, bins):
    H, W = 48, 200
    ev = synth.synth_events(300 + bins, 120_000, W, H, 400_000)
    dat = synth.to_dat8(ev)
    want = orc.ev_stream_dat8(dat, (H, W), (H, W), bins, 400_000, 250_000)  # drops t <= 150 000
    out, u8 = er.encode_ev_dat(dat_dev(ev), (H, W), 400_000, 250_000, bins, want_u8=True)
    assert_bitexact(host(out), want, f"ev bins={bins}")
    assert_bitexact(host(u8), orc.quantize_u8(want, clip255=True), "ev u8")


def test_sae_dat_window_filter_and_downscale(er, orc):
    sensor, shape = (120, 200), (60, 100)
    ev = synth.synth_events(41, 80_000, sensor[1], sensor[0], 3_000_000, t_offset=10_000_000)
    dat = synth.to_dat8(ev)
    xm, ym = er.coordinate_maps(sensor, shape, "cuda")
    now = 13_000_000
    want, wmem = orc.sae_stream_dat8(dat, sensor, shape, LAMDAS, None, now, 1_000_000)
    out, _, mem = er.encode_sae_dat(dat_dev(ev), shape, LAMDAS, None, now, 1_000_000, xmap=xm, ymap=ym)
    assert_bitexact(host(mem), wmem, "sae memory (filter + downscale)")
    assert_ulp(host(out), want, 2, "sae out")


def test_workspace_reuse_across_shapes_and_kinds(er, orc):
    """One cached workspace serves calls of different kinds / sizes back to back (no stale state)."""
    for H, W, n in ((240, 304, 50_000), (8, 12, 300), (720, 1280, 400_000), (240, 304, 50_000)):
        ev = synth.synth_events(n, n, W, H, 80_000)
        dat = synth.to_dat8(ev)
        out, _ = er.encode_eci_dat(dat_dev(ev), (H, W))
        assert_bitexact(host(out), orc.eci_stream_dat8(dat, (H, W), (H, W)), "eci")
        st0 = np.full((H, W, 2, 8), -6000, np.float32)
        _, ost = orc.taf_stream_dat8(dat, (H, W), (H, W), 8, 0, 10_000, 8, st0)
        st = dev(st0)
        er.encode_taf_dat(dat_dev(ev), (H, W), st, 0, 10_000, 8, 8)
        assert_bitexact(host(st), ost, "taf")


@pytest.mark.parametrize("n", [0, 3000])
def test_taf_u8_with_mostly_empty_tiles(er, orc, n):
    """A wide frame (W > 512: 1024-thread tile workgroups) whose events sit in one corner, and a stream with no events
    at all: the uint8 planes of the empty tiles come from the threshold table alone, which every wavefront must see
    complete before the write-out (the barrier after the table fill in taf_tile_body)."""
    H, W, K, n_windows, window_us = 64, 640, 8, 4, 10_000
    ev = synth.synth_events(77, n, 40, 16, n_windows * window_us, t_offset=0)  # n = 0: empty arrays
    st0 = np.random.default_rng(9).uniform(-300, 0, size=(H, W, 2, K)).astype(np.float32)
    dat = synth.to_dat8(ev)
    oview, ost = orc.taf_stream_dat8(dat, (H, W), (H, W), K, 0, window_us, n_windows, st0)
    want_u8 = orc.quantize_u8(orc.leaky_transform(oview.reshape(K, 2, H, W)))
    d = torch.from_numpy(dat.view(np.uint8).reshape(-1, 8).copy()).cuda()
    for _ in range(3):  # timing-dependent bugs need more than one try
        st = dev(st0)
        u8, view = er.encode_taf_dat(d, (H, W), st, 0, window_us, n_windows, K, want_view=True, flip_k=False, fast=False)
        assert_bitexact(host(st), ost, "taf state")
        assert_bitexact(host(view), oview, "taf view")
        assert_u8_budget(host(u8), want_u8, 1e-4, "taf u8")


def test_ev_samples_laid_out_in_one_frame_equal_their_own_encodes(er):
    """Event Volume has no per-sequence rule: independent samples placed side by side (2 across x 2 down here, the layout
    of bench.py's ev_gen1 x64 row) come out of ONE encode bit-identical to their own encodes."""
    H, W, n, span = 40, 72, 30_000, 250_000
    parts, singles = [], []
    for j in range(4):
        e = dict(synth.synth_events(300 + j, n, W, H, span, hotspot=bool(j & 1)))
        out, _ = er.encode_ev_dat(dat_dev(e), (H, W), span, span, volume_bins=5)
        singles.append(out)
        e["x"] = e["x"] + (j % 2) * W
        e["y"] = e["y"] + (j // 2) * H
        parts.append(synth.to_dat8(e))
    dat = torch.from_numpy(np.concatenate(parts).view(np.uint8).reshape(-1, 8).copy()).cuda()
    both, _ = er.encode_ev_dat(dat, (2 * H, 2 * W), span, span, volume_bins=5)
    for j in range(4):
        r0, c0 = (j // 2) * H, (j % 2) * W
        assert torch.equal(both[:, r0:r0 + H, c0:c0 + W], singles[j]), j


@pytest.mark.parametrize("shape,n,maps", [((240, 304), 100_000, False), ((17, 33), 3_000, False), ((97, 131), 60_000, False),
                                          ((512, 640), 12_000, True), ((240, 304), 1, False), ((256, 320), 90_000, True),
                                          ((240, 304), 15_000, False)])
def test_eci_single_launch_equals_general_path_and_oracle(er, orc, shape, n, maps, monkeypatch):
    """Small Event Count Image calls take TWO launches through the chunk-major partition (kf_scatter_cm + the counting form of
    kf_sae_sub: 16 384 events and more on frames of the GEN1 class) or ONE (k_eci_scan: every workgroup counts its 2048 pixels
    over all events, generate_eventcountimage.py:19-41) -- against the five-launch general path (forced by the tuning knob) and the oracle, bit for
    bit: hot pixels beyond the 20-add saturation, x >= W aliasing into the next row, down-scale maps, a single event; and an
    event outside the frame raises IndexError on both paths."""
    from frlw_evd_amd import _lib
    H, W = shape
    Hs, Ws = (720, 1280) if maps else (H, W)
    ev = synth.synth_events(7100 + n, n, Ws, Hs, 50_000, hotspot=n > 10)
    if n > 100:
        ev["x"][:40] = Ws // 3  # one pixel far beyond 20 events
        ev["y"][:40] = Hs // 3
        ev["p"][:40] = 1
    if not maps and n > 100:
        ev["x"][50:60] = W + 2  # aliases into the next row like the reference's flat index
        ev["y"][50:60] = 3
    rec = synth.to_dat8(ev)
    xm = ym = None
    if maps:
        xm, ym = er.coordinate_maps((Hs, Ws), (H, W), "cuda")
    dat = torch.from_numpy(rec.view(np.uint8).reshape(-1, 8).copy()).cuda()
    c0 = path_counts()
    out1, u1 = er.encode_eci_dat(dat, (H, W), want_u8=True, xmap=xm, ymap=ym)
    c1 = path_counts()
    monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(staged_scatter=0))  # the general path
    out0, u0 = er.encode_eci_dat(dat, (H, W), want_u8=True, xmap=xm, ymap=ym)
    monkeypatch.setattr(er, "TUNING", None)
    c2 = path_counts()
    # WHICH form ran (the equality below holds on a silent fall-back too): the two-launch form for the GEN1-class calls of at
    # least 16 384 events, the scan / general path below that and under the knob
    two_launch = n >= 16384 and (H, W) in ((240, 304), (97, 131), (256, 320))
    assert list(c1 - c0) == ([0, 0, 1, 0] if two_launch else [0, 0, 0, 1]), (c0, c1)
    assert list(c2 - c1) == [0, 0, 0, 1], (c1, c2)
    assert torch.equal(out1, out0) and torch.equal(u1, u0)
    assert_bitexact(host(out1), orc.eci_stream_dat8(rec, (Hs, Ws), (H, W)), "single-launch eci vs oracle")
    if not maps and n > 100:
        bad = rec.copy()
        bad["_"][7] = (np.uint32(W - 1) & 16383) | (np.uint32(H + 5) << 14)  # flat index behind the frame
        datb = torch.from_numpy(bad.view(np.uint8).reshape(-1, 8).copy()).cuda()
        for tun in (None, _lib.FrlwTuning(staged_scatter=0)):
            monkeypatch.setattr(er, "TUNING", tun)
            with pytest.raises(IndexError):
                er.encode_eci_dat(datb, (H, W))
        monkeypatch.setattr(er, "TUNING", None)
        er.encode_eci_dat(dat, (H, W))  # the next clean call starts clean


@pytest.mark.parametrize("shape,n,maps,shuffle", [((240, 304), 1_000_000, False, False), ((240, 304), 300_000, False, True),
                                                  ((97, 131), 40_000, False, False), ((256, 320), 200_000, True, False)])
def test_sae_two_launch_form_equals_general_path(er, orc, shape, n, maps, shuffle, monkeypatch):
    """Surface of Active Events, single calls of the GEN1 class: the chunk-major scatter + kf_sae_sub (two launches) against the
    five-launch general path (forced by the tuning knob), bit for bit in the memory AND the outputs (same expf), and against the
    oracle's memory: last writer in STREAM order also on a shuffled stream (generate_surfaceofactiveevents.py:49), events
    outside the frame and events at or in front of now - window dropped (:72, :176-190), memory carried over two calls, a hot
    spot, down-scale maps."""
    from frlw_evd_amd import _lib
    H, W = shape
    Hs, Ws = (720, 1280) if maps else (H, W)
    now, win = 40_000_000, 5_541_263
    ev = synth.synth_events(7300 + n % 1000, n, Ws, Hs, 7_000_000, hotspot=True, t_offset=33_500_000)  # some events in front of now - window
    if not maps:
        ev["x"][::211] = W + 3   # outside the frame: dropped, not an error
        ev["y"][5::977] = H + 1
    if shuffle:
        perm = np.random.default_rng(3).permutation(n)
        ev = {k: v[perm] for k, v in ev.items()}
    rec = synth.to_dat8(ev)
    xm = ym = None
    if maps:
        xm, ym = er.coordinate_maps((Hs, Ws), (H, W), "cuda")
    dat = torch.from_numpy(rec.view(np.uint8).reshape(-1, 8).copy()).cuda()
    half = n // 2
    res = []
    for tun in (None, _lib.FrlwTuning(staged_scatter=0)):
        monkeypatch.setattr(er, "TUNING", tun)
        c0 = path_counts()
        o1, u1, m1 = er.encode_sae_dat(dat[:half], (H, W), LAMDAS, None, now - 1_000_000, win, want_u8=True, xmap=xm, ymap=ym)
        o2, u2, m2 = er.encode_sae_dat(dat[half:], (H, W), LAMDAS, m1, now, win, want_u8=True, xmap=xm, ymap=ym)
        # WHICH form ran: both calls through the two-launch form by default, both through the general path under the knob
        assert list(path_counts() - c0) == ([2, 0, 0, 0] if tun is None else [0, 2, 0, 0]), tun
        res.append((o1, u1, m1, o2, u2, m2))
    monkeypatch.setattr(er, "TUNING", None)
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    _, om1 = orc.sae_stream_dat8(rec[:half], (Hs, Ws), (H, W), LAMDAS, None, now - 1_000_000, win)
    _, om2 = orc.sae_stream_dat8(rec[half:], (Hs, Ws), (H, W), LAMDAS, om1, now, win)
    assert_bitexact(host(res[0][5]), om2, "memory after two calls")


@pytest.mark.parametrize("kind,n", [("sae", 1_000_000), ("sae", 20_000), ("eci", 100_000)])
def test_workspace_of_exactly_the_queried_size_gets_the_two_launch_form(er, kind, n):
    """A C caller allocates exactly frlw_encoder_workspace_bytes() -- the only size query include/frlw_evd.h documents for
    frlw_sae_encode / frlw_eci_encode -- and must get the same two-launch form a larger workspace gets (the
    chunk-major tables are larger than the general plan's, and the query used to return the general plan's size)."""
    import ctypes as C
    from frlw_evd_amd import _lib
    lib = _lib.load()
    H, W = 240, 304
    ev = synth.synth_events(7400 + n % 97, n, W, H, 5_000_000, t_offset=30_000_000)
    dat = dat_dev(ev)
    need = lib.frlw_encoder_workspace_bytes(n, H, W)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")  # exactly the queried size, no slack
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.frlw_workspace_init(C.c_void_p(ws.data_ptr()), ws.numel(), st), "init")
    desc = _lib.FrlwEvents(dat.data_ptr(), n, _lib.LAYOUT_DAT8, 0, None, None, 0, 0, None)
    status = C.c_int(0)
    c0 = path_counts()
    if kind == "sae":
        lam = (C.c_double * len(LAMDAS))(*LAMDAS)
        out = torch.empty((2 * len(LAMDAS), H, W), device="cuda")
        mem = torch.empty((2, H, W), device="cuda")
        _lib.check(lib.frlw_sae_encode(C.byref(desc), H, W, lam, len(LAMDAS), None, C.c_void_p(mem.data_ptr()), 35_000_000, 5_541_263,
                                       C.c_void_p(out.data_ptr()), None, C.c_void_p(ws.data_ptr()), ws.numel(), st), "sae")
        assert list(path_counts() - c0) == [1, 0, 0, 0]
        want, _u, wmem = er.encode_sae_dat(dat, (H, W), LAMDAS, None, 35_000_000, 5_541_263)
        assert torch.equal(mem, wmem)
    else:
        out = torch.empty((2, H, W), device="cuda")
        _lib.check(lib.frlw_eci_encode(C.byref(desc), H, W, C.c_void_p(out.data_ptr()), None, C.c_void_p(ws.data_ptr()), ws.numel(), st), "eci")
        assert list(path_counts() - c0) == [0, 0, 1, 0]
        want, _u = er.encode_eci_dat(dat, (H, W))
    _lib.check(lib.frlw_encoder_status(C.c_void_p(ws.data_ptr()), st, C.byref(status)), "status")
    assert status.value == 0
    assert torch.equal(out, want)
