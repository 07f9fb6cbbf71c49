"""The batched Event Volume path (csrc/taf_fast.hip, ``frlw_ev_encode_batch``) against the CPU oracle (pinned to the
reference's goldens, tests/test_oracle_golden.py), the reference-generated GEN1 goldens and the general path
(``frlw_ev_encode``), bit for bit.  Harness semantics: generate_eventvolume.py:139-157 per label window."""
import hashlib
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from frlw_evd_amd import synth  # noqa: E402
from golden_util import assert_bitexact  # noqa: E402

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


@pytest.mark.parametrize("tag,hot", [("", False), ("hot_", True)])
def test_gen1_golden_single_sequence(er, golden_dir, tag, hot):
    """BASELINE.json configs[1] (1 M events, 304x240, 5 bins) through the batch entry point with one sequence: the sha256
    of the reference's own output."""
    g = np.load(os.path.join(golden_dir, "gen1_ev.npz"))
    H, W = 240, 304
    rec = synth.to_dat8(synth.synth_events(1002, 1_000_000, W, H, 250_000, hotspot=hot))
    out, _ = er.encode_ev_batch(to_dev(rec), [0, len(rec)], (H, W), 250_000, 250_000, 5)
    assert hashlib.sha256(host(out[0]).tobytes()).hexdigest() == str(g[tag + "native_sha"])


@pytest.mark.parametrize("tile_walk", [0, 1])
@pytest.mark.parametrize("bins", [1, 2, 5, 8])
def test_batch_vs_oracle_and_general_path(er, orc, bins, tile_walk, monkeypatch):
    """Eight label windows in one call (320 (sequence, tile) pairs): own t_end each, a sparse one, an empty one, one with a
    hot spot, one unsorted, one with events in front of its window (dropped like the harness' time filter) and events exactly
    on t_end.  Both second-level forms: the split pass + kf_ev_sub (default) and the opt-in tile walk (kf_ev_tile)."""
    from frlw_evd_amd import _lib
    if tile_walk:
        monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(taf_tile_walk=1))
    H, W, B, win = 240, 304, 8, 50_000
    t_end = [50_000, 50_000, 1_050_000, 50_000, 50_000, 80_000, 50_000, 50_000]
    recs = []
    for j in range(B):
        n = 200_000 if j != 1 else 3_000
        ev = synth.synth_events(700 + j, n, W, H, win, hotspot=(j == 3), t_offset=t_end[j] - win + 1)
        ev["t"][-2:] = t_end[j]                      # exactly the end of the window: t* = bins
        if j == 5:
            ev["t"][:5000] -= 30_000                 # in front of the window: dropped (generate_eventvolume.py:139)
        keep = np.ones(n, bool) if j != 2 else np.zeros(n, bool)
        r = synth.to_dat8({k: v[keep] for k, v in ev.items()})
        if j == 6:
            r = r[np.random.default_rng(2).permutation(len(r))]
        recs.append(r)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = to_dev(np.concatenate(recs))
    out, u8 = er.encode_ev_batch(dat, offs, (H, W), t_end, win, bins, want_u8=True)
    assert out.shape == (B, 2 * bins, H, W)
    for j in range(B):
        want = orc.ev_stream_dat8(recs[j], (H, W), (H, W), bins, t_end[j], win)
        assert_bitexact(host(out[j]), want, f"sequence {j}")
        if len(recs[j]) and not tile_walk:
            oj, uj = er.encode_ev_dat(to_dev(recs[j]), (H, W), t_end[j], win, bins, want_u8=True)
            assert torch.equal(oj, out[j]) and torch.equal(uj, u8[j]), f"general path, sequence {j}"
    assert float(out[2].abs().sum()) == 0.0          # the empty sequence: an all-zero volume
    assert float(out[0].max()) > 0


def test_few_pairs_and_downscale_maps(er, orc):
    """A single 1 Mpx stream down-scaled to 512 x 640 by the coordinate maps (generate_eventvolume.py:143-146): 90 pairs,
    so every tile goes through the segment split + kf_ev_sub."""
    Hs, Ws, H, W, win = 720, 1280, 512, 640, 80_000
    xmap, ymap = er.coordinate_maps((Hs, Ws), (H, W), "cuda")
    rec = synth.to_dat8(synth.synth_events(1013, 600_000, Ws, Hs, win, hotspot=True, t_offset=1))
    out, _ = er.encode_ev_batch(to_dev(rec), [0, len(rec)], (H, W), win, win, 5, xmap=xmap, ymap=ymap)
    assert_bitexact(host(out[0]), orc.ev_stream_dat8(rec, (Hs, Ws), (H, W), 5, win, win), "down-scaled volume")


def test_event_behind_the_window_is_outside_the_contract(er):
    H, W, win = 64, 96, 10_000
    ev = synth.synth_events(5, 20_000, W, H, win, t_offset=1)
    ev["t"][-1] = win + 7   # behind t_end
    dat = to_dev(synth.to_dat8(ev))
    with pytest.raises(ValueError):
        er.encode_ev_batch(dat, [0, len(ev["t"])], (H, W), win, win, 5)
    er.encode_ev_batch(dat, [0, len(ev["t"])], (H, W), win, win, 5, check=False)
    with pytest.raises(ValueError):
        er.raise_deferred()
    with pytest.raises(NotImplementedError):  # window beyond the 20 bits of the 4-byte record
        er.encode_ev_batch(dat, [0, len(ev["t"])], (H, W), 2_000_000, 2_000_000, 5)


def test_label_inside_the_first_window_of_a_file(er, orc):
    """t_end < window: the window starts before time 0 (t0 negative), which the 32-bit SIMPLE decode of kf_hist / kf_scatter does
    not cover -- the call takes the general instantiation.  Mixed with an ordinary sequence in one batch."""
    H, W, win = 120, 160, 50_000
    t_end = [20_000, 60_000]
    recs = [synth.to_dat8(synth.synth_events(40 + j, 80_000, W, H, t_end[j] - 1, t_offset=1)) for j in range(2)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    out, _ = er.encode_ev_batch(to_dev(np.concatenate(recs)), offs, (H, W), t_end, win, 5)
    for j in range(2):
        assert_bitexact(host(out[j]), orc.ev_stream_dat8(recs[j], (H, W), (H, W), 5, t_end[j], win), f"sequence {j}")


@pytest.mark.parametrize("n_seq", [1, 40])
def test_direct_bins_equal_tile_bins(er, n_seq, monkeypatch):
    """frlw_tuning_t::direct_bins 1 against 0 for the Event Volume batch (see tests/test_taf_fast_gpu.py)."""
    from frlw_evd_amd import _lib
    H, W, win = 240, 304, 50_000
    recs = [synth.to_dat8(synth.synth_events(1300 + j, 0 if j == 3 else 40_000 + 500 * j, W, H, win - 1, hotspot=(j % 5 == 1), t_offset=1))
            for j in range(n_seq)]
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = to_dev(np.concatenate(recs))
    outs = []
    for direct, cmaj in ((1, 1), (1, 0), (0, 1), (0, 0)):  # x the chunk-major partition (the sub-tile wavefronts gather their own lists)
        monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(direct_bins=direct, chunk_major=cmaj))
        outs.append(er.encode_ev_batch(dat, offs, (H, W), win, win, 5, want_u8=True))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])


@pytest.mark.parametrize("bins", [1, 2, 5, 8])
@pytest.mark.parametrize("n_seq", [1, 3])
def test_lds_float_atomic_kernel_equals_ticket_kernel_and_oracle(er, orc, bins, n_seq, monkeypatch):
    """kf_ev_fadd (small direct-mode calls: the sums are made by ds_add_f32 in stream order, csrc/taf_fast.hip) against the
    ticket-sort kernel (``ev_lds_float_atomics`` 1 / 0) and the oracle, bit for bit: a time-sorted stream (one run of equal
    floor(t*) per instruction except at the slice boundaries), a SHUFFLED one (runs of one record: the upper weight of an earlier
    record and the lower weight of a later one meet in one bin -- the order the reference's sequential index_add_ defines,
    generate_eventvolume.py:28-32), a hot spot (hundreds of records per cell), events exactly on the bin centres and on t_end."""
    from frlw_evd_amd import _lib
    H, W, win = 240, 304, 60_000
    recs = []
    for j in range(n_seq):
        ev = synth.synth_events(8800 + 10 * bins + j, 260_000, W, H, win, hotspot=(j == 0), t_offset=1)
        if j == 0:  # events exactly on the bin centres (t* integer) and on the window's end
            ev["t"][::97] = (np.arange(len(ev["t"][::97])) % bins + 1) * (win // bins)
            ev["t"][-5:] = win
            ev["t"].sort()
        if j == 1:  # not time-sorted at all
            perm = np.random.default_rng(5).permutation(len(ev["t"]))
            ev = {k: v[perm] for k, v in ev.items()}
        recs.append(synth.to_dat8(ev))
    offs = np.concatenate([[0], np.cumsum([len(r) for r in recs])])
    dat = to_dev(np.concatenate(recs))
    outs = []
    for knob in (1, 0):
        monkeypatch.setattr(er, "TUNING", _lib.FrlwTuning(ev_lds_float_atomics=knob, direct_bins=1))
        out, u8 = er.encode_ev_batch(dat, offs, (H, W), win, win, bins, want_u8=True)
        outs.append((out, u8))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for j, r in enumerate(recs):
        assert_bitexact(host(outs[0][0][j]), orc.ev_stream_dat8(r, (H, W), (H, W), bins, win, win), f"sequence {j}")


def test_single_window_shim_takes_the_two_launch_path(er, orc):
    """``encode_ev_dat(fast=True)`` = the batch entry point with one window; an event behind t_end makes the checked call fall back
    to the general path, which places it like the reference (generate_eventvolume.py:139 filters only the front)."""
    H, W, win = 240, 304, 100_000
    rec = synth.to_dat8(synth.synth_events(8900, 300_000, W, H, win, t_offset=1))
    want = orc.ev_stream_dat8(rec, (H, W), (H, W), 5, win, win)
    for fast in (True, False, "auto"):
        out, _ = er.encode_ev_dat(to_dev(rec), (H, W), win, win, 5, fast=fast)
        assert_bitexact(host(out), want, f"fast={fast}")
    late = rec.copy()
    late["t"][-2:] += 7  # behind t_end: outside the batch path's contract
    out, _ = er.encode_ev_dat(to_dev(late), (H, W), win, win, 5, fast=True)
    assert_bitexact(host(out), orc.ev_stream_dat8(late, (H, W), (H, W), 5, win, win), "late events through the fall-back")
