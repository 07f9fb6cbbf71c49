"""DAT ingest against the reference's own reader: tests/golden/dat_io.npz holds what ``PSEELoader`` / ``parse_header``
and the label loop of ``generate_taf.py`` (lines 160-193, run from the reference's source) answered on synthetic files
that both sides regenerate from a seed (tests/golden/make_golden_dat.py).  GPU: a label's record slice through the fused
TAF encode equals the oracle."""
import os

import numpy as np
import pytest

from frlw_evd_amd import dat_io, synth

HERE = os.path.dirname(os.path.abspath(__file__))
sys_path_golden = os.path.join(HERE, "golden")

FILES = {  # the recipe of tests/golden/make_golden_dat.py
    "dense": (4101, 400_000, 304, 240, 20_000, 0),
    "long": (4102, 350_000, 304, 240, 3_000_000, 5_000),
    "small": (4103, 30_000, 64, 48, 400_000, 0),
}


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(sys_path_golden, "dat_io.npz"))


def make_file(name, folder):
    seed, n, W, H, span, t0 = FILES[name]
    ev = synth.synth_events(seed, n, W, H, span, t_offset=t0)
    path = os.path.join(str(folder), f"{name}_td.dat")
    dat_io.write_dat(path, synth.to_dat8(ev), H, W)
    return path, ev


@pytest.mark.parametrize("name", sorted(FILES))
def test_header_seek_load_like_pseeloader(tmp_path, golden, name):
    path, ev = make_file(name, tmp_path)
    bod, ev_type, ev_size, size = dat_io.parse_header(path)
    assert [bod, ev_type, ev_size, size[0], size[1]] == list(golden[f"{name}_header"])
    f = dat_io.DatFile(path)
    assert [f.event_count(), f.total_time()] == list(golden[f"{name}_count_total"])
    back = synth.from_dat8(np.asarray(f.records))
    for k in "xypt":
        assert np.array_equal(back[k], ev[k])
    # seek_time: index (or None), current_time, done and cursor, incl. the exact-hit early return among tied stamps
    early = 0
    for t, (want, cur, done, pos) in zip(golden[f"{name}_probes"], golden[f"{name}_seek_time"]):
        f.reset()
        got = f.seek_time(int(t))
        assert (-1 if got is None else got, f.current_time, int(f.done), f.pos) == (want, cur, done, pos), int(t)
        if want >= 0 and want != np.searchsorted(ev["t"], t, side="left"):
            early += 1
    if name == "dense":
        assert early >= 3, "the fixture must exercise the early return on an exact hit (psee_loader.py:207-218)"
    # seek_event + load_n_events
    for k, cnt, ct0, done0, n_loaded, ct1, done1, pos, chk in golden[f"{name}_ops"]:
        f.seek_event(int(k))
        assert (f.current_time, int(f.done)) == (ct0, done0), int(k)
        recs = f.load_n_events(int(cnt))
        e = synth.from_dat8(np.asarray(recs))
        mine = int(e["t"].sum() + 3 * e["x"].sum() + 5 * e["y"].sum() + 7 * e["p"].sum()) if len(recs) else 0
        if n_loaded == 0 and ct1 != f.current_time:
            # nothing left to load: the reference reads current_time from an uninitialised buffer element (np.empty,
            # psee_loader.py:99-106); only cursor and done are defined
            assert (len(recs), int(f.done), f.pos) == (0, done1, pos)
            continue
        assert (len(recs), f.current_time, int(f.done), f.pos, mine) == (n_loaded, ct1, done1, pos, chk), (int(k), int(cnt))


@pytest.mark.parametrize("key,name,min_count", [("long_slices", "long", 50000000), ("long_slices_min1000", "long", 1000),
                                                ("dense_slices", "dense", 50000000)])
def test_label_slices_like_generate_taf(tmp_path, golden, key, name, min_count):
    path, _ = make_file(name, tmp_path)
    labels = golden[f"{name}_labels"]
    got = list(dat_io.taf_label_slices(dat_io.DatFile(path), labels, 10000, 8, min_count))
    want = golden[key]
    assert len(got) == len(want) < len(labels), "labels behind the last event are skipped (generate_taf.py:163-164)"
    for g, w in zip(got, want):
        assert [g["label_time"], g["start_count"], g["end_count"], g["start_time"], g["end_time"],
                g["end_count"] - g["start_count"], int(g["fresh"])] == list(w)


@pytest.mark.gpu
def test_file_labels_to_fused_taf_encode(tmp_path):
    """generate_taf.py:160-235 for the labels of one file: slices from ``taf_label_slices``, records straight from the
    memory map to the GPU, windows in groups of at most 64 per launch, FIFO state carried between contiguous labels --
    equals the oracle run slice by slice."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import event_representation as er
    from oracle import oracle as orc
    path, _ = make_file("long", tmp_path)
    f = dat_io.DatFile(path)
    H, W = f.size
    labels = [120_000, 370_000, 620_000, 2_400_000, 2_650_000, 3_100_000]
    st = ost = None
    n_done = 0
    for sl in dat_io.taf_label_slices(f, labels, 10000, 8):
        if sl["fresh"]:
            st = torch.full((H, W, 2, 8), -6000.0, device="cuda")
            ost = np.full((H, W, 2, 8), -6000, np.float32)
        dat = f.to_device(sl["start_count"], sl["end_count"] - sl["start_count"])
        u8 = er.encode_taf_label(dat, (H, W), st, sl["start_time"], 10_000, sl["bins"], 8)
        rec = np.asarray(f.records[sl["start_count"]:sl["end_count"]])
        view, ost = orc.taf_stream_dat8(rec, (H, W), (H, W), 8, sl["start_time"], 10_000, sl["bins"], ost)
        assert st.cpu().numpy().tobytes() == ost.tobytes(), sl
        ou8 = orc.quantize_u8(np.ascontiguousarray(orc.leaky_transform(view.reshape(8, 2, H, W))[::-1]))
        d = np.abs(u8.cpu().numpy().astype(np.int16) - ou8.astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() <= 1e-4
        n_done += 1
    assert n_done == 5
