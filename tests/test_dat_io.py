"""DAT ingest: header parsing and seek on CPU; file -> GPU -> fused TAF encode equals the oracle on GPU."""
import numpy as np
import pytest

from frlw_evd_amd import dat_io, synth


def _file(tmp_path, n=50_000, H=240, W=304):
    ev = synth.synth_events(9, n, W, H, 2_000_000, t_offset=1_000)
    path = str(tmp_path / "seq_td.dat")
    dat_io.write_dat(path, synth.to_dat8(ev), H, W)
    return path, ev


def test_header_and_seek(tmp_path):
    path, ev = _file(tmp_path)
    start, ev_type, ev_size, size = dat_io.parse_header(path)
    assert (ev_type, ev_size, size) == (0, 8, (240, 304)) and start > 0
    f = dat_io.DatFile(path)
    assert len(f) == 50_000 and f.total_time() == int(ev["t"][-1])
    back = synth.from_dat8(np.asarray(f.records))
    for k in "xypt":
        assert np.array_equal(back[k], ev[k])
    for t in (0, 1_000, 777_777, int(ev["t"][-1]), int(ev["t"][-1]) + 5):
        want = len(ev["t"]) if t > ev["t"][-1] else int(np.searchsorted(ev["t"], t, side="left"))
        assert f.seek_time(t) == want


@pytest.mark.gpu
def test_file_to_fused_taf_encode(tmp_path):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from frlw_evd_amd import event_representation as er
    from oracle import oracle as orc
    path, ev = _file(tmp_path, n=300_000)
    f = dat_io.DatFile(path)
    H, W = f.size
    end_time = 1_500_000  # a label time stamp; 8 windows of 10 ms before it (generate_taf.py:160-186)
    start_time = end_time - 80_000
    lo, hi = f.seek_time(start_time), f.seek_time(end_time)
    dat = f.to_device(lo, hi - lo)
    st = torch.full((H, W, 2, 8), -6000.0, device="cuda")
    er.encode_taf_dat(dat, (H, W), st, start_time, 10_000, 8, 8)
    rec = np.asarray(f.records[lo:hi])
    _, ost = orc.taf_stream_dat8(rec, (H, W), (H, W), 8, start_time, 10_000, 8, np.full((H, W, 2, 8), -6000, np.float32))
    assert st.cpu().numpy().tobytes() == ost.tobytes()
