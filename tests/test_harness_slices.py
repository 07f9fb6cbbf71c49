"""Label slicing of the four offline harnesses (``dat_io.*_label_slices`` + ``DatFile.load_delta_t``) against what the
reference's own scripts loaded: tests/golden/harness.npz holds, per script, the (file, first record, count) of every
``PSEELoader.load_n_events`` / ``load_delta_t`` call of a full run of generate_eventcountimage.py:130-160,
generate_eventvolume.py:118-137, generate_surfaceofactiveevents.py:147-176 and generate_taf.py:160-193 on the fabricated
dataset of tests/harness_data.py (make_golden_harness.py).  CPU only."""
import os

import numpy as np
import pytest

import harness_data
from frlw_evd_amd import dat_io

SLICERS = {"eci": dat_io.eci_label_slices, "ev": dat_io.ev_label_slices, "sae": dat_io.sae_label_slices,
           "taf": dat_io.taf_label_slices}


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    return harness_data.build(str(tmp_path_factory.mktemp("harness")))


@pytest.mark.parametrize("key", sorted(SLICERS))
def test_loads_equal_the_reference_scripts(dataset, golden_dir, key):
    raw, _ = dataset
    g = np.load(os.path.join(golden_dir, "harness.npz"))
    order = list(harness_data.SEQUENCES)
    got = []
    for fi in sorted(range(len(order)), key=lambda i: ("train", "val", "test").index(order[i][0])):  # the scripts' walk order
        mode, name = order[fi]
        f = dat_io.DatFile(os.path.join(raw, mode, name + "_td.dat"))
        for sl in SLICERS[key](f, harness_data.label_times(mode, name)):
            got.append([fi, *sl["loaded"]] if "loaded" in sl else [fi, sl["start_count"], sl["end_count"] - sl["start_count"]])
    assert got == g[key + "/loads"].tolist()


def test_eci_windows_are_suffixes_of_memory_plus_load(dataset):
    """The Event Count Image harness concatenates its ``memory`` (the last 200 000 events it has seen) with each load and
    takes the LAST N events (generate_eventcountimage.py:148-157): replay that literally on the host and compare with the
    contiguous ranges ``eci_label_slices`` hands to the encoder."""
    raw, _ = dataset
    windows = (50000, 100000, 200000)
    for (mode, name) in harness_data.SEQUENCES:
        f = dat_io.DatFile(os.path.join(raw, mode, name + "_td.dat"))
        idx = np.arange(len(f))
        memory = None
        for sl in dat_io.eci_label_slices(f, harness_data.label_times(mode, name), windows):
            lo, n = sl["loaded"]
            events = idx[lo:lo + n]
            if memory is not None:
                events = np.concatenate([memory, events])
            memory = events[-max(windows):]
            for N in windows:
                want = events[-N:]
                a = max(sl["tail_start"], sl["end_count"] - N)
                assert np.array_equal(want, idx[a:sl["end_count"]]), (name, sl["label_time"], N)


def test_load_delta_t_cursor_semantics(dataset):
    """psee_loader.py:117-159: events from the cursor with t < current_time + delta_t; current_time advances by delta_t (or
    to last + 1 at the end of the file); a second call continues; delta_t < 1 raises; at the end an empty slice."""
    raw, _ = dataset
    f = dat_io.DatFile(os.path.join(raw, "train", "seqB_td.dat"))
    t = f.records["t"].astype(np.int64)
    f.seek_time(400_000)
    lo = f.pos
    got = f.load_delta_t(250_000)
    assert f.current_time == 650_000 and not f.done
    assert lo + len(got) == int(np.searchsorted(t, 650_000, side="left")) and int(got["t"][-1]) < 650_000
    more = f.load_delta_t(10_000_000)   # runs off the end of the file
    assert f.done and f.current_time == int(t[-1]) + 1 and f.pos == len(t) and len(more) == len(t) - lo - len(got)
    assert len(f.load_delta_t(5)) == 0
    with pytest.raises(ValueError):
        f.load_delta_t(0)
