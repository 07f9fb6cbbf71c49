"""A fabricated mini-dataset in the layout the reference's ``generate_*.py`` harnesses walk
(``<raw_dir>/<mode>/<seq>_td.dat`` + ``<label_dir>/<mode>/<seq>_bbox.npy``, generate_taf.py:112-151): regenerated from seeds on
both sides (tests/golden/make_golden_harness.py runs the REFERENCE scripts on it, the tests run the product on it), so only
the answers are committed.

Sequences (GEN1 sensor, 304x240):
  test/seqA   900 000 events over 7.2 s.  Labels: a first one whose TAF window reaches back to the file start, contiguous
              ones, two labels 3 ms apart (the second rounds onto the first: TAF ``bins == 0``, generate_taf.py:181), a gap
              longer than the SAE window, one label behind the last event (skipped: ``seek_time`` returns None).
  train/seqB  150 000 events over 2 s with a hot pixel; two labels.
"""
import os

import numpy as np

from frlw_evd_amd import dat_io, synth

SENSOR = (240, 304)
SEQUENCES = {
    # (mode, name): (seed, n events, span us, t_offset, hotspot, label times)
    ("test", "seqA"): (5101, 900_000, 7_200_000, 2_000, False,
                       [350_000, 600_000, 603_000, 1_850_000, 5_400_000, 5_650_000, 7_100_000, 7_300_000]),
    ("train", "seqB"): (5102, 150_000, 2_000_000, 0, True, [500_000, 1_400_000]),
}
BBOX_DTYPE = np.dtype([("t", "<u8"), ("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"), ("class_id", "u1"),
                       ("class_confidence", "<f4"), ("track_id", "<u4")])


def records(mode, name):
    seed, n, span, t0, hot, _ = SEQUENCES[(mode, name)]
    H, W = SENSOR
    return synth.to_dat8(synth.synth_events(seed, n, W, H, span, hotspot=hot, t_offset=t0))


def label_times(mode, name):
    return np.asarray(SEQUENCES[(mode, name)][5], dtype=np.int64)


def build(root):
    """Write the dataset under ``root``; returns (raw_dir, label_dir)."""
    raw, lab = os.path.join(root, "raw"), os.path.join(root, "label")
    H, W = SENSOR
    for (mode, name) in SEQUENCES:
        os.makedirs(os.path.join(raw, mode), exist_ok=True)
        os.makedirs(os.path.join(lab, mode), exist_ok=True)
        dat_io.write_dat(os.path.join(raw, mode, name + "_td.dat"), records(mode, name), H, W)
        times = label_times(mode, name)
        boxes = np.zeros(2 * len(times), dtype=BBOX_DTYPE)   # two boxes per annotated timestamp
        boxes["t"] = np.repeat(times, 2)
        boxes["x"], boxes["y"], boxes["w"], boxes["h"] = 50, 60, 40, 30
        boxes["class_id"] = np.tile([0, 1], len(times))
        boxes["class_confidence"] = 1.0
        np.save(os.path.join(lab, mode, name + "_bbox.npy"), boxes)
    return raw, lab


def sample_positions(size, k=32768, seed=17):
    """Seeded byte positions of a file of ``size`` bytes (the golden keeps the reference's bytes there)."""
    return np.sort(np.random.default_rng(seed).choice(size, size=min(k, size), replace=False))
