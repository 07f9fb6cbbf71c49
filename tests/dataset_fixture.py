"""A fabricated representation dataset (what the ``generate_*.py`` commands would have written) + its annotation files, in
the layout ``propheseeDataset`` / ``propheseeTafDataset`` read (data/dataset.py:78-113,238-308): regenerated from seeds by
tests/golden/make_golden_dataset_files.py (which runs the REFERENCE's classes on it) and by the tests (which run the product's)."""
import os

import numpy as np

IMG = [32, 40]          # (H, W) of the stored volumes (a detector-input size: multiples of 32 are not needed for the loaders)
BINS = 5                # Event Volume bins -> 10 channels
K = 8                   # TAF K -> bins4 (newest four slots, 8 channels) + bins8 (the older four)
BBOX_DTYPE = np.dtype([("t", "<u8"), ("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"), ("class_id", "u1"),
                       ("class_confidence", "<f4"), ("track_id", "<u4")])
SEQS = {"train": ["s0", "s1"], "val": ["s2"], "test": ["s3"]}
TIMES = {"s0": [100_000, 350_000, 600_000], "s1": [200_000, 450_000], "s2": [150_000, 400_000], "s3": [120_000, 300_000, 900_000]}
MISSING = {("s0", 600_000), ("s3", 300_000)}   # annotated timestamps WITHOUT a representation file (the scan skips them)


def volume(seq, t, channels, tag):
    rng = np.random.default_rng(sum(map(ord, seq)) * 1_000_003 + int(t) * 31 + {"ev": 1, "b4": 2, "b8": 3}[tag])
    return rng.integers(0, 256, size=(channels, IMG[0], IMG[1]), dtype=np.uint8)


def boxes(seq):
    rng = np.random.default_rng(sum(map(ord, seq)))
    rows = []
    for t in TIMES[seq]:
        for k in range(int(rng.integers(1, 4))):
            rows.append((t, rng.uniform(20, 200), rng.uniform(20, 150), rng.uniform(20, 80), rng.uniform(20, 70), int(rng.integers(0, 2)), 1.0, k))
    return np.array(rows, dtype=BBOX_DTYPE)


def build(root):
    """-> (bbox_dir, event-volume data_dir, taf data_dir)"""
    bbox, ev, taf = (os.path.join(root, d) for d in ("bbox", "EventVolume250000", "taf"))
    for mode, seqs in SEQS.items():
        for d in (os.path.join(bbox, mode), os.path.join(ev, mode), os.path.join(taf, mode, "bins4"), os.path.join(taf, mode, "bins8")):
            os.makedirs(d, exist_ok=True)
        for seq in seqs:
            np.save(os.path.join(bbox, mode, seq + "_bbox.npy"), boxes(seq))
            for t in TIMES[seq]:
                if (seq, t) in MISSING:
                    continue
                volume(seq, t, 2 * BINS, "ev").tofile(os.path.join(ev, mode, f"{seq}_{t}.npy"))
                volume(seq, t, K, "b4").tofile(os.path.join(taf, mode, "bins4", f"{seq}_{t}.npy"))
                volume(seq, t, K, "b8").tofile(os.path.join(taf, mode, "bins8", f"{seq}_{t}.npy"))
    return bbox, ev, taf
