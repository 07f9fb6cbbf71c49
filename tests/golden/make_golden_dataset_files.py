#!/usr/bin/env python3
"""Golden vectors of the disk-backed datasets, produced by RUNNING the reference's ``propheseeDataset`` /
``propheseeTafDataset`` (data/dataset.py:23-308) and ``collate_events`` (data/loader.py:34-46) on the fabricated
directory of tests/dataset_fixture.py: the sample lists (which annotated timestamps have a representation file),
``load_data`` of every sample -- including the HEAD line that stacks the channel mean twice (data/dataset.py:245) --
``__getitem__`` without augmentation (image, labels, name, timestamp) and one collated batch.

    python tests/golden/make_golden_dataset_files.py     # rewrites tests/golden/dataset_files.npz; build container only
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FRLW_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)
for name in ("h5py", "cv2"):
    sys.modules.setdefault(name, types.ModuleType(name))

import dataset_fixture as fx  # noqa: E402
from data.dataset import propheseeDataset, propheseeTafDataset  # noqa: E402
from data.loader import collate_events  # noqa: E402


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        bbox, ev, taf = fx.build(tmp)
        for mode in ("train", "val", "test"):
            d = propheseeDataset(bbox, ev, "gen1", fx.IMG, fx.IMG, fx.BINS, 10000, 1, mode, False, False)
            order = np.argsort([f"{n}_{int(t):012d}" for n, t in zip(d.file_name, d.sequence_end_t)])  # os.listdir order is arbitrary
            out[f"ev_{mode}_names"] = np.array([d.file_name[i] for i in order])
            out[f"ev_{mode}_times"] = np.array([int(d.sequence_end_t[i]) for i in order], dtype=np.int64)
            out[f"ev_{mode}_load_data"] = np.stack([d.load_data(int(i)) for i in order])  # HEAD: (2, H, W) channel mean, twice
            items = [d[int(i)] for i in order]
            out[f"ev_{mode}_img"] = np.stack([np.ascontiguousarray(it[0]) for it in items])
            out[f"ev_{mode}_labels"] = np.stack([it[1] for it in items])
            if mode == "val":
                b = collate_events(items)
                out["ev_val_batch_img"], out["ev_val_batch_labels"] = b[0].numpy(), b[1].numpy()
                out["ev_val_batch_names"], out["ev_val_batch_times"] = np.array(b[2]), np.asarray(b[3]).astype(np.int64)
            for K in (8, 4):
                t = propheseeTafDataset(bbox, taf, "gen1", fx.IMG, fx.IMG, 10000, K, mode, False, False)
                order = np.argsort([f"{n}_{int(ts):012d}" for n, ts in zip(t.file_name, t.sequence_end_t)])
                out[f"taf{K}_{mode}_names"] = np.array([t.file_name[i] for i in order])
                out[f"taf{K}_{mode}_times"] = np.array([int(t.sequence_end_t[i]) for i in order], dtype=np.int64)
                out[f"taf{K}_{mode}_load_data"] = np.stack([t.load_data(int(i)) for i in order])
                items = [t[int(i)] for i in order]
                out[f"taf{K}_{mode}_img"] = np.stack([np.ascontiguousarray(it[0]) for it in items])
                out[f"taf{K}_{mode}_labels"] = np.stack([it[1] for it in items])
    path = os.path.join(HERE, "dataset_files.npz")
    np.savez_compressed(path, **out)
    print("dataset_files.npz", os.path.getsize(path) // 1024, "KiB")
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
