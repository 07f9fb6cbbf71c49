#!/usr/bin/env python3
"""Golden vectors of the dataset-side sample transform, produced by the REFERENCE's own
``propheseeDataset.__getitem__`` (data/dataset.py:116-239) run on fabricated files: ``/255``, nearest zoom-in by
sr in [1, 1.5], crop, horizontal flip and the label transform, with Python's ``random`` seeded per sample.

Runs only in the build container.  Inputs (uint8 volume, boxes) are regenerated from their seeds by the tests.

    python tests/golden/make_golden_dataset.py     # rewrites tests/golden/dataset.npz
"""
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FRLW_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

for name in ("h5py", "cv2"):
    sys.modules.setdefault(name, types.ModuleType(name))

from data.dataset import propheseeDataset  # noqa: E402

BBOX_DTYPE = np.dtype([("t", "<u8"), ("x", "<f4"), ("y", "<f4"), ("w", "<f4"), ("h", "<f4"), ("class_id", "u1"),
                       ("class_confidence", "<f4"), ("track_id", "<u4")])
IN_SIZE = [64, 80]      # detector input (H, W) of the fixture
SENSOR = (240, 304)     # GEN1 sensor (H, W): box coordinates live here
C = 4


def sample_inputs(seed):
    """uint8 volume (C, H, W) and the boxes of one label time (sensor pixels), from PCG64(seed)."""
    rng = np.random.default_rng(seed)
    vol = rng.integers(0, 256, size=(C, IN_SIZE[0], IN_SIZE[1]), dtype=np.uint8)
    n = int(rng.integers(1, 6))
    b = np.zeros(n, dtype=BBOX_DTYPE)
    b["t"] = 1_000_000
    b["w"] = rng.uniform(10, 120, n).astype(np.float32)
    b["h"] = rng.uniform(10, 100, n).astype(np.float32)
    b["x"] = rng.uniform(-5, SENSOR[1] - 20, n).astype(np.float32)
    b["y"] = rng.uniform(-5, SENSOR[0] - 20, n).astype(np.float32)
    b["class_id"] = rng.integers(0, 2, n)
    b["class_confidence"] = 1.0
    b["track_id"] = np.arange(n)
    return vol, b


def main():
    out = {"seeds": [], "modes": []}
    with tempfile.TemporaryDirectory() as tmp:
        for k, (seed, mode, augment, clipping) in enumerate(
                [(s, "train", True, False) for s in range(100, 112)] + [(200, "train", True, True), (201, "val", False, False),
                                                                       (202, "train", False, False)]):
            vol, boxes = sample_inputs(seed)
            name = f"seq{k}"
            np.save(os.path.join(tmp, name + "_bbox.npy"), boxes)
            stub = types.SimpleNamespace(
                root=tmp, file_name=[name], sequence_end_t=[1_000_000], input_img_size=IN_SIZE, height=SENSOR[0],
                width=SENSOR[1], augment=augment, dataset="gen1", mode=mode, clipping=clipping,
                load_data=lambda idx, v=vol: v.astype(np.float32), after_process=lambda img: img[:, :, :, None, None])
            random.seed(seed)
            img, labels, fname, t = propheseeDataset.__getitem__(stub, 0)
            out["seeds"].append(seed)
            out["modes"].append([mode == "train", augment, clipping])
            out[f"img_{seed}"] = np.ascontiguousarray(img)
            out[f"labels_{seed}"] = labels
    out["seeds"] = np.array(out["seeds"])
    out["modes"] = np.array(out["modes"])
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("dataset.npz", os.path.getsize(os.path.join(HERE, "dataset.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
