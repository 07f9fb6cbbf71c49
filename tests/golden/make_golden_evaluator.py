#!/usr/bin/env python3
"""Golden vectors of the evaluator hand-off, produced by the REFERENCE's own ``evaluate.evaluator.evaluator``
(evaluate/evaluator.py:9-115: transform_gt / transform_dt / add_result, the Prophesee min-size filters of
evaluate/src/io/box_filtering.py:17-47) and ``recorder`` (:117-133).  COCO mAP itself (pycocotools) is not run.

    python tests/golden/make_golden_evaluator.py     # rewrites tests/golden/evaluator.npz
"""
import os
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


def stub(name, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


stub("torchvision")
stub("pycocotools")
stub("pycocotools.coco", COCO=object)
stub("pycocotools.cocoeval", COCOeval=object)

from evaluate.evaluator import evaluator as RefEvaluator, recorder as RefRecorder  # noqa: E402


def batch(seed, B=4):
    """Detections (list of (n_i, 6) f32 [cx, cy, w, h, cls, score]), targets (B, 80, 8) f64
    [cx, cy, w, h, class, t, confidence, track] (val mode, data/dataset.py:205-209), timestamps, file names."""
    rng = np.random.default_rng(seed)
    outs, ts = [], []
    tg = np.zeros((B, 80, 8))
    for b in range(B):
        n = int(rng.integers(0, 7))
        d = np.zeros((max(n, 1), 6), np.float32)
        if n:
            d[:, 0] = rng.uniform(0, 320, n); d[:, 1] = rng.uniform(0, 256, n)
            d[:, 2] = rng.uniform(2, 90, n); d[:, 3] = rng.uniform(2, 90, n)
            d[:, 4] = rng.integers(0, 2, n); d[:, 5] = rng.uniform(0.05, 1, n)
        outs.append(torch.from_numpy(d))
        t = int(rng.integers(1, 40)) * 100_000 + (0 if b else 300_000)
        ts.append(t)
        g = int(rng.integers(0, 4)) if b != 1 else 0   # image 1 has no ground truth: skipped by add_result
        for k in range(g):
            tg[b, k] = [rng.uniform(20, 300), rng.uniform(20, 230), rng.uniform(4, 80), rng.uniform(4, 80),
                        rng.integers(0, 2), t, 1.0, k]
    return outs, torch.from_numpy(tg), ts, [f"seq{seed}_{b}" for b in range(B)]


def main():
    out = {}
    for dataset, ori, inp in (("gen1", (304, 240), (320, 256)), ("gen4", (1280, 720), (640, 512))):
        with tempfile.TemporaryDirectory() as tmp:
            rec = RefRecorder(tmp)
            ev = RefEvaluator(["car", "ped"], 4, 10000, ori[0], ori[1], inp[0], inp[1], dataset=dataset, recorder=rec)
            for seed in (1, 2, 3):
                outs, tg, ts, names = batch(seed)
                ev.add_result(outs, ts, tg, names, 0.01, 0.0)
            out[f"{dataset}_n"] = np.array(len(ev.dt_to_eval))
            for i, (g, d) in enumerate(zip(ev.gt_to_eval, ev.dt_to_eval)):
                out[f"{dataset}_gt_{i}"] = g
                out[f"{dataset}_dt_{i}"] = d
                out[f"{dataset}_gtf_{i}"] = ev.filter_boxes(g)
                out[f"{dataset}_dtf_{i}"] = ev.filter_boxes(d)
            rec.save()
            z = np.load(os.path.join(tmp, "summarise.npz"))
            out[f"{dataset}_rec_names"] = z["file_names"]
            out[f"{dataset}_rec_dts"] = z["dts"]
            out[f"{dataset}_tol"] = np.array(ev.tol)
            out[f"{dataset}_times"] = np.array([ev.infer_time, ev.infer_count])
    np.savez_compressed(os.path.join(HERE, "evaluator.npz"), **out)
    print("evaluator.npz", os.path.getsize(os.path.join(HERE, "evaluator.npz")) // 1024, "KiB")
    print({k: (v.dtype, v.shape) for k, v in out.items() if k.startswith("gen1_") and k.endswith("_0")})


if __name__ == "__main__":
    main()
