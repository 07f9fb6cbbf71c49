"""Evaluator hand-off against goldens produced by the reference's own
``evaluate.evaluator`` classes (tests/golden/make_golden_evaluator.py)."""
import os

import numpy as np
import pytest
import torch

from frlw_evd_amd.evaluator import evaluator, recorder


def batch(seed, B=4):  # the same recipe as the golden script
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
        g = int(rng.integers(0, 4)) if b != 1 else 0
        for k in range(g):
            tg[b, k] = [rng.uniform(20, 300), rng.uniform(20, 230), rng.uniform(4, 80), rng.uniform(4, 80),
                        rng.integers(0, 2), t, 1.0, k]
    return outs, torch.from_numpy(tg), ts, [f"seq{seed}_{b}" for b in range(B)]


CONFIGS = [("gen1", (304, 240), (320, 256)), ("gen4", (1280, 720), (640, 512))]


def run(dataset, ori, inp, tmp, device):
    rec = recorder(str(tmp))
    ev = evaluator(["car", "ped"], 4, 10000, ori[0], ori[1], inp[0], inp[1], dataset=dataset, recorder=rec)
    for seed in (1, 2, 3):
        outs, tg, ts, names = batch(seed)
        ev.add_result([o.to(device) for o in outs], ts, tg, names, 0.01, 0.0)
    return ev, rec


def check(ev, rec, g, dataset, tmp):
    assert len(ev.dt_to_eval) == int(g[f"{dataset}_n"]) == len(ev.gt_to_eval)
    for i, (gt, dt) in enumerate(zip(ev.gt_to_eval, ev.dt_to_eval)):
        assert gt.dtype == g[f"{dataset}_gt_{i}"].dtype and np.array_equal(gt, g[f"{dataset}_gt_{i}"])
        assert dt.dtype == g[f"{dataset}_dt_{i}"].dtype and np.array_equal(dt, g[f"{dataset}_dt_{i}"]), (i, dt, g[f"{dataset}_dt_{i}"])
        assert np.array_equal(ev.filter_boxes(gt), g[f"{dataset}_gtf_{i}"])
        assert np.array_equal(ev.filter_boxes(dt), g[f"{dataset}_dtf_{i}"])
    assert ev.tol == int(g[f"{dataset}_tol"])
    assert [ev.infer_time, ev.infer_count] == pytest.approx(list(g[f"{dataset}_times"]))
    res = ev.evaluate()  # pycocotools is absent: the filtered, paired lists come back
    z = np.load(os.path.join(str(tmp), "summarise.npz"))
    assert list(z["file_names"]) == list(g[f"{dataset}_rec_names"])
    assert np.array_equal(z["dts"], g[f"{dataset}_rec_dts"])
    if isinstance(res, dict):
        assert len(res["gt_boxes_list"]) == len(res["dt_boxes_list"]) > 0
        assert all(len(x) > 0 for x in res["gt_boxes_list"])


@pytest.mark.parametrize("dataset,ori,inp", CONFIGS)
def test_host_path_matches_reference(golden_dir, tmp_path, dataset, ori, inp):
    g = np.load(os.path.join(golden_dir, "evaluator.npz"))
    ev, rec = run(dataset, ori, inp, tmp_path, "cpu")
    check(ev, rec, g, dataset, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("dataset,ori,inp", CONFIGS)
def test_device_path_matches_reference(golden_dir, tmp_path, dataset, ori, inp):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = np.load(os.path.join(golden_dir, "evaluator.npz"))
    ev, rec = run(dataset, ori, inp, tmp_path, "cuda")
    check(ev, rec, g, dataset, tmp_path)
    # the kernel's filter mask equals the reference's filter on the same rows
    outs, tg, ts, names = batch(5, B=6)
    dts, keeps = ev.transform_dt_batch([o.cuda() for o in outs], ts)
    for d, k in zip(dts, keeps):
        assert np.array_equal(d[k], ev.filter_boxes(d))
