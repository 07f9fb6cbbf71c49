"""The build's own detector definition (frlw_evd_amd.yolox, plain PyTorch) against golden vectors produced
by the REFERENCE's modules (tests/golden/make_golden_detector.py).  CPU only.  This pins the torch fp32
reference that the HIP engine is compared with on the GPU (tests/test_detector_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from frlw_evd_amd.yolox import build_yolox
from frlw_evd_amd.yolox.model import recipe_state_dict
from frlw_evd_amd.yolox.yolo_head import nms_reference


def detector_input(seed, B, C=10, H=256, W=320):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.integers(0, 256, size=(B, C, H, W, 1, 1)).astype(np.float32) / np.float32(255))


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "detector.npz"))


@pytest.mark.parametrize("tag,C", [("ev10", 10), ("taf16", 16)])
def test_eager_matches_reference(golden, tag, C):
    torch.set_num_threads(8)
    m = build_yolox(C, 2)
    assert sum(p.numel() for p in m.parameters()) == int(golden[f"{tag}_params"])
    m.load_state_dict(recipe_state_dict(m, seed=1004))
    m.eval()
    x = detector_input(1004, 2, C)
    with torch.no_grad():
        raw = m.reference_outputs(x[..., 0])
    want = golden[f"{tag}_raw"]
    # same ops on the same weights: only the conv summation order of the CPU backend may differ
    assert np.abs(raw.numpy() - want).max() <= 1e-5 * np.abs(want).max()
    dec = m.head.decode_boxes(raw)
    assert np.abs(dec.numpy() - golden[f"{tag}_decoded"]).max() <= 1e-4


def test_param_names_and_count():
    m = build_yolox(10, 2)
    keys = list(m.state_dict().keys())
    assert "backbone.stem.conv.conv.weight" in keys and "backbone.stem.conv.bn.running_mean" in keys
    assert "head.cls_preds.0.bias" in keys and "neck.C3_p4.m.0.conv2.bn.weight" in keys
    assert sum(p.numel() for p in m.parameters()) == 14_375_765


def test_eval_cpu_forward_list_of_dets():
    m = build_yolox(10, 2)
    m.load_state_dict(recipe_state_dict(m))
    m.eval()
    x = detector_input(3, 1)
    with torch.no_grad():
        out = m(x)  # (B, C, H, W, 1, T=1) like data/dataset.py:251-252
    assert isinstance(out, list) and out[0].shape[1] == 6


def test_nms_reference_semantics():
    boxes = torch.tensor([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    keep = nms_reference(boxes, scores, 0.6)
    assert keep.tolist() == [3, 2]  # 0 and 1 overlap box 3 with IoU > 0.6
    keep = nms_reference(boxes, scores, 0.99)
    assert keep.tolist() == [3, 1, 2]  # only the exact duplicate (IoU = 1 > 0.99) goes


def test_training_branch_is_loud():
    m = build_yolox(10, 2).train()
    with pytest.raises(NotImplementedError):
        m(detector_input(1, 1), torch.zeros(1, 80, 5))
