#!/usr/bin/env python3
"""Golden vectors of the YOLOX detector, produced by the REFERENCE's own modules
(core/yolox/models/*, core/model.py imported from /root/reference, torch-CPU fp32) with recipe weights.

Runs only in the build container.  Weights are not stored: frlw_evd_amd.yolox.model.recipe_state_dict
regenerates them from (seed, parameter name) on both sides; the input is regenerated from its seed.

    python tests/golden/make_golden_detector.py     # rewrites tests/golden/detector.npz
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FRLW_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# modules the reference imports but this image lacks
stub("turtle", forward=None)
stub("loguru", logger=types.SimpleNamespace(info=print, warning=print, error=print))
tv = stub("torchvision")
tv.ops = stub("torchvision.ops", nms=None, batched_nms=None)
stub("cv2")
stub("thop", profile=None)
timm = stub("timm")
timm.models = stub("timm.models")
timm.models.layers = stub("timm.models.layers", DropPath=torch.nn.Identity, trunc_normal_=lambda *a, **k: None)
torch.cuda.synchronize = lambda *a, **k: None
torch.set_num_threads(8)

from core.yolox.models.darknet import CSPDarknet  # noqa: E402
from core.yolox.models.network_blocks import Focus  # noqa: E402
from core.yolox.models.yolo_head import YOLOXHead  # noqa: E402
from core.yolox.models.yolo_pafpn import YOLOPAFPN  # noqa: E402
from core.model import model as RefModel  # noqa: E402

from frlw_evd_amd.yolox.model import build_yolox, recipe_state_dict  # noqa: E402


def detector_input(seed, B, C=10, H=256, W=320):
    """uint8-quantised U{0..255} / 255 as (B, C, H, W, 1, 1) f32 (SURVEY.md section 8d cfg 4)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.integers(0, 256, size=(B, C, H, W, 1, 1)).astype(np.float32) / np.float32(255)))


def train_labels():
    """(4, 80, 5) float64 [cls, cx, cy, w, h] in detector pixels, zero-padded (data/dataset.py:211-217)."""
    lab = torch.zeros((4, 80, 5), dtype=torch.float64)
    lab[0, 0] = torch.tensor([1, 100.0, 120.0, 40.0, 60.0])
    lab[0, 1] = torch.tensor([0, 200.0, 80.0, 30.0, 30.0])
    lab[1, 0] = torch.tensor([0, 160.0, 128.0, 80.0, 50.0])
    lab[2, 0] = torch.tensor([1, 30.5, 40.25, 21.0, 33.0])
    lab[2, 1] = torch.tensor([1, 36.0, 44.0, 25.0, 30.0])   # overlapping pair: anchors claimed twice
    lab[2, 2] = torch.tensor([0, 290.0, 230.0, 50.0, 40.0])
    return lab                                              # image 3 has no box


def main():
    out = {}
    for tag, C, nc in (("ev10", 10, 2), ("taf16", 16, 2)):
        chans = [128, 256, 512]
        ref = RefModel(CSPDarknet(C, 0.33, 0.5, stem=Focus),
                       YOLOPAFPN(0.33, in_features=["dark3", "dark4", "dark5"], in_channels=chans, act="silu"), None,
                       YOLOXHead(nc, in_channels=chans, act="silu", strides=[8, 16, 32], radius=5))
        mine = build_yolox(C, nc)
        sd = recipe_state_dict(mine, seed=1004)
        assert list(sd.keys()) == list(ref.state_dict().keys()), "parameter names differ from the reference"
        for (k, a), (_, b) in zip(sd.items(), ref.state_dict().items()):
            assert a.shape == b.shape, k
        ref.load_state_dict(sd)
        ref.eval()
        x = detector_input(1004, 2, C)
        with torch.no_grad():
            feats = ref.backbone(x[..., 0])
            fpn = ref.neck(feats)
            head = ref.head
            head.decode_in_inference = False
            raw = head(fpn)  # (B, 1680, 5 + nc) pre-decode
        out[f"{tag}_raw"] = raw.numpy()
        for name, t in zip(("dark3", "dark4", "dark5"), feats):
            out[f"{tag}_{name}_stats"] = np.array([t.mean().item(), t.norm().item(), t.abs().max().item()])
        for name, t in zip(("pan2", "pan1", "pan0"), fpn):
            out[f"{tag}_{name}_stats"] = np.array([t.mean().item(), t.norm().item(), t.abs().max().item()])
        out[f"{tag}_params"] = np.array(sum(p.numel() for p in ref.parameters()))
        # decoded (pre-NMS) boxes with the reference's own arithmetic (yolo_head.py:258-272)
        grids, strides = [], []
        for (h, w), s in zip(head.hw, head.strides):
            yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)])
            grids.append(torch.stack((xv, yv), 2).view(1, -1, 2))
            strides.append(torch.full((1, h * w, 1), s))
        grids = torch.cat(grids, 1).float()
        strides = torch.cat(strides, 1).float()
        dec = raw.clone()
        dec[..., :2] = (dec[..., :2] + grids) * strides
        dec[..., 2:4] = torch.square(dec[..., 2:4]) * strides
        out[f"{tag}_decoded"] = dec.numpy()
        if tag == "ev10":
            # train branch: SimOTA + losses + backward on a fixed label set (yolo_head.py:305-473)
            ref.load_state_dict(sd)
            ref.train()
            ref.head.decode_in_inference = True
            xt = detector_input(1005, 4, C)
            labels = train_labels()
            loss = ref(xt, labels, None, None)          # core/model.py:50-56 returns losses[0]
            out["train_loss"] = np.array(loss.item())
            ref.zero_grad()
            loss.backward()
            for grp in ("backbone", "neck", "head"):
                out[f"train_gradnorm_{grp}"] = np.array(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for n, p in ref.named_parameters() if n.startswith(grp) and p.grad is not None))))
            feats = ref.neck(ref.backbone(xt[..., 0, 0][..., None]))
            tup = ref.head(feats, labels, xt[..., 0])
            out["train_tuple"] = np.array([float(v) for v in tup])
    np.savez_compressed(os.path.join(HERE, "detector.npz"), **out)
    print("detector.npz", os.path.getsize(os.path.join(HERE, "detector.npz")) // 1024, "KiB")


def main_bfm():
    """The ``yolox_taf_bfm`` recipes (core/exp.py:588-591): the reference's Temporal_Active_Focus_connect stem."""
    from core.Others.Temporal_Active_Focus import Temporal_Active_Focus_connect
    out = {}
    for tag, C, nc in (("bfm8", 8, 2), ("bfm16", 16, 2)):
        chans = [128, 256, 512]
        ref = RefModel(CSPDarknet(C, 0.33, 0.5, stem=Temporal_Active_Focus_connect),
                       YOLOPAFPN(0.33, in_features=["dark3", "dark4", "dark5"], in_channels=chans, act="silu"), None,
                       YOLOXHead(nc, in_channels=chans, act="silu", strides=[8, 16, 32], radius=5))
        mine = build_yolox(C, nc, stem="bfm")
        sd = recipe_state_dict(mine, seed=1004)
        assert list(sd.keys()) == list(ref.state_dict().keys()), "parameter names differ from the reference"
        for (k, a), (_, b) in zip(sd.items(), ref.state_dict().items()):
            assert a.shape == b.shape, k
        ref.load_state_dict(sd)
        ref.eval()
        x = detector_input(1006, 2, C)
        with torch.no_grad():
            stem = ref.backbone.stem(x[..., 0])
            feats = ref.backbone(x[..., 0])
            head = ref.head
            head.decode_in_inference = False
            raw = head(ref.neck(feats))
        out[f"{tag}_raw"] = raw.numpy()
        out[f"{tag}_stem_stats"] = np.array([stem.mean().item(), stem.norm().item(), stem.abs().max().item()])
        out[f"{tag}_stem_crop"] = stem[:, :, 40:48, 100:108].numpy()   # (2, 32, 8, 8)
        out[f"{tag}_params"] = np.array(sum(p.numel() for p in ref.parameters()))
    np.savez_compressed(os.path.join(HERE, "detector_bfm.npz"), **out)
    print("detector_bfm.npz", os.path.getsize(os.path.join(HERE, "detector_bfm.npz")) // 1024, "KiB")


def nms_primitive(boxes, scores, iou_threshold):
    """Stand-in for the ONE call the image cannot run, ``torchvision.ops.nms`` (yolo_head.py:281; torchvision==0.5.0,
    requirements.txt:48, is neither in the reference tree nor in this image).  Its documented contract, as a plain loop:
    visit boxes by descending score (ties: lower index first, the order a stable sort gives), drop every later box whose
    IoU with a kept one is > threshold, IoU = inter / (area_a + area_b - inter) on xyxy without +1, return the kept
    indices in descending-score order.  Everything AROUND this call in the goldens below is the reference's own code."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    sc = scores.detach().cpu().numpy()
    order = np.argsort(-sc, kind="stable")
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(len(b), bool)
    keep = []
    for oi, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[oi + 1:]
        w = np.maximum(np.float32(0), np.minimum(b[i, 2], b[rest, 2]) - np.maximum(b[i, 0], b[rest, 0]))
        h = np.maximum(np.float32(0), np.minimum(b[i, 3], b[rest, 3]) - np.maximum(b[i, 1], b[rest, 1]))
        inter = w * h
        iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou > np.float32(iou_threshold)]] = True
    return torch.from_numpy(np.asarray(keep, dtype=np.int64))


def crafted_raw(A=1680, F=7):
    """(3, A, F) head tensors that exercise the post-processing: image 0 large overlapping boxes with tied objectness,
    image 1 nothing above obj > 0.3 (-> the zeros((1, 8)) row, yolo_head.py:277-278), image 2 a handful of boxes with
    exact duplicates (IoU = 1) and an objectness exactly at the 0.3 threshold (strict >)."""
    rng = np.random.default_rng(20)
    raw = np.zeros((3, A, F), np.float32)
    raw[..., 0:2] = rng.uniform(-0.5, 0.5, (3, A, 2))
    raw[..., 2:4] = rng.uniform(1.0, 3.0, (3, A, 2))
    raw[..., 4] = rng.choice(np.array([0.1, 0.35, 0.5, 0.5, 0.9], np.float32), (3, A))
    raw[..., 5:] = rng.uniform(0, 1, (3, A, F - 5))
    raw[1, :, 4] = 0.1
    raw[2, :, 4] = 0.05
    raw[2, 10, 4], raw[2, 11, 4], raw[2, 12, 4], raw[2, 500, 4], raw[2, 1679, 4] = 0.8, 0.8, np.float32(0.3), 0.6, 0.31
    raw[2, 11, 0:4] = raw[2, 10, 0:4]
    raw[2, 11, 0] += 1.0  # anchor 11 sits one cell right of anchor 10: same decoded box up to the grid offset
    return raw


def main_nms():
    """Post-threshold goldens of yolo_head.py:258-303 produced by the reference's OWN ``YOLOXHead.decode_outputs`` with only
    ``torchvision.ops.nms`` replaced by ``nms_primitive``: the obj > 0.3 filter, the xyxy conversion, the zeros((1, 8)) row
    and the [cx, cy, w, h, argmax cls, obj * max cls] emission are the reference's statements."""
    import torchvision
    torchvision.ops.nms = nms_primitive
    out = {}
    chans = [128, 256, 512]
    ref = RefModel(CSPDarknet(10, 0.33, 0.5, stem=Focus),
                   YOLOPAFPN(0.33, in_features=["dark3", "dark4", "dark5"], in_channels=chans, act="silu"), None,
                   YOLOXHead(2, in_channels=chans, act="silu", strides=[8, 16, 32], radius=5))
    ref.load_state_dict(recipe_state_dict(build_yolox(10, 2), seed=1004))
    ref.eval()
    x = detector_input(1004, 4, 10)
    with torch.no_grad():
        head = ref.head
        fpn = ref.neck(ref.backbone(x[..., 0]))
        head.decode_in_inference = False
        raw = head(fpn).clone()
        head.decode_in_inference = True
        dets = head(fpn)                      # the reference's forward -> decode_outputs (yolo_head.py:232-233)
        out["b4_raw"] = raw.numpy()
        out["b4_counts"] = np.array([len(d) for d in dets])
        out["b4_dets"] = torch.cat(dets).numpy()
        assert all(d.shape[1] == 6 for d in dets)
        craft = torch.from_numpy(crafted_raw())
        head.hw = [(32, 40), (16, 20), (8, 10)]
        dets = head.decode_outputs(craft.clone(), craft.type())
        out["craft_raw"] = craft.numpy()
        out["craft_counts"] = np.array([len(d) for d in dets])
        out["craft_dets"] = torch.cat(dets).numpy()
        assert dets[1].shape == (1, 6) and float(dets[1].abs().sum()) == 0.0
        print("counts", out["b4_counts"], out["craft_counts"])
    np.savez_compressed(os.path.join(HERE, "detector_nms.npz"), **out)
    print("detector_nms.npz", os.path.getsize(os.path.join(HERE, "detector_nms.npz")) // 1024, "KiB")


def main_b32():
    """BASELINE.json configs[3] at its full size: B = 32, (10, 256, 320), recipe weights, eval -- the REFERENCE's modules on
    torch-CPU fp32 (373 GFLOP: a minute or two on eight threads).  The (32, 1680, 7) head tensor is kept as its sha256, its
    per-channel |max|, three whole images and 8192 sampled values, so that the GPU engine at the bench configuration is held to
    the reference itself and not to another GPU library (core/yolox/models/yolo_head.py:209-235)."""
    import hashlib
    chans = [128, 256, 512]
    ref = RefModel(CSPDarknet(10, 0.33, 0.5, stem=Focus),
                   YOLOPAFPN(0.33, in_features=["dark3", "dark4", "dark5"], in_channels=chans, act="silu"), None,
                   YOLOXHead(2, in_channels=chans, act="silu", strides=[8, 16, 32], radius=5))
    ref.load_state_dict(recipe_state_dict(build_yolox(10, 2), seed=1004))
    ref.eval()
    x = detector_input(1004, 32, 10)
    with torch.no_grad():
        ref.head.decode_in_inference = False
        raw = ref.head(ref.neck(ref.backbone(x[..., 0]))).numpy()
    assert raw.shape == (32, 1680, 7) and raw.dtype == np.float32
    idx = np.sort(np.random.default_rng(3204).choice(raw.size, size=8192, replace=False))
    out = {"b32_shape": np.array(raw.shape), "b32_sha": np.array(hashlib.sha256(raw.tobytes()).hexdigest()),
           "b32_absmax": np.abs(raw).reshape(-1, 7).max(axis=0), "b32_idx": idx, "b32_val": raw.reshape(-1)[idx],
           "b32_images": np.array([0, 7, 31]), "b32_img": raw[[0, 7, 31]]}
    np.savez_compressed(os.path.join(HERE, "detector_b32.npz"), **out)
    print("detector_b32.npz", os.path.getsize(os.path.join(HERE, "detector_b32.npz")) // 1024, "KiB")


if __name__ == "__main__":
    if "--b32-only" in sys.argv:
        main_b32()
        sys.exit(0)
    if "--nms-only" in sys.argv:
        main_nms()
        sys.exit(0)
    if "--bfm-only" not in sys.argv:
        main()
        main_nms()
        main_b32()
    main_bfm()
