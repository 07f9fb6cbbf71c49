"""Dataset-side sample transform of the training loader (reference: data/dataset.py:116-239,
``propheseeDataset.__getitem__``): ``/255``, nearest zoom-in by ``sr`` in [1, 1.5], crop, horizontal flip, and
the matching label transform -- the image part as ONE batched gfx950 kernel (``frlw_sample_transform_u8``)
on the uint8 tensors the encoders leave in HBM, instead of DataLoader worker processes.

The random draws and the box arithmetic are host logic (a handful of boxes per sample) and follow the reference
statement by statement, including its quirks: the order and short-circuiting of the ``random`` calls, the
dead ``sr < 1`` branch, and the 100-retry rule that keeps the LAST drawn image parameters while resetting the
boxes to the un-augmented ones.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
from numpy.lib import recfunctions as rfn

MAX_LABELS = 80


class SampleParams:
    """What one pass of the reference's augmentation loop decided for a sample."""

    __slots__ = ("sr", "flip", "cx", "cy")

    def __init__(self, sr=1.0, flip=False, cx=0, cy=0):
        self.sr, self.flip, self.cx, self.cy = sr, flip, cx, cy

    def resized(self, input_img_size):
        """(int(H * sr), int(W * sr)): the size F.interpolate is asked for (data/dataset.py:221)."""
        return int(input_img_size[0] * self.sr), int(input_img_size[1] * self.sr)


def _boxes_xyxy(bboxes, rw, rh, cx, cy):
    a = rfn.structured_to_unstructured(bboxes)[:, [1, 2, 3, 4, 5, 0, 6, 7]]  # x, y, w, h, class, t, conf, track
    return np.stack([a[:, 0] * rw + cx, a[:, 1] * rh + cy, (a[:, 0] + a[:, 2]) * rw + cx, (a[:, 1] + a[:, 3]) * rh + cy,
                     a[:, 4], a[:, 5], a[:, 6], a[:, 7]], axis=-1)


def _clip(b, size):
    np.clip(b[:, 0], 0, size[1], out=b[:, 0])
    np.clip(b[:, 1], 0, size[0], out=b[:, 1])
    np.clip(b[:, 2], 0, size[1], out=b[:, 2])
    np.clip(b[:, 3], 0, size[0], out=b[:, 3])


def sample_labels(bboxes, rnd, input_img_size, sensor_hw, dataset="gen1", mode="train", augment=True, clipping=False):
    """The label half of ``__getitem__`` (data/dataset.py:119-216).

    ``bboxes``: the structured box records of ONE label time (fields t, x, y, w, h, class_id, class_confidence,
    track_id; sensor pixels); ``rnd``: a ``random.Random`` (or the ``random`` module) -- the draws are made in
    the reference's order.  Returns ``(padded_labels (80, 5 | 8) float64, SampleParams)``."""
    H, W = input_img_size
    rh_ori, rw_ori = H / sensor_hw[0], W / sensor_hw[1]
    unique_ts = np.unique(bboxes["t"])
    count = 0
    ng = True
    p = SampleParams()
    while ng:
        p.sr = rnd.uniform(1.0, 1.5) if (augment and rnd.random() < 0.5) else 1.0
        p.flip = bool(augment and rnd.random() < 0.5)
        rh, rw = p.sr * rh_ori, p.sr * rw_ori
        if p.sr < 1.0:  # never taken (sr >= 1); kept for the draw order
            p.cx = int(rnd.uniform(0, int(W - p.sr * W)))
            p.cy = int(rnd.uniform(0, int(H - p.sr * H)))
        if p.sr > 1.0:
            p.cx = int(rnd.uniform(int(W - p.sr * W), 0))
            p.cy = int(rnd.uniform(int(H - p.sr * H), 0))
        else:
            p.cx = p.cy = 0
        b = _boxes_xyxy(bboxes, rw, rh, p.cx, p.cy)
        if dataset == "gen4":
            if augment:
                _clip(b, input_img_size)
                b = b[(b[:, 2] - b[:, 0] > 5) & (b[:, 3] - b[:, 1] > 5)]
        elif augment:
            b = b[(b[:, 2] > 10) & (b[:, 0] < W - 10) & (b[:, 1] < H - 10) & (b[:, 3] > 10)]
        for t in unique_ts:
            ng = len(b[b[:, 5] == t]) == 0
            if ng:
                break
        count += 1
        if count > 100:  # give up: un-augmented boxes, but the image keeps the last drawn sr / crop / flip
            b = _boxes_xyxy(bboxes, rw_ori, rh_ori, 0, 0)
            break
    if (mode == "train" and clipping) or dataset == "gen4":
        _clip(b, input_img_size)
    boxes = b[:, :4].copy()
    labels = b[:, 4:].copy()
    if p.flip:
        boxes[:, 0::2] = W - boxes[:, 2::-2] - 1
    boxes[:, 2] = boxes[:, 2] - boxes[:, 0]  # xyxy2cxcywh, data/utils.py:3-8
    boxes[:, 3] = boxes[:, 3] - boxes[:, 1]
    boxes[:, 0] = boxes[:, 0] + boxes[:, 2] * 0.5
    boxes[:, 1] = boxes[:, 1] + boxes[:, 3] * 0.5
    targets = np.hstack((labels[:, 0:1], boxes)) if mode == "train" else np.hstack((boxes, labels))
    padded = np.zeros((MAX_LABELS, targets.shape[1]), dtype=float)
    padded[range(len(targets))] = targets
    return padded, p


def transform_images(u8, params, out=None):
    """The image half for a batch (data/dataset.py:217-231): ``u8`` (B, C, H, W) uint8 on the GPU, ``params`` a list
    of B :class:`SampleParams` -> (B, C, H, W, 1, 1) float32: nearest resize to ``(int(H sr), int(W sr))``, ``/255``,
    crop ``[-cy : H - cy, -cx : W - cx]``, horizontal flip.  One launch; no CPU fallback."""
    import torch

    from . import _lib
    lib = _lib.load()
    if not (u8.is_cuda and u8.dtype == torch.uint8 and u8.dim() == 4):
        raise ValueError("transform_images needs a (B, C, H, W) uint8 tensor on the GPU")
    u8 = u8.contiguous()
    B, Cc, H, W = u8.shape
    if len(params) != B:
        raise ValueError("one SampleParams per image")
    tab = np.zeros((B, 5), dtype=np.int32)  # resized H, resized W, crop y0, crop x0, flip
    for b, p in enumerate(params):
        hr, wr = p.resized((H, W))
        tab[b] = (hr, wr, -p.cy, -p.cx, 1 if p.flip else 0)
        if hr < 1 or wr < 1 or -p.cy < 0 or -p.cx < 0 or -p.cy + H > max(hr, H) or -p.cx + W > max(wr, W):
            raise ValueError(f"sample {b}: crop window outside the resized image")
    tab_d = torch.from_numpy(tab).to(u8.device, non_blocking=True)
    if out is None:
        out = torch.empty((B, Cc, H, W, 1, 1), dtype=torch.float32, device=u8.device)
    _lib.check(lib.frlw_sample_transform_u8(u8.data_ptr(), B, Cc, H, W, tab_d.data_ptr(), out.data_ptr(),
                                            torch.cuda.current_stream(u8.device).cuda_stream), "frlw_sample_transform_u8")
    return out
