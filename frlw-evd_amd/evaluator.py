"""Evaluator hand-off (reference: evaluate/evaluator.py:9-133, evaluate/src/io/box_filtering.py:17-47).

Same class names, constructor arguments and methods as the reference's ``evaluator`` / ``recorder``:
``add_result`` rescales detections and ground truth from detector to sensor pixels and collects them per image,
the Prophesee minimum-size / first-0.5-s filters are applied at ``evaluate`` time, ``recorder.save`` writes
``summarise.npz``.  Detections that are still on the GPU go through one kernel for the whole batch
(``frlw_eval_transform_dt``: rescale + the filter mask) instead of one ``.cpu()`` per image.
COCO mAP itself is third-party (pycocotools, evaluate/src/metrics/coco_eval.py) and not part of this build:
``evaluate`` returns the filtered, paired lists that ``evaluate_detection`` consumes.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def filter_boxes(boxes, skip_ts=int(5e5), min_box_diag=60, min_box_height=20, min_box_width=20):
    """box_filtering.py:17-39 on (n, 8) rows [t, x, y, w, h, class, confidence, track]."""
    ts, width, height = boxes[:, 0], boxes[:, 3], boxes[:, 4]
    diag_square = width ** 2 + height ** 2
    mask = (ts > skip_ts) * (diag_square >= min_box_diag ** 2) * (width >= min_box_width) * (height >= min_box_height)
    return boxes[mask]


FILTERS = {"gen1": (5e5, 30, 10, 10), "kitti": (0, 0, 25, 0), "large": (5e5, 60, 20, 20)}  # skip_ts, diag, h, w


def filter_boxes_gen1(boxes):
    return filter_boxes(boxes, *FILTERS["gen1"])


def filter_boxes_large(boxes):
    return filter_boxes(boxes, *FILTERS["large"])


def filter_boxes_kitti(boxes):
    return filter_boxes(boxes, *FILTERS["kitti"])


class evaluator:
    def __init__(self, classes, batchsize, infer_time, ori_width, ori_height, input_width, input_height, dataset="gen1",
                 recorder=None):
        self.dt_to_eval = []
        self.gt_to_eval = []
        self.rw = ori_width / input_width
        self.rh = ori_height / input_height
        self.ori_width = ori_width
        self.ori_height = ori_height
        self.batchsize = batchsize
        self.infer_time = 0
        self.represent_time = 0
        self.infer_count = 0
        self.first_batch = True
        self.classes = classes
        self._filter_key = dataset if dataset in ("gen1", "kitti") else "large"
        self.filter_boxes = {"gen1": filter_boxes_gen1, "kitti": filter_boxes_kitti, "large": filter_boxes_large}[self._filter_key]
        self.tol = int(infer_time / 2 - 1)
        self.recorder = recorder

    def cal_time(self, infer_time, represent_time):
        if self.first_batch:  # the first batch is warm-up (evaluator.py:34-41)
            self.first_batch = False
        else:
            self.infer_time += infer_time
            self.represent_time += represent_time
            self.infer_count += 1

    def _scale(self):
        return np.array([self.rw, self.rh, self.rw, self.rh])

    def transform_gt(self, bounding_box):
        """(80, 8) [cx, cy, w, h, class, t, confidence, track] -> rows [t, x, y, w, h, class, confidence, track]
        in sensor pixels; padding rows (confidence 0) dropped (evaluator.py:43-54)."""
        return self.transform_gt_batch(bounding_box[None])[0]

    def transform_gt_batch(self, bounding_boxes):
        """Whole batch (B, 80, 8) in one device->host copy -> list of B arrays (n_i, 8), float64 like the labels."""
        g = (bounding_boxes.cpu().numpy() if torch.is_tensor(bounding_boxes) else np.asarray(bounding_boxes))
        corner = np.concatenate([g[..., 0:2] - g[..., 2:4] / 2, g[..., 2:4]], axis=-1) * self._scale()
        rows = np.concatenate([g[..., 5:6], corner, g[..., 4:5], g[..., 6:8]], axis=-1)
        return [r[c > 0] for r, c in zip(rows, g[..., 6])]

    def transform_dt(self, detected_bbox, bins_time_stamp):
        """(n, 6) [cx, cy, w, h, class, score] -> (n, 8) [t, x, y, w, h, class, score, 0] (evaluator.py:56-63);
        host-side twin of ``frlw_eval_transform_dt`` for detections that are not on the GPU, same float32 arithmetic."""
        d = detected_bbox.reshape(-1, 6)
        sc = torch.tensor([self.rw, self.rh, self.rw, self.rh])
        corner = torch.cat([d[:, 0:2] - d[:, 2:4] / 2, d[:, 2:4]], dim=1) * sc.to(d.device)
        out = np.zeros((d.shape[0], 8), dtype=corner.cpu().numpy().dtype)
        out[:, 0] = int(bins_time_stamp)
        out[:, 1:5] = corner.cpu().numpy()
        out[:, 5:7] = d[:, 4:6].cpu().numpy()
        return out

    def transform_dt_batch(self, outputs, bins_time_stamps):
        """All images of a batch at once on the GPU -> list of (n_i, 8) float32 arrays (one device->host copy)."""
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        dev = outputs[0].device
        rows = torch.cat([o.reshape(-1, 6).float() for o in outputs], 0).contiguous()
        n = rows.shape[0]
        img = torch.repeat_interleave(torch.arange(len(outputs), dtype=torch.int32),
                                      torch.tensor([o.shape[0] for o in outputs])).to(dev)
        ts = torch.tensor([int(t) for t in bins_time_stamps], dtype=torch.int64, device=dev)
        out = torch.empty((n, 8), dtype=torch.float32, device=dev)
        keep = torch.empty((n,), dtype=torch.uint8, device=dev)
        skip_ts, diag, min_h, min_w = FILTERS[self._filter_key]
        _lib.check(lib.frlw_eval_transform_dt(rows.data_ptr(), img.data_ptr(), ts.data_ptr(), n, C.c_float(self.rw),
                                              C.c_float(self.rh), C.c_float(skip_ts), C.c_float(diag * diag),
                                              C.c_float(min_w), C.c_float(min_h), out.data_ptr(), keep.data_ptr(),
                                              torch.cuda.current_stream(dev).cuda_stream), "frlw_eval_transform_dt")
        out_h, keep_h = out.cpu().numpy(), keep.cpu().numpy().astype(bool)
        sizes = np.cumsum([0] + [o.shape[0] for o in outputs])
        return [out_h[a:b] for a, b in zip(sizes[:-1], sizes[1:])], [keep_h[a:b] for a, b in zip(sizes[:-1], sizes[1:])]

    def add_result(self, outputs, bins_time_stamps, bounding_box, filename, infer_time, represent_time):
        self.cal_time(infer_time, represent_time)
        on_gpu = len(outputs) > 0 and all(torch.is_tensor(o) and o.is_cuda for o in outputs)
        dts = self.transform_dt_batch(outputs, bins_time_stamps)[0] if on_gpu else None
        gts = self.transform_gt_batch(torch.stack(list(bounding_box)) if not torch.is_tensor(bounding_box) else bounding_box)
        for i in range(len(outputs)):
            gt_trans = gts[i]
            if len(gt_trans) == 0:
                continue
            self.gt_to_eval.append(gt_trans)
            dt_trans = dts[i] if on_gpu else self.transform_dt(outputs[i], bins_time_stamps[i])
            self.dt_to_eval.append(dt_trans)
            if self.recorder is not None:
                self.recorder.record(dt_trans, filename[i])

    def end_a_batch(self):
        pass

    def filtered_lists(self):
        """The pairing rule of ``evaluate`` (evaluator.py:87-98): images whose filtered ground truth is empty are
        dropped; an image with no surviving detection gets one all-zero detection at the GT time."""
        gts, dts = [], []
        for g, d in zip(map(self.filter_boxes, self.gt_to_eval), map(self.filter_boxes, self.dt_to_eval)):
            if len(g) > 0:
                gts.append(g)
                dts.append(np.array([[g[0, 0], 0, 0, 0, 0, 0, 0, 0]]) if len(d) == 0 else d)
        return gts, dts

    def evaluate(self, metric_fn=None):
        """``metric_fn``: the COCO scorer to hand the paired lists to -- the reference calls
        ``evaluate_detection(gt_boxes_list, dt_boxes_list, time_tol=, classes=, height=, width=)``
        (evaluate/src/metrics/coco_eval.py, pycocotools).  It is third-party and never imported from here: without a
        callable the keyword arguments of that call are returned."""
        gts, dts = self.filtered_lists()
        if self.recorder is not None:
            self.recorder.save()
        kw = {"time_tol": self.tol, "classes": self.classes, "height": self.ori_height, "width": self.ori_width}
        if metric_fn is not None:
            return metric_fn(gts, dts, **kw)
        return dict(kw, gt_boxes_list=gts, dt_boxes_list=dts,
                    avg_infer_ms=1000 * self.infer_time / max(self.infer_count, 1))


class recorder:
    """evaluator.py:117-133: every detection row with its file name -> ``summarise.npz``."""

    def __init__(self, save_path):
        self.data_names = []
        self.dt = []
        self.save_path = save_path

    def record(self, dt_trans, file_name):
        for j in range(len(dt_trans)):
            self.data_names.append(file_name)
            self.dt.append(dt_trans[j])

    def save(self):
        path = os.path.join(self.save_path, "summarise.npz")
        np.savez(path, file_names=self.data_names, dts=self.dt)
        return path
