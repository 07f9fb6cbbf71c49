"""End-to-end synthetic harness of BASELINE.json config 5: TAF encode (HIP) -> detector input -> YOLOX train /
eval step, one process per GPU (SURVEY.md section 8d cfg 5).

Per sample: a 304x240 stream of 8 windows x 125 000 events -> TAF K=8 (16 channels) -> leaky transform ->
uint8 (the file the reference would write, generate_taf.py:228-235) -> /255 (data/dataset.py:294-308) ->
nearest resize to 256x320 -> YOLOX(16-channel Focus stem) with 2 GT boxes.  Real datasets, augmentation and
the evaluator are out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import event_representation as er
from . import synth
from .yolox import build_yolox
from .yolox.model import recipe_state_dict

GEN1_SENSOR = (240, 304)
GEN1_DETECTOR = (256, 320)


class SyntheticTafSource:
    """Independent per-sample event streams (one sequence each), resident on the GPU as raw DAT records."""

    def __init__(self, n_samples, seed=1005, events_per_window=125_000, n_windows=8, device="cuda"):
        H, W = GEN1_SENSOR
        self.n_windows, self.K = n_windows, 8
        recs = [synth.to_dat8(synth.synth_events(seed + i, events_per_window * n_windows, W, H, 10_000 * n_windows))
                for i in range(n_samples)]
        self.offsets = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.int64)
        # all sequences back to back: sample i owns records [offsets[i], offsets[i + 1]) -- what frlw_taf_encode_batch
        # takes.  Every sequence keeps its own FIFO state and its own "window without events" rule
        # (generate_taf.py:40-41 is evaluated per file), unlike a frame of stacked samples.
        self.dat = torch.from_numpy(np.concatenate(recs).view(np.uint8).reshape(-1, 8)).to(device)
        self.device = device

    def __len__(self):
        return len(self.offsets) - 1

    def labels(self, n):
        """(n, 80, 5) float64 [cls, cx, cy, w, h] with 2 boxes per sample, zero-padded (data/dataset.py:211-217)."""
        rng = np.random.default_rng(99)
        lab = torch.zeros((n, 80, 5), dtype=torch.float64)
        for i in range(n):
            for j in range(2):
                lab[i, j] = torch.tensor([rng.integers(0, 2), rng.uniform(60, 260), rng.uniform(50, 200),
                                          rng.uniform(20, 80), rng.uniform(20, 80)])
        return lab.to(self.device)

    def encode_u8(self, idx, batched=True):
        """-> (B, 16, 240, 304) uint8, the files the reference would write (generate_taf.py:228-235), newest slot first.
        ``batched``: runs of consecutive samples go through ONE ``frlw_taf_encode_batch`` call (up to 64 sequences);
        otherwise one general-path encode per sample -- the two agree bit for bit (tests/test_e2e_gpu.py)."""
        H, W = GEN1_SENSOR
        idx = list(idx)
        out = torch.empty((len(idx), 2 * self.K, H, W), dtype=torch.uint8, device=self.device)
        pos = 0
        while pos < len(idx):
            run = 1
            if batched:
                while pos + run < len(idx) and run < 64 and idx[pos + run] == idx[pos + run - 1] + 1:
                    run += 1
            i0 = idx[pos]
            lo, hi = int(self.offsets[i0]), int(self.offsets[i0 + run])
            if batched:
                state = torch.full((run, H, W, 2, self.K), -6000.0, device=self.device)
                u8, _ = er.encode_taf_batch(self.dat[lo:hi], self.offsets[i0:i0 + run + 1] - lo, (H, W), state, 0, 10_000,
                                            self.n_windows, self.K, check=False)
                out[pos:pos + run] = u8.reshape(run, 2 * self.K, H, W)
            else:
                state = torch.full((H, W, 2, self.K), -6000.0, device=self.device)
                u8, _ = er.encode_taf_dat(self.dat[lo:hi], (H, W), state, 0, 10_000, self.n_windows, self.K, check=False,
                                          fast=False)
                out[pos] = u8.reshape(2 * self.K, H, W)
            pos += run
        er.raise_deferred("SyntheticTafSource.encode_u8")  # one sync per batch: never train on a stale / unwritten state
        return out

    def encode_batch(self, idx, batched=True):
        """-> (B, 16, 256, 320, 1, 1) f32 in [0, 1]: what propheseeTafDataset hands to the model (uint8 file / 255,
        data/dataset.py:294-308, after the nearest resize to the detector shape, generate_taf.py:221-222)."""
        u8 = self.encode_u8(idx, batched)
        B = u8.shape[0]
        u8 = er.resize_nearest(u8.reshape(B * 2 * self.K, *GEN1_SENSOR), GEN1_DETECTOR).reshape(B, 2 * self.K, *GEN1_DETECTOR)
        return (u8.float() / 255.0)[..., None, None]


class EncodeAhead:
    """The encode of batch i + 1 on its own HIP stream while the train step of batch i runs on the current one: the encoders
    are VALU / HBM work, the step's convolutions sit on the matrix cores, and the step leaves the host idle until its loss
    is read back.  ``start(idx)`` right after a step has been queued (``Trainer.train_step(after_launch=...)``), ``take()``
    before the next one.  Every batch is encoded exactly once into its own buffer; the consumer's stream is recorded on it."""

    def __init__(self, source, batched=True):
        self.src, self.batched = source, batched
        self.side = torch.cuda.Stream(device=source.device)
        self.ready = None

    def start(self, idx):
        with torch.cuda.stream(self.side):
            self.ready = self.src.encode_batch(idx, self.batched)  # ends with the deferred-status check of ITS stream only

    def take(self):
        if self.ready is None:
            raise RuntimeError("EncodeAhead.take() without start()")
        main = torch.cuda.current_stream(self.src.device)
        main.wait_stream(self.side)
        t, self.ready = self.ready, None
        t.record_stream(main)
        return t


def build_model(in_channels=16, num_classes=2, device="cuda", seed=1004):
    net = build_yolox(in_channels, num_classes)
    net.load_state_dict(recipe_state_dict(net, seed=seed))
    return net.to(device)
