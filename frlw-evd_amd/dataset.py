"""Disk-backed datasets and loader of the entry points -- ``propheseeDataset`` / ``propheseeTafDataset`` / ``Loader`` with
the reference's constructor signatures and batch format (data/dataset.py:23-308, data/loader.py:7-46), over the files the
``generate_*.py`` commands write (``<data_dir>/<mode>/<seq>_<ts>.npy`` raw uint8 volumes; TAF: ``bins4/`` + ``bins8/``) and
the ``<bbox_dir>/<mode>/<seq>_bbox.npy`` annotations.

MI355X-first data path: a sample stays ``uint8`` until it is on the GPU.  Reader threads fill a pinned ``(B, C, H, W)``
uint8 batch (a quarter of the float32 bytes the reference's workers ship over PCIe), the next batch is read while the
current one trains, and ``/255`` + zoom + crop + flip run as ONE kernel on the batch (``transforms.transform_images``)
instead of per sample in DataLoader worker processes.  The label half (``transforms.sample_labels``) is the reference's
host arithmetic, statement by statement.

One documented deviation: ``propheseeDataset.load_data`` at the reference's HEAD replaces the volume by the mean over its
channels, stacked twice (data/dataset.py:245) -- two channels, where the experiment that uses this class builds a network
with ``2 * event_volume_bins`` input channels (core/exp.py:47,582): run as shipped, the ``yolox`` recipe of README.md:112
raises in its first convolution.  Here all ``2 * time_channels`` channels of the file are loaded;
``reference_mean_quirk=True`` reproduces the HEAD line (pinned by tests/golden/dataset_files.npz either way).
"""
from __future__ import annotations

import os
import random
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import transforms

GEN1_CLASSES = ["Car", "Pedestrian"]  # data/dataset.py:52-63
GEN4_CLASSES = ["pedestrian", "two wheeler", "car", "truck", "bus", "traffic sign", "traffic light"]


def read_boxes(bbox_file):
    """The structured box records of a ``*_bbox.npy`` file (timestamp field ``t``; ``ts`` / ``confidence`` in older files
    are renamed like src/io/npy_events_tools.py:57-58 does)."""
    boxes = np.load(bbox_file)
    names = list(boxes.dtype.names)
    renamed = [{"ts": "t", "confidence": "class_confidence"}.get(n, n) for n in names]
    if renamed != names:
        boxes = boxes.copy()
        boxes.dtype.names = tuple(renamed)
    return boxes


class propheseeDataset:
    def __init__(self, bbox_dir, data_dir, dataset="gen1", input_img_size=[256, 320], img_size=[256, 320], time_channels=5,
                 infer_time=10000, train_memory_steps=1, mode="train", augment=True, clipping=False,
                 reference_mean_quirk=False):
        self.mode, self.augment = mode, augment
        file_dir = os.path.join(bbox_dir, self.mode)
        self.files = [f[:-9] for f in os.listdir(file_dir) if f[-3:] == "npy"]  # "<seq>_bbox.npy" -> "<seq>"
        self.root, self.data_dir = file_dir, data_dir
        if dataset == "gen1":
            self.width, self.height, self.object_classes = 304, 240, list(GEN1_CLASSES)
        elif dataset == "kitti":
            self.width, self.height, self.object_classes = 1242, 375, list(GEN1_CLASSES)
        else:
            self.width, self.height, self.object_classes = 1280, 720, list(GEN4_CLASSES)
        self.clipping, self.dataset = clipping, dataset
        self.input_img_size, self.img_size, self.time_channels = input_img_size, img_size, time_channels
        self.infer_time, self.train_memory_steps = infer_time, train_memory_steps
        self.reference_mean_quirk = reference_mean_quirk
        self.sequence_end_t = []
        self._boxes = {}
        self.createAllBBoxDataset()
        self.nr_samples = len(self.file_name)

    # ---- sample list (data/dataset.py:78-113): every annotated timestamp whose representation file exists ----------------
    def _sample_root(self):
        return os.path.join(self.data_dir, self.mode)

    def createAllBBoxDataset(self):
        file_names = []
        root = self._sample_root()
        for file_name in self.files:
            boxes = self._file_boxes(file_name)
            for unique_time in np.unique(boxes["t"]):
                if os.path.exists(os.path.join(root, f"{file_name}_{unique_time}.npy")):
                    self.sequence_end_t.append(unique_time)
                    file_names.append(file_name)
        self.file_name = file_names

    def _file_boxes(self, file_name):
        b = self._boxes.get(file_name)
        if b is None:  # (the reference re-reads the annotation file for every sample; once per sequence is the same data)
            b = self._boxes[file_name] = read_boxes(os.path.join(self.root, file_name + "_bbox.npy"))
        return b

    def __len__(self):
        return len(self.file_name)

    # ---- one sample ----------------------------------------------------------------------------------------------------
    @property
    def channels(self):
        return 2 if self.reference_mean_quirk else int(2 * self.time_channels)

    def sample_files(self, idx):
        """The representation file(s) of sample ``idx``, in channel order."""
        return [os.path.join(self._sample_root(), f"{self.file_name[idx]}_{self.sequence_end_t[idx]}.npy")]

    def load_u8(self, idx, out=None):
        """The sample's volume as the files hold it: ``(C, H, W)`` uint8 (into ``out`` when given)."""
        H, W = self.img_size
        parts = [np.fromfile(f, dtype=np.uint8) for f in self.sample_files(idx)]
        vol = (parts[0] if len(parts) == 1 else np.concatenate(parts)).reshape(-1, H, W)
        if out is None:
            return vol
        out[...] = vol
        return out

    def load_data(self, idx):
        """data/dataset.py:238-247: the float32 volume ``__getitem__`` transforms (CPU; the loader keeps uint8 instead)."""
        volume = self.load_u8(idx).reshape(int(2 * self.time_channels), self.img_size[0], self.img_size[1]).astype(np.float32)
        if self.reference_mean_quirk:
            volume = np.stack([volume.mean(0), volume.mean(0)])
        return volume

    def labels(self, idx, rnd=random):
        """(padded labels (80, 5 | 8) float64, SampleParams): the label half of ``__getitem__`` (data/dataset.py:119-216);
        ``rnd``: where the augmentation draws come from (the reference uses the ``random`` module)."""
        boxes = self._file_boxes(self.file_name[idx])
        bboxes = boxes[boxes["t"] == self.sequence_end_t[idx]]
        return transforms.sample_labels(bboxes, rnd, self.input_img_size, (self.height, self.width), self.dataset, self.mode,
                                        self.augment, self.clipping)

    def __getitem__(self, idx):
        """-> (uint8 volume (C, H, W), padded labels, SampleParams, sequence name, label time): the image half of the
        reference's ``__getitem__`` (``/255``, zoom, crop, flip: :217-231) happens on the GPU, per batch (``Loader``)."""
        padded, params = self.labels(idx)
        return self.load_u8(idx), padded, params, self.file_name[idx], self.sequence_end_t[idx]


class propheseeTafDataset(propheseeDataset):
    """data/dataset.py:254-308: samples listed from ``bins8/``, volume = ``bins{K/2}`` ++ ``bins{K}`` (K > 4) or ``bins{K}``."""

    def __init__(self, bbox_dir, data_dir, dataset="gen1", input_img_size=[256, 320], img_size=[256, 320], infer_time=10000,
                 event_volume_bins=5, mode="train", augment=True, clipping=False):
        super().__init__(bbox_dir, data_dir, dataset, input_img_size, img_size, event_volume_bins, infer_time, 1, mode, augment,
                         clipping)

    def _sample_root(self):
        return os.path.join(self.data_dir, self.mode, "bins8")

    def sample_files(self, idx):
        root = os.path.join(self.data_dir, self.mode)
        name = f"{self.file_name[idx]}_{self.sequence_end_t[idx]}.npy"
        K = int(self.time_channels)
        if K > 4:
            return [os.path.join(root, f"bins{K // 2}", name), os.path.join(root, f"bins{K}", name)]
        return [os.path.join(root, f"bins{K}", name)]

    def load_data(self, idx):
        return self.load_u8(idx).astype(np.float32)


class Loader:
    """``Loader(dataset, batch_size, num_workers, pin_memory, device, shuffle=True, sampler=None)`` (data/loader.py:7-32):
    iterating yields ``[imgs (B, C, H, W, 1, 1) float32 on the GPU, labels (B, 80, 5 | 8) float64 on the GPU, names,
    timestamps]``; ``len()`` = batches per epoch (the last one may be short: ``drop_last=False``).

    ``num_workers`` reader THREADS (file reads release the GIL) fill a pinned uint8 batch while the previous batch trains."""

    def __init__(self, dataset, batch_size, num_workers, pin_memory, device, shuffle=True, sampler=None):
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.sampler = shuffle, sampler
        self.num_workers = max(1, int(num_workers) if num_workers else 1)
        self.pin = bool(pin_memory) and torch.cuda.is_available()

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return (n + self.batch_size - 1) // self.batch_size

    def _order(self):
        if self.sampler is not None:
            return [int(i) for i in self.sampler]
        idx = list(range(len(self.dataset)))
        if self.shuffle:  # torch.utils.data.SubsetRandomSampler: a random permutation per epoch
            idx = [idx[i] for i in torch.randperm(len(idx)).tolist()]
        return idx

    def _read(self, batch):
        ds = self.dataset
        H, W = ds.img_size
        if ds.reference_mean_quirk:  # data/dataset.py:245 as shipped: numpy's float32 channel mean, stacked twice, on the host
            buf = torch.from_numpy(np.stack(list(self._pool.map(lambda i: ds.load_data(i), batch))))
        else:
            buf = torch.empty((len(batch), ds.channels, H, W), dtype=torch.uint8, pin_memory=self.pin)
            view = buf.numpy()
            list(self._pool.map(lambda j: ds.load_u8(batch[j], view[j]), range(len(batch))))
        labels, params = zip(*[ds.labels(i) for i in batch])  # (host arithmetic on a handful of boxes; draws in sample order)
        return buf, np.stack(labels), list(params), [ds.file_name[i] for i in batch], np.array([ds.sequence_end_t[i] for i in batch])

    def __iter__(self):
        order = self._order()
        batches = [order[i:i + self.batch_size] for i in range(0, len(order), self.batch_size)]
        if not batches:
            return
        with ThreadPoolExecutor(self.num_workers) as self._pool, ThreadPoolExecutor(1) as ahead:
            nxt = ahead.submit(self._read, batches[0])
            for k in range(len(batches)):
                buf, labels, params, names, stamps = nxt.result()
                if k + 1 < len(batches):
                    nxt = ahead.submit(self._read, batches[k + 1])  # read while this batch trains
                if self.dataset.reference_mean_quirk:  # the HEAD line's path stays on the host end to end, with the reference's
                    imgs = self._transform_f32(buf, params).to(self.device, non_blocking=True)  # own torch-CPU operations
                else:
                    imgs = transforms.transform_images(buf.to(self.device, non_blocking=True), params)
                yield [imgs, torch.from_numpy(labels).to(self.device, non_blocking=True), names, stamps]

    @staticmethod
    def _transform_f32(vol, params):
        """The image half of ``__getitem__`` on float32 volumes with plain torch ops ON THE HOST (only the HEAD quirk's path
        needs it: its mean is not a uint8 any more, and torch's GPU division is not the correctly rounded one of the CPU)."""
        B, C, H, W = vol.shape
        out = torch.empty((B, C, H, W, 1, 1), dtype=torch.float32, device=vol.device)
        for b, p in enumerate(params):
            hr, wr = p.resized((H, W))
            img = torch.nn.functional.interpolate(vol[b:b + 1], size=(hr, wr), mode="nearest")[0] / 255
            img = img[:, -p.cy:H - p.cy, -p.cx:W - p.cx]
            if p.flip:
                img = img.flip(-1)
            out[b, ..., 0, 0] = img
        return out
