"""Experiment objects of the entry points -- ``yolox(settings)`` / ``yoloxtafBFM(settings)`` with ``.train()`` /
``.test()`` like the reference's (core/exp.py:44-391,580-591): same build steps (``configModel``, ``buildBackbone`` ...
``buildModel`` with DistributedDataParallel(broadcast_buffers=False)), same optimiser / LR schedule / GradScaler quirk,
same checkpoint files and dict layout (``saveCheckpoint`` :198-210, ``loadCheckpoint`` :155-165), so checkpoints
interchange with the reference's -- pinned by tests/golden/entry_points.json.

Data: with ``--bbox_path`` / ``--data_path`` the pre-encoded ``uint8`` files of the ``generate_*.py`` commands are read by
``frlw_evd_amd.dataset`` (the reference's dataset classes and ``Loader``: uint8 batches over PCIe, the sample transform as
one kernel on the GPU).  A run without them draws SYNTHETIC event streams and encodes them on the GPU every step
(``e2e.SyntheticTafSource``), which is BASELINE.json's config 5.  The evaluator hand-off is the real one
(``frlw_evd_amd.evaluator``); COCO mAP needs pycocotools and is reported only when a ``metric_fn`` is injected.
"""
from __future__ import annotations

import os
from math import ceil

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel

from .evaluator import evaluator, recorder
from .trainer import LRScheduler
from .yolox.darknet import CSPDarknet
from .yolox.model import model
from .yolox.network_blocks import Focus
from .yolox.yolo_head import YOLOXHead
from .yolox.yolo_pafpn import YOLOPAFPN

GEN1_CLASSES = ["car", "pedestrian"]  # data/dataset.py:52-63
GEN4_CLASSES = ["pedestrian", "two wheeler", "car", "truck", "bus", "traffic sign", "traffic light"]


class SyntheticLoader:
    """Stands in for ``Loader(propheseeDataset)``: yields ``(imgs, targets, file_names, time_stamps)`` batches
    (data/loader.py:26-32) from GPU-encoded synthetic streams; ``len()`` = batches per epoch."""

    def __init__(self, settings, n_batches, seed, with_track=False):
        from . import e2e
        self.settings, self.n_batches, self.with_track = settings, n_batches, with_track
        self.bins = int(settings.event_volume_bins)
        if 2 * self.bins != 16:
            raise NotImplementedError("the synthetic loader encodes TAF K=8 (16 channels): use --event_volume_bins 8")
        self.src = e2e.SyntheticTafSource(settings.batch_size, seed=seed, events_per_window=20_000)
        self.labels = self.src.labels(settings.batch_size)

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        B = self.settings.batch_size
        idx = list(range(B))
        for i in range(self.n_batches):
            imgs = self.src.encode_batch(idx)
            if self.with_track:  # evaluation labels carry (t, confidence, track) too, data/dataset.py:211-217
                lab = torch.zeros((B, 80, 8), dtype=torch.float64, device=imgs.device)
                lab[..., :5] = self.labels[..., [1, 2, 3, 4, 0]]
                lab[..., 5] = 1_000_000 + 50_000 * i
                lab[..., 6] = (self.labels[..., 3] > 0).double()
                targets = lab
            else:
                targets = self.labels
            yield imgs, targets, [f"synthetic_{i}_{j}" for j in range(B)], [1_000_000 + 50_000 * i] * B


class basicExp:
    """The parts of core/exp.py:44-391 the yolox recipes use; ``buildBackbone`` comes from the subclass."""

    def __init__(self, settings):
        self.settings = settings
        self.nr_input_channels = int(2 * self.settings.event_volume_bins)
        self.input_layer = Focus
        self.synthetic_batches = int(os.environ.get("FRLW_SYNTHETIC_BATCHES", "4"))  # batches per synthetic epoch
        self.metric_fn = None  # inject evaluate_detection (pycocotools) to get mAP out of .test()

    # ---- data -------------------------------------------------------------------------------------------
    def _classes(self):
        return GEN1_CLASSES if self.settings.dataset_name == "gen1" else GEN4_CLASSES

    def _dataset(self, mode, augment, clipping):
        """core/exp.py:60-84: ``propheseeDataset`` over the representation files the ``generate_*.py`` commands wrote."""
        from .dataset import propheseeDataset
        s = self.settings
        return propheseeDataset(s.bbox_path, s.data_path, s.dataset_name, s.input_img_size, s.img_size, s.event_volume_bins,
                                s.infer_time, s.train_memory_steps, mode, augment, clipping)

    def _disk_loaders(self, test_only=False):
        """core/exp.py:56-124 (and :401-466 / :593-660 for the TAF datasets): datasets + ``Loader``s from ``--bbox_path`` /
        ``--data_path``; the training split goes through a DistributedSampler like the reference's."""
        from .dataset import Loader
        s = self.settings
        if s.bbox_path is None or s.data_path is None:
            raise ValueError("--bbox_path and --data_path come together (omit both for the synthetic GPU-encoded run)")
        if test_only:
            val = self._dataset("test", False, False)
            self.object_classes = val.object_classes
            self.val_loader = Loader(val, batch_size=s.batch_size, device=s.gpu_device, num_workers=s.num_cpu_workers,
                                     pin_memory=False, shuffle=False)
        else:
            train = self._dataset("train", s.augment, s.clipping if self._clip_train else False)
            self.object_classes = train.object_classes
            val = self._dataset("val", False, False)
            sampler = torch.utils.data.distributed.DistributedSampler(train) if dist.is_initialized() else None
            self.train_loader = Loader(train, batch_size=s.batch_size, device=s.gpu_device, num_workers=s.num_cpu_workers,
                                       pin_memory=True, sampler=sampler)
            vs = torch.utils.data.distributed.DistributedSampler(val) if (dist.is_initialized() and self._shard_val) else None
            self.val_loader = Loader(val, batch_size=s.batch_size, device=s.gpu_device, num_workers=s.num_cpu_workers,
                                     pin_memory=True, shuffle=False, sampler=vs)
            print(f"train_loader_len: {len(self.train_loader)}, test_loader_len: {len(self.val_loader)}")
            self.nr_train_epochs = len(self.train_loader)
        if test_only:
            print(f"test_loader_len: {len(self.val_loader)}")
        self.nr_val_epochs = len(self.val_loader)
        self.ori_width, self.ori_height = val.width, val.height

    _clip_train = True   # basicExp passes settings.clipping to the training split (core/exp.py:70), the TAF exps False (:414)
    _shard_val = True    # basicExp gives the validation split a DistributedSampler (core/exp.py:87), the TAF exps do not (:430-435)

    def createDatasets(self):
        if not self.settings.synthetic:
            return self._disk_loaders()
        self.object_classes = self._classes()
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.train_loader = SyntheticLoader(self.settings, self.synthetic_batches, seed=1005 + 1000 * rank)
        self.val_loader = SyntheticLoader(self.settings, max(1, self.synthetic_batches // 2), seed=2005 + 1000 * rank, with_track=True)
        self.nr_train_epochs, self.nr_val_epochs = len(self.train_loader), len(self.val_loader)
        self.ori_width, self.ori_height = (304, 240) if self.settings.dataset_name == "gen1" else (1280, 720)

    def createDatasetsTest(self):
        if not self.settings.synthetic:
            return self._disk_loaders(test_only=True)
        self.object_classes = self._classes()
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.val_loader = SyntheticLoader(self.settings, self.synthetic_batches, seed=2005 + 1000 * rank, with_track=True)
        self.nr_val_epochs = len(self.val_loader)
        self.ori_width, self.ori_height = (304, 240) if self.settings.dataset_name == "gen1" else (1280, 720)

    # ---- model ------------------------------------------------------------------------------------------
    def configModel(self):
        self.in_channels = [256, 256, 256]
        self.out_features = ["dark3", "dark4", "dark5"]
        self.strides = [8, 16, 32]
        self.depth = 0.33

    def buildBackbone(self):
        raise NotImplementedError("the AED backbone (Darknet-21) of the basic / taf recipes is out of scope; use yolox")

    def buildNeck(self):
        self.neck = YOLOPAFPN(self.depth, in_features=self.out_features, in_channels=self.in_channels, act="silu")

    def buildMemory(self):
        self.memory = None

    def buildHead(self):
        radius = 5 if self.settings.dataset_name == "gen1" else 2.5
        self.head = YOLOXHead(len(self.object_classes), in_channels=self.in_channels, act="silu", strides=self.strides,
                              radius=radius)

    def buildModel(self):
        net = model(self.backbone, self.neck, self.memory, self.head)
        print(f"{sum(p.numel() for p in net.parameters()):,} total parameters.")
        if torch.cuda.is_available():
            net = net.cuda()
        ids = [self.settings.local_rank] if torch.cuda.is_available() else None
        # core/exp.py:391; bucket view + static graph: the gradient all-reduce reuses its buckets (dist.ddp_kwargs)
        from .dist import agree_fast_path, ddp_kwargs
        agree_fast_path()  # the encoders' fast paths run on every rank or on none (one all-reduce)
        self.model = DistributedDataParallel(net, device_ids=ids, broadcast_buffers=False, **ddp_kwargs())

    # ---- optimisation -----------------------------------------------------------------------------------
    def getOptimizer(self, lr):
        params = filter(lambda p: p.requires_grad, self.model.parameters())
        return torch.optim.Adam(params, lr=self.settings.warmup_lr if self.settings.warmup_epochs > 0 else lr)

    def getLearningRate(self):
        return self.optimizer.param_groups[0]["lr"]

    def get_lr_scheduler(self, lr, iters_per_epoch):
        return LRScheduler("yoloxwarmcos", lr, iters_per_epoch, self.settings.max_epoch,
                           warmup_epochs=self.settings.warmup_epochs, warmup_lr_start=self.settings.warmup_lr,
                           no_aug_epochs=0, min_lr_ratio=self.settings.min_lr_ratio)

    def update_lr(self, i_batch):
        lr = self.scheduler.update_lr(self.epoch_step * self.nr_train_epochs + i_batch + 1)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        return lr

    # ---- checkpoints (core/exp.py:155-210) ------------------------------------------------------------------
    def _map_location(self):
        return {"cuda:0": f"cuda:{self.settings.local_rank}"} if torch.cuda.is_available() else "cpu"

    def loadCheckpoint(self, filename, with_optimizer=True):
        if not os.path.isfile(filename):
            raise Exception(f"=> no checkpoint found at '{filename}'")
        print(f"=> loading checkpoint '{filename}'")
        checkpoint = torch.load(filename, map_location=self._map_location(), weights_only=False)
        self.epoch_step = checkpoint["epoch"] + 1
        self.model.load_state_dict(checkpoint["state_dict"])
        if with_optimizer:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        print(f"=> loaded checkpoint '{filename}' (epoch {checkpoint['epoch']})")

    def loadCheckpointTest(self, filename):
        self.loadCheckpoint(filename, with_optimizer=False)

    def saveCheckpoint(self, name):
        base = os.path.join(self.settings.ckpt_dir, name)
        print("save to ", base + ".pth")
        torch.save({"state_dict": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
                    "epoch": self.epoch_step}, base + ".pth")
        torch.save({"state_dict": self.backbone.state_dict()}, base + "_backbone.pth")
        torch.save({"state_dict": self.neck.state_dict()}, base + "_neck.pth")

    # ---- loops ------------------------------------------------------------------------------------------
    def _build_all(self):
        self.configModel()
        self.buildBackbone()
        self.buildNeck()
        self.buildMemory()
        self.buildHead()
        self.buildModel()

    def _evaluator(self, rec=None):
        s = self.settings
        return evaluator(self.object_classes, s.batch_size, s.infer_time, self.ori_width, self.ori_height, s.img_size[1],
                         s.img_size[0], s.dataset_name, rec)

    def train(self):
        """core/exp.py:212-260: epochs of trainEpoch + validationEpoch, ``last_epoch`` / ``best_epoch`` checkpoints."""
        self.createDatasets()
        self._build_all()
        lr = self.settings.init_lr
        self.epoch_step = 0
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        self.scaler = torch.amp.GradScaler(dev, enabled=True)
        self.optimizer = self.getOptimizer(lr)
        self.scheduler = self.get_lr_scheduler(lr, self.nr_train_epochs)
        self.max_score = 0.0
        if self.settings.resume_training:
            self.loadCheckpoint(self.settings.resume_ckpt_file)
        stop = int(os.environ.get("FRLW_MAX_EPOCHS", self.settings.max_epoch_to_stop))
        while self.epoch_step < stop:
            self.trainEpoch()
            s = self.settings
            if (not s.reduce_evaluate) or ((self.epoch_step > 0) and (self.epoch_step % ceil(s.max_epoch_to_stop / 10) == 0)
                                           or self.epoch_step >= s.max_epoch_to_stop / 5 * 3):
                self.validationEpoch(self._evaluator())
            self.epoch_step += 1

    def test(self):
        """core/exp.py:262-281."""
        self.createDatasetsTest()
        self._build_all()
        self.epoch_step = 0
        if self.settings.resume_ckpt_file is not None:
            self.loadCheckpointTest(self.settings.resume_ckpt_file)
        rec = recorder(self.settings.log_dir) if self.settings.record else None
        return self.testingEpoch(self._evaluator(rec))

    def trainEpoch(self):
        """core/exp.py:283-315: zero_grad, forward, ``scaler.scale(loss).backward()``, ``optimizer.step()`` (NOT
        ``scaler.step``: the 65536x scale reaches Adam), LR update, rank 0 saves ``last_epoch``."""
        self.model = self.model.train()
        train_loss = 0.0
        for i_batch, (imgs, targets, file_names, time_stamps) in enumerate(self.train_loader):
            self.optimizer.zero_grad()
            loss = self.model(imgs, targets, file_names, time_stamps)
            self.scaler.scale(loss).backward()
            self.optimizer.step()
            lr = self.update_lr(i_batch)
            train_loss += float(loss.detach().cpu())
            if self.settings.local_rank == 0:
                print(f"Epoch{self.epoch_step}, iter{i_batch}/{self.nr_train_epochs}, trainloss{float(loss):.6f}, lr{lr:.3e}")
        self.last_train_loss = train_loss / max(self.nr_train_epochs, 1)
        if self.settings.local_rank == 0:
            self.saveCheckpoint("last_epoch")

    def validationEpoch(self, result):
        eval_results = self.testingEpoch(result)
        score = eval_results[0] if isinstance(eval_results, (list, tuple)) else 0.0  # mAP needs an injected metric_fn
        if score > self.max_score or not os.path.exists(os.path.join(self.settings.ckpt_dir, "best_epoch.pth")):
            self.max_score = max(self.max_score, score)
            if self.settings.local_rank == 0:
                self.saveCheckpoint("best_epoch")
        if self.settings.local_rank == 0:
            print(f"Epoch {self.epoch_step}: best score {self.max_score}")

    def testingEpoch(self, result):
        self.model = self.model.eval()
        for imgs, targets, file_names, time_stamps in self.val_loader:
            with torch.no_grad():
                result = self.model(imgs, targets, file_names, time_stamps, evaluator=result)
        return result.evaluate(self.metric_fn)


class yolox(basicExp):
    """core/exp.py:580-586."""

    def buildBackbone(self):
        self.backbone = CSPDarknet(self.nr_input_channels, 0.33, 0.5, stem=self.input_layer)

    def configModel(self):
        super().configModel()
        self.in_channels = [128, 256, 512]


class yoloxtafBFM(yolox):
    """core/exp.py:588-591: the BFM stem in front of the same detector."""

    def __init__(self, settings):
        super().__init__(settings)
        from .yolox.bfm import Temporal_Active_Focus_connect
        self.input_layer = Temporal_Active_Focus_connect

    _clip_train = False
    _shard_val = False

    def _dataset(self, mode, augment, clipping):
        """core/exp.py:593-660: ``propheseeTafDataset`` (bins4 + bins8 files of ``generate_taf.py``)."""
        from .dataset import propheseeTafDataset
        s = self.settings
        return propheseeTafDataset(s.bbox_path, s.data_path, s.dataset_name, s.input_img_size, s.img_size, s.infer_time,
                                   s.event_volume_bins, mode, augment, clipping)


EXPERIMENTS = {"yolox": yolox, "yolox_taf_bfm": yoloxtafBFM}
OTHER_RECIPES = ("basic", "taf", "taf_bfm", "yolov3", "yolov3_taf_bfm")  # AED / YOLOv3 detectors: out of scope
