"""Trainer step of the ``yolox`` experiment (reference: core/exp.py:126-153,227-231,283-315,386-391,
settings.py:41,80-94, core/yolox/utils/lr_scheduler.py:121-148).

Quirks kept on purpose, because they change the optimisation trajectory:
  * ``GradScaler(enabled=True)`` scales the loss by 65536 but ``optimizer.step()`` is called directly and
    ``scaler.update()`` never, so Adam sees 65536x gradients (core/exp.py:227,295,299); no autocast anywhere;
  * the loss is float64 (labels are float64, data/dataset.py:216);
  * per-GPU batch = batch_size / nodes, lr = 0.0133333 / 64 * per_gpu_batch * nodes (settings.py:41,87);
  * DistributedDataParallel(broadcast_buffers=False): BatchNorm statistics stay per rank (core/exp.py:391).

Every training-mode ``BaseConv`` on a ROCm tensor runs forward and backward in the gfx950 kernels of
csrc/train_ops.hip (``yolox/train_ops.py``: fp32 MFMA convolution, data and weight gradient, BatchNorm + SiLU), SimOTA
in csrc/simota.hip; torch autograd only strings the blocks together.  The gradient all-reduce is DDP's bucketed RCCL
all-reduce, or the direct reduce-scatter + all-gather hook of ``dist.py``.
"""
from __future__ import annotations

import math

import torch


def yolox_warm_cos_lr(lr, min_lr_ratio, total_iters, warmup_total_iters, warmup_lr_start, no_aug_iter, iters):
    """Quadratic warm-up from warmup_lr_start, cosine to lr * min_lr_ratio (lr_scheduler.py:121-148)."""
    min_lr = lr * min_lr_ratio
    if iters <= warmup_total_iters:
        return (lr - warmup_lr_start) * pow(iters / float(warmup_total_iters), 2) + warmup_lr_start
    if iters >= total_iters - no_aug_iter:
        return min_lr
    return min_lr + 0.5 * (lr - min_lr) * (1.0 + math.cos(
        math.pi * (iters - warmup_total_iters) / (total_iters - warmup_total_iters - no_aug_iter)))


class LRScheduler:
    """``LRScheduler("yoloxwarmcos", lr, iters_per_epoch, max_epoch, warmup_epochs=5, warmup_lr_start=0,
    no_aug_epochs=0, min_lr_ratio=0.05)`` as core/exp.py:134-147 builds it."""

    def __init__(self, name, lr, iters_per_epoch, total_epochs, warmup_epochs=5, warmup_lr_start=0.0,
                 no_aug_epochs=0, min_lr_ratio=0.05):
        if name != "yoloxwarmcos":
            raise ValueError("only the yoloxwarmcos schedule is on the hot path")
        self.lr, self.min_lr_ratio = lr, min_lr_ratio
        self.total_iters = iters_per_epoch * total_epochs
        self.warmup_total_iters = iters_per_epoch * warmup_epochs
        self.warmup_lr_start = warmup_lr_start
        self.no_aug_iters = iters_per_epoch * no_aug_epochs

    def update_lr(self, iters):
        return yolox_warm_cos_lr(self.lr, self.min_lr_ratio, self.total_iters, self.warmup_total_iters,
                                 self.warmup_lr_start, self.no_aug_iters, iters)


def init_lr(global_batch, nodes):
    """settings.py:41,87: per-GPU batch = int(global / nodes); lr = 0.0133333 / 64 * per_gpu * nodes."""
    per_gpu = int(global_batch / nodes)
    return 0.0133333 / 64.0 * per_gpu * nodes, per_gpu


class Trainer:
    """One rank of the reference's training loop around an already built ``model``."""

    def __init__(self, model, global_batch=64, nodes=1, iters_per_epoch=100, max_epoch=50, warmup_epochs=5,
                 local_rank=None, ddp=False, comm_hook=None):
        self.lr0, self.per_gpu_batch = init_lr(global_batch, nodes)
        self.model = model
        self.comm_hook = None
        if ddp:
            from torch.nn.parallel import DistributedDataParallel
            ids = [local_rank] if (local_rank is not None and next(model.parameters()).is_cuda) else None
            from .dist import ddp_kwargs, install_comm_hook
            self.model = DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False,  # core/exp.py:391
                                                 **ddp_kwargs())
            self.comm_hook = install_comm_hook(self.model, comm_hook)
        params = [p for p in self.model.parameters() if p.requires_grad]
        # core/exp.py:126-128: torch.optim.Adam with its defaults.  On the GPU the update of all ~300 tensors runs as ONE
        # multi-tensor kernel per chunk (fused=True: the same arithmetic per element as the default implementation, which
        # issues eight launches per chunk -- 0.6 ms of a 33 ms step)
        fused = bool(params) and all(p.is_cuda for p in params)
        self.optimizer = torch.optim.Adam(params, lr=0.0 if warmup_epochs > 0 else self.lr0, **({"fused": True} if fused else {}))
        self.scheduler = LRScheduler("yoloxwarmcos", self.lr0, iters_per_epoch, max_epoch, warmup_epochs=warmup_epochs,
                                     warmup_lr_start=0.0, no_aug_epochs=0, min_lr_ratio=0.05)
        dev = "cuda" if next(model.parameters()).is_cuda else "cpu"
        self.scaler = torch.amp.GradScaler(dev, enabled=True)  # core/exp.py:227
        self.iters_per_epoch = iters_per_epoch
        self.epoch_step = 0

    def train_step(self, imgs, targets, i_batch=0):
        """core/exp.py:292-303 for one batch; returns (loss as a Python float, lr)."""
        self.model.train()
        self.optimizer.zero_grad()
        loss = self.model(imgs, targets, None, None)
        self.scaler.scale(loss).backward()
        self.optimizer.step()  # NOT scaler.step: the 65536x scale reaches Adam (reference behaviour)
        lr = self.scheduler.update_lr(self.epoch_step * self.iters_per_epoch + i_batch + 1)
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        return float(loss.detach().cpu()), lr
