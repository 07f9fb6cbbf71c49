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

One rank without DDP can replay the whole step -- forward, SimOTA + losses, backward, Adam: ~1 200 launches, none of them
data-dependent on the host -- as ONE HIP graph (``Trainer(..., graph=True)`` or ``capture()``): the step is launch-bound
in its loss and gradient-accumulation stretches and ends in the reference's ``loss.cpu()``, which drains the queue
(15-20 % of the step idle in an eager trace, tools/train_gaps.sh); the replay leaves no gaps.

Several ranks: ``Trainer(ddp=True, graph=True)`` replays the step as TWO graphs around one eagerly launched all-reduce of a
flat gradient buffer -- forward + backward + gather | all-reduce | Adam -- without the DistributedDataParallel wrapper (its
reducer cannot be captured); ``ddp=True`` alone keeps the wrapper and eager launches.
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
                 local_rank=None, ddp=False, comm_hook=None, graph=False):
        self.lr0, self.per_gpu_batch = init_lr(global_batch, nodes)
        self._graph = None          # (HIP graph, static images, static labels, static loss) once captured
        self._pinned = False
        # ddp + graph on a GPU: the step is TWO HIP graphs around ONE eager all-reduce of a flat gradient buffer (see capture):
        # no DistributedDataParallel wrapper (its reducer cannot be captured), the same averaged gradient
        # (ddp="flat" asks for the wrapper-free exchange whatever the launch form: eager on a CPU, or without graph=True)
        self._flat_ddp = ddp == "flat" or (bool(ddp) and bool(graph) and next(model.parameters()).is_cuda)
        self._flat = None           # the flat gradient buffer of that form (every p.grad is a view of it after the capture)
        self._world = 1
        self._want_graph = bool(graph) and (not ddp or self._flat_ddp) and next(model.parameters()).is_cuda
        self.model = model
        self.comm_hook = None
        if self._flat_ddp:
            import torch.distributed as tdist
            from .dist import broadcast_module_state
            if not tdist.is_initialized():
                raise RuntimeError("Trainer(ddp=True, graph=True) needs an initialised process group (dist.init_from_env)")
            self._world = tdist.get_world_size()
            broadcast_module_state(model)  # rank 0's parameters: what DistributedDataParallel's constructor does (buffers stay per rank)
        elif ddp:
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
        self._want_graph = self._want_graph and fused
        lr_start = 0.0 if warmup_epochs > 0 else self.lr0
        if self._want_graph:  # a graph replays device work only: the step counter and the learning rate live on the device
            self.optimizer = torch.optim.Adam(params, lr=torch.tensor(lr_start, dtype=torch.float32, device=params[0].device),
                                              fused=True, capturable=True)
        else:
            self.optimizer = torch.optim.Adam(params, lr=lr_start, **({"fused": True} if fused else {}))
        self.scheduler = LRScheduler("yoloxwarmcos", self.lr0, iters_per_epoch, max_epoch, warmup_epochs=warmup_epochs,
                                     warmup_lr_start=0.0, no_aug_epochs=0, min_lr_ratio=0.05)
        dev = "cuda" if next(model.parameters()).is_cuda else "cpu"
        self.scaler = torch.amp.GradScaler(dev, enabled=True)  # core/exp.py:227
        self.iters_per_epoch = iters_per_epoch
        self.epoch_step = 0

    def _set_lr(self, lr):
        for g in self.optimizer.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr

    def _forward(self, imgs, targets):
        if imgs.is_cuda:  # the GEMM operands of all BaseConv weights in ONE launch (the layers skip their own layout kernels)
            from .yolox import train_ops
            train_ops.layout_all_weights(self.model.module if hasattr(self.model, "module") else self.model)
        return self.model(imgs, targets, None, None)

    def _eager_step(self, imgs, targets):
        self.optimizer.zero_grad()
        loss = self._forward(imgs, targets)
        self.scaler.scale(loss).backward()
        if self._flat_ddp:  # (a batch the captured graphs were not made for, e.g. the short last batch of an epoch)
            self._average_gradients()
        self.optimizer.step()  # NOT scaler.step: the 65536x scale reaches Adam (reference behaviour)
        return loss

    def _average_gradients(self):
        """The gradient exchange of the two-graph form, launched eagerly: divide by the world size, SUM over the ranks."""
        import torch.distributed as tdist
        grads = [p.grad for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads]).div_(self._world)
        tdist.all_reduce(flat)
        torch._foreach_copy_(grads, [v.view_as(g) for g, v in zip(grads, flat.split([g.numel() for g in grads]))])

    def capture(self, imgs, targets, warmup=3):
        """Capture one whole step for batches shaped like (imgs, targets) into a HIP graph.  The ``warmup`` steps torch's capture
        rules ask for (on a side stream) run on the given batch but DO NOT TRAIN: parameters, BatchNorm buffers (running
        statistics, ``num_batches_tracked``) and the optimizer state (moments, step counters) are snapshotted before and
        copied back IN PLACE after the capture (addresses stay fixed for the graph), so that a graphed trainer follows the
        same trajectory as the eager one -- and as core/exp.py:292-303 -- on the same data stream.  Returns the number of
        steps that trained: 0.  The gradients are allocated inside the graph's memory pool, so ``zero_grad`` is part of the
        replay (set_to_none before capture)."""
        if not self._want_graph:
            raise RuntimeError("graph capture needs Trainer(graph=True) and parameters on the GPU")
        self.model.train()
        x, lab = imgs.clone(), targets.clone()
        snap_model = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        snap_opt = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                    for p, st in self.optimizer.state.items()}
        snap_lr = [g["lr"].clone() if torch.is_tensor(g["lr"]) else g["lr"] for g in self.optimizer.param_groups]
        if self._flat_ddp:
            warmup = max(int(warmup), 1)  # (the flat gradient buffer is laid out from the gradients a backward has left)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.optimizer.zero_grad(set_to_none=True)
                loss = self._forward(x, lab)
                self.scaler.scale(loss).backward()
                self.optimizer.step()
                del loss  # (nothing of this iteration's autograd graph may live into the capture)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        if self._flat_ddp:
            # Several ranks: graph A = forward + loss + backward + the gradients gathered into ONE flat buffer and divided by
            # the world size (DDP's default: divide, then SUM); the all-reduce of that buffer is launched eagerly between the
            # replays (one large message: xGMI rings are per-link bound); graph B = Adam on views of the buffer.  The warm-up
            # steps above ran WITHOUT any collective on every rank alike (they do not train), so no rank waits for another
            # inside the capture.
            params = [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
            sizes = [p.numel() for p in params]
            self._flat = torch.empty(sum(sizes), dtype=params[0].dtype, device=params[0].device)
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                loss = self._forward(x, lab)
                self.scaler.scale(loss).backward()
                torch.cat([p.grad.reshape(-1) for p in params], out=self._flat)
                self._flat.div_(self._world)
            for p, v in zip(params, self._flat.split(sizes)):
                p.grad = v.view_as(p)
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b):
                self.optimizer.step()
            self._graph = (graph, x, lab, loss, graph_b)
        else:
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                loss = self._forward(x, lab)
                self.scaler.scale(loss).backward()
                self.optimizer.step()
            self._graph = (graph, x, lab, loss)
        # undo the warm-up: everything the steps above moved goes back to its snapshot, in place
        with torch.no_grad():
            for k, v in self.model.state_dict().items():
                v.copy_(snap_model[k])
            for p, st in self.optimizer.state.items():
                old = snap_opt.get(p)
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if old is not None and k in old:
                            v.copy_(old[k])
                        else:
                            v.zero_()  # a fresh Adam starts from zero moments and step 0
            for g, lr in zip(self.optimizer.param_groups, snap_lr):
                if torch.is_tensor(g["lr"]):
                    g["lr"].copy_(lr)
                else:
                    g["lr"] = lr
        torch.cuda.synchronize()
        from . import _pins
        if not self._pinned:
            _pins.pin()  # module-level scratch / operand caches the graph points into must outlive it
            self._pinned = True
        return 0

    def __del__(self):
        try:
            if getattr(self, "_pinned", False):
                from . import _pins
                self._graph = None
                _pins.unpin()
        except Exception:
            pass

    def input_buffers(self):
        """(images, labels) the captured graph reads: a data pipeline that writes its batch into these (and passes them to
        ``train_step``) saves the device-to-device copy of the batch (84 MB at batch 64).  None before the capture."""
        return None if self._graph is None else (self._graph[1], self._graph[2])

    def train_step(self, imgs, targets, i_batch=0, sync=True, after_launch=None):
        """core/exp.py:292-303 for one batch; returns (loss as a Python float, lr).  ``sync=False`` returns the loss as a
        device tensor instead (the caller reads it when it logs: no queue drain per step).  ``after_launch``: called once
        the step's device work is queued and before the loss is read back (the place to start the next batch's encode on
        another stream, ``e2e.EncodeAhead``)."""
        if not self.model.training:
            self.model.train()
        if self._want_graph and self._graph is None:
            self.capture(imgs, targets, warmup=3)
        if self._graph is not None and imgs.shape == self._graph[1].shape and targets.shape == self._graph[2].shape:
            graph, x, lab, loss = self._graph[:4]
            if imgs.data_ptr() != x.data_ptr():  # (a producer that writes straight into input_buffers() skips this copy)
                x.copy_(imgs, non_blocking=True)
            if targets.data_ptr() != lab.data_ptr():
                lab.copy_(targets, non_blocking=True)
            graph.replay()
            if self._flat_ddp:  # the one collective of the step, between the two graphs
                import torch.distributed as tdist
                tdist.all_reduce(self._flat)
                self._graph[4].replay()
            # the replay moved the parameters without telling autograd: bump their version counters so that every cache
            # keyed on them (weight layouts, the folded inference engine) sees the change
            torch.autograd.graph.increment_version([p for g in self.optimizer.param_groups for p in g["params"]])
        else:
            loss = self._eager_step(imgs, targets)
        lr = self.scheduler.update_lr(self.epoch_step * self.iters_per_epoch + i_batch + 1)
        self._set_lr(lr)
        if after_launch is not None:
            after_launch()
        if not sync:
            return loss.detach().clone() if self._graph is not None else loss.detach(), lr
        return float(loss.detach().cpu()), lr
