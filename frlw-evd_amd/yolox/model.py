"""``model(backbone, neck, memory, head)`` wrapper (reference: core/model.py:10-70) and the ``yolox``
experiment's constructor arguments (core/exp.py:372,377-384,582)."""
import time

import torch
import torch.nn as nn

from .darknet import CSPDarknet
from .network_blocks import Focus
from .yolo_head import YOLOXHead
from .yolo_pafpn import YOLOPAFPN


class model(nn.Module):
    """``xs``: (B, C, H, W, 1, T) -- one trailing singleton is dropped here (core/model.py:40) and one by
    ``Focus`` (network_blocks.py:221); T = 1 in every shipped experiment (settings.py:45).

    Eval mode on a ROCm tensor runs the gfx950 engine (:class:`frlw_evd_amd.detector.DetectorEngine`);
    there is no silent fallback: a missing HIP library raises.
    """

    def __init__(self, backbone, neck, memory, head):
        super().__init__()
        self.backbone = backbone
        self.neck = neck
        self.memory = memory
        self.head = head
        self._engine = None
        self._engine_sig = None

    def reference_outputs(self, x):
        """Plain-PyTorch eval forward up to the pre-NMS tensor (B, A, 5 + nc); x: (B, C, H, W, 1)."""
        outs = self.head.raw_outputs(self.neck(self.backbone(x)))
        self.head.hw = [o.shape[-2:] for o in outs]
        return torch.cat([o.flatten(start_dim=2) for o in outs], dim=2).permute(0, 2, 1)

    def _weights_signature(self):
        """Identity + in-place version of every parameter and buffer the module tree holds NOW: changes on optimizer steps,
        load_state_dict(), BatchNorm running-statistics updates, .to() / .cuda() -- and when a tensor is replaced behind this
        module's back (``net.backbone.cuda()``, a new ``nn.Parameter`` assigned to a sub-module, a parametrization).  The tree is
        walked on EVERY call (about 440 tensors: ~0.2 ms beside a forward of milliseconds): a cached tensor list keeps the ids of
        the OLD objects alive and valid, so a replaced tensor would go unnoticed -- until round 4 for up to 31 forwards."""
        return tuple((id(t), t.data_ptr(), t._version) for t in self._state_tensors())

    def _state_tensors(self):
        yield from self.parameters()
        yield from self.buffers()

    def engine(self):
        """The gfx950 engine for the CURRENT weights.  The engine folds BatchNorm into the convolutions and keeps its
        own device copies, so it is rebuilt whenever a parameter or buffer has changed since it was built (the
        reference alternates train and validation epochs on one model, core/exp.py:237-258)."""
        sig = self._weights_signature()
        if self._engine is None or self._engine_sig != sig:
            from ..detector import DetectorEngine
            self._engine = DetectorEngine(self)
            self._engine_sig = sig
        return self._engine

    def drop_engine(self):
        self._engine = None
        self._engine_sig = None

    def _apply(self, fn, *a, **k):  # .to() / .cuda() / .float(): new tensors
        self.drop_engine()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.drop_engine()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        if mode:
            self.drop_engine()
        return super().train(mode)

    def __getstate__(self):  # the engine holds a ctypes handle: never pickled / deep-copied
        d = self.__dict__.copy()
        d["_engine"] = None
        d["_engine_sig"] = None
        return d

    def detect(self, x):
        """Eval forward of one time step: list of (n_i, 6) detections per image."""
        if x.is_cuda:
            return self.engine().detect(x)
        return self.head(self.neck(self.backbone(x)))

    def forward(self, xs, targets=None, filenames=None, timestamps=None, evaluator=None):
        if self.memory is not None:
            raise NotImplementedError("recurrent memory is disabled in every shipped experiment (core/exp.py:374-375)")
        outputs_tol = None
        for i in range(xs.shape[-1]):
            start = time.time()
            if i < xs.shape[-1] - 1:
                continue  # earlier time steps only feed memory / seq-NMS, both disabled
            if self.training:
                losses = self.head(self.neck(self.backbone(xs[..., i])), targets, xs[..., i])
                outputs_tol = losses[0] if outputs_tol is None else outputs_tol + losses[0]
            else:
                outputs = self.detect(xs[..., i])
                if xs.is_cuda:
                    torch.cuda.synchronize()
                infer_time = time.time() - start
                if evaluator is not None:
                    evaluator.add_result(outputs, timestamps, targets, filenames, infer_time, 0)
        if self.head.seq_nms:
            self.head.clean_seqnms()
        if self.training:
            return outputs_tol
        if evaluator is not None:
            evaluator.end_a_batch()
            return evaluator
        return outputs


def build_yolox(in_channels=10, num_classes=2, radius=5.0, stem="focus"):
    """The ``yolox`` experiment: CSPDarknet(C, 0.33, 0.5, Focus) + YOLOPAFPN(0.33, [128, 256, 512]) +
    YOLOXHead(nc, strides [8, 16, 32], in_channels [128, 256, 512]) (core/exp.py:372,377-384,580-586).
    ``stem="bfm"``: the ``yolox_taf_bfm`` experiment (core/exp.py:588-591), ``Temporal_Active_Focus_connect``."""
    chans = [128, 256, 512]
    if stem == "bfm":
        from .bfm import Temporal_Active_Focus_connect
        backbone = CSPDarknet(in_channels, 0.33, 0.5, stem=Temporal_Active_Focus_connect)
    else:
        backbone = CSPDarknet(in_channels, 0.33, 0.5, stem=Focus)
    neck = YOLOPAFPN(0.33, in_features=["dark3", "dark4", "dark5"], in_channels=chans, act="silu")
    head = YOLOXHead(num_classes, in_channels=chans, act="silu", strides=[8, 16, 32], radius=radius)
    return model(backbone, neck, None, head)


def recipe_state_dict(module, seed=1004):
    """Deterministic random weights by recipe: every tensor is drawn from its own
    PCG64 stream keyed by (seed, crc32(name)), so any model with the same parameter names and shapes gets
    the same values without shipping a 57 MB weight file.
    conv weights N(0, sqrt(2 / fan_in)); BN gamma U(0.5, 1.5), beta N(0, 0.1), running_mean N(0, 0.1),
    running_var U(0.5, 1.5); prediction biases N(0, 0.1)."""
    import zlib

    import numpy as np
    out = {}
    for name, t in module.state_dict().items():
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            v = np.zeros(shape, np.int64)
        elif name.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            v = rng.normal(0.0, 0.1, shape)
        elif ".bn." in name and name.endswith("weight"):
            v = rng.uniform(0.5, 1.5, shape)
        elif name.endswith("bias"):
            v = rng.normal(0.0, 0.1, shape)
        elif name.endswith("weight_g"):  # weight-normed convs of the BFM stem: per-output-channel gain
            v = rng.uniform(0.5, 1.5, shape)
        else:  # conv weight (Cout, Cin, kh, kw)
            fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0.0, np.sqrt(2.0 / fan_in), shape)
        out[name] = torch.from_numpy(np.asarray(v)).to(t.dtype)
    return out
