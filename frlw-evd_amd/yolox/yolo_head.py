"""Decoupled YOLOX head: eval forward, decode and NMS (reference: core/yolox/models/yolo_head.py:23-303).

Per level: 1x1 stem -> 256 (fixed, no width multiplier), cls and reg towers of two 3x3 256->256
BaseConv, biased 1x1 predictions (num_classes / 4 / 1).  Eval: ``cat[reg, sigmoid(obj), sigmoid(cls)]``,
flatten, permute -> (B, A, 5 + nc), then :meth:`decode_outputs`.

Training: the raw per-level outputs go to :func:`frlw_evd_amd.yolox.losses.yolox_losses` (SimOTA assignment +
losses, yolo_head.py:305-707) and the 6-tuple (loss, 5*iou, obj, cls, l1, fg/gt) is returned.
"""
import math

import torch
import torch.nn as nn

from .network_blocks import BaseConv


def nms_reference(xyxy, scores, iou_threshold):
    """Documented semantics of ``torchvision.ops.nms`` (absent in this image; yolo_head.py:281): visit boxes
    by descending score, keep a box unless an already kept one overlaps it with IoU > threshold,
    IoU = inter / (area_a + area_b - inter) on xyxy without +1; returns kept indices in visiting order."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = xyxy[order]
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    suppressed = torch.zeros(len(b), dtype=torch.bool, device=b.device)
    keep = []
    for i in range(len(b)):
        if suppressed[i]:
            continue
        keep.append(i)
        xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
        inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
        ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > iou_threshold
    return order[torch.tensor(keep, dtype=torch.long, device=b.device)]


class YOLOXHead(nn.Module):
    def __init__(self, num_classes, strides=(16, 32, 64), in_channels=(256, 512, 1024), act="silu",
                 depthwise=False, radius=2.5, seq_nms=False):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise convolutions are outside the hot path (SURVEY.md section 8)")
        self.n_anchors = 1
        self.num_classes = num_classes
        self.decode_in_inference = True
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        self.cls_preds = nn.ModuleList()
        self.reg_preds = nn.ModuleList()
        self.obj_preds = nn.ModuleList()
        self.stems = nn.ModuleList()
        for c in in_channels:
            self.stems.append(BaseConv(int(c), 256, 1, 1, act=act))
            self.cls_convs.append(nn.Sequential(BaseConv(256, 256, 3, 1, act=act), BaseConv(256, 256, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(BaseConv(256, 256, 3, 1, act=act), BaseConv(256, 256, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(256, self.n_anchors * num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(256, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(256, self.n_anchors, 1, 1, 0))
        self.use_l1 = False
        self.strides = list(strides)
        self.seq_nms = seq_nms  # sequence NMS is disabled in every shipped experiment
        self.radius = radius
        self.obj_threshold = 0.3  # yolo_head.py:276
        self.nms_threshold = 0.6  # yolo_head.py:281
        self.hw = None

    def clean_seqnms(self):  # called by model.forward when seq_nms is set (core/model.py:65-66)
        pass

    def initialize_biases(self, prior_prob):
        for preds in (self.cls_preds, self.obj_preds):
            for conv in preds:
                conv.bias.data.fill_(-math.log((1 - prior_prob) / prior_prob))

    def raw_outputs(self, xin):
        """Per level ``cat[reg, sigmoid(obj), sigmoid(cls)]`` (yolo_head.py:170-213, eval branch)."""
        outs = []
        for k, x in enumerate(xin):
            x = self.stems[k](x)
            cls_feat = self.cls_convs[k](x)
            reg_feat = self.reg_convs[k](x)
            outs.append(torch.cat([self.reg_preds[k](reg_feat), self.obj_preds[k](reg_feat).sigmoid(),
                                   self.cls_preds[k](cls_feat).sigmoid()], 1))
        return outs

    def train_outputs(self, xin):
        """Per level ``cat[reg, obj, cls]`` without sigmoid (yolo_head.py:186, training branch)."""
        outs = []
        from . import train_ops
        for k, x in enumerate(xin):
            x = self.stems[k](x)
            if x.is_cuda and len(self.cls_convs[k]) == 2 and len(self.reg_convs[k]) == 2 \
                    and train_ops.pair_eligible(x, self.cls_convs[k][0], self.reg_convs[k][0]):
                # the two towers read the same x: their first convolutions as one autograd node (dx summed in an epilogue)
                c0, r0 = train_ops.pair_train(x, self.cls_convs[k][0], self.reg_convs[k][0])
                cls_feat, reg_feat = self.cls_convs[k][1](c0), self.reg_convs[k][1](r0)
            else:
                cls_feat = self.cls_convs[k](x)
                reg_feat = self.reg_convs[k](x)
            if train_ops.pred_eligible(reg_feat, cls_feat, self.reg_preds[k], self.obj_preds[k], self.cls_preds[k]):
                outs.append(train_ops.pred_level(reg_feat, cls_feat, self.reg_preds[k], self.obj_preds[k], self.cls_preds[k]))
                continue
            outs.append(torch.cat([self._pred(self.reg_preds[k], reg_feat), self._pred(self.obj_preds[k], reg_feat),
                                   self._pred(self.cls_preds[k], cls_feat)], 1))
        return outs

    @staticmethod
    def _pred(conv, feat):
        """The biased 1x1 prediction convolution.  On channels_last tensors (what the native BaseConv kernels produce)
        it is a plain (B*H*W, C) x (C, n) GEMM on the NHWC view -- MIOpen would pick a naive NHWC weight-gradient
        kernel for these 1-, 2- and 4-channel outputs (5.5 ms each, measured)."""
        if feat.is_cuda and feat.dim() == 4 and feat.is_contiguous(memory_format=torch.channels_last) \
                and not feat.is_contiguous():
            B, C, H, W = feat.shape
            out = torch.nn.functional.linear(feat.permute(0, 2, 3, 1).reshape(B * H * W, C),
                                             conv.weight.view(conv.out_channels, C), conv.bias)
            return out.view(B, H, W, conv.out_channels).permute(0, 3, 1, 2)
        return conv(feat)

    def forward(self, xin, labels=None, imgs=None):
        if self.training:
            from .losses import yolox_losses
            return yolox_losses(self.train_outputs(xin), self.strides, labels, self.num_classes, self.radius)
        outs = self.raw_outputs(xin)
        self.hw = [o.shape[-2:] for o in outs]
        outputs = torch.cat([o.flatten(start_dim=2) for o in outs], dim=2).permute(0, 2, 1)
        if self.decode_in_inference:
            return self.decode_outputs(outputs, dtype=xin[0].type())
        return outputs

    def grids_and_strides(self, device, dtype=torch.float32):
        grids, strides = [], []
        for (h, w), s in zip(self.hw, self.strides):
            yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
            g = torch.stack((xv, yv), 2).view(1, -1, 2)  # (x, y) order, yolo_head.py:262-263
            grids.append(g)
            strides.append(torch.full((1, g.shape[1], 1), s))
        return torch.cat(grids, 1).to(device=device, dtype=dtype), torch.cat(strides, 1).to(device=device, dtype=dtype)

    def decode_boxes(self, outputs):
        """xy = (xy + grid) * stride, wh = square(wh) * stride -- not exp (yolo_head.py:271-272)."""
        grids, strides = self.grids_and_strides(outputs.device, outputs.dtype)
        outputs = outputs.clone()
        outputs[..., :2] = (outputs[..., :2] + grids) * strides
        outputs[..., 2:4] = torch.square(outputs[..., 2:4]) * strides
        return outputs

    def decode_outputs(self, outputs, dtype=None):
        """-> list of (n_i, 6) [cx, cy, w, h, argmax cls, obj * max cls]; (1, 6) zeros when nothing passes
        obj > 0.3 (yolo_head.py:258-303)."""
        outputs = self.decode_boxes(outputs)
        nc = self.num_classes
        result = []
        for output in outputs:
            output = output[output[:, 4] > self.obj_threshold]
            if len(output) == 0:
                output = torch.zeros((1, 8), device=output.device)
            else:
                xyxy = torch.cat([output[:, 0:1] - output[:, 2:3] / 2, output[:, 1:2] - output[:, 3:4] / 2,
                                  output[:, 0:1] + output[:, 2:3] / 2, output[:, 1:2] + output[:, 3:4] / 2], dim=-1)
                output = output[nms_reference(xyxy, output[:, 4], self.nms_threshold)]
            result.append(torch.cat([output[:, 0:4], torch.argmax(output[:, 5:5 + nc], 1)[:, None],
                                     (output[:, 4] * torch.max(output[:, 5:5 + nc], 1)[0])[:, None]], dim=1))
        return result
