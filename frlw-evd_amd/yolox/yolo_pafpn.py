"""PAFPN neck (reference: core/yolox/models/yolo_pafpn.py:11-113; built at core/exp.py:372 as
``YOLOPAFPN(0.33, in_features, in_channels=[128, 256, 512], act="silu")``)."""
import torch
import torch.nn as nn

from .network_blocks import BaseConv, CSPLayer


class YOLOPAFPN(nn.Module):
    def __init__(self, depth=1.0, in_features=("dark3", "dark4", "dark5"), in_channels=(256, 512, 1024),
                 depthwise=False, act="silu"):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise convolutions are outside the hot path (SURVEY.md section 8)")
        self.in_features = in_features
        self.in_channels = list(in_channels)
        c3, c4, c5 = (int(c) for c in in_channels)
        n = round(3 * depth)
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.lateral_conv0 = BaseConv(c5, c4, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c4, c4, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c4, c3, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c3, c3, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = BaseConv(c3, c3, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c3, c4, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = BaseConv(c4, c4, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c4, c5, n, False, depthwise=depthwise, act=act)

    def forward(self, out_features):
        x2, x1, x0 = out_features  # dark3, dark4, dark5
        fpn_out0 = self.lateral_conv0(x0)
        f_out0 = self.C3_p4(torch.cat([self.upsample(fpn_out0), x1], 1))
        fpn_out1 = self.reduce_conv1(f_out0)
        pan_out2 = self.C3_p3(torch.cat([self.upsample(fpn_out1), x2], 1))
        pan_out1 = self.C3_n3(torch.cat([self.bu_conv2(pan_out2), fpn_out1], 1))
        pan_out0 = self.C3_n4(torch.cat([self.bu_conv1(pan_out1), fpn_out0], 1))
        return [pan_out2, pan_out1, pan_out0]
