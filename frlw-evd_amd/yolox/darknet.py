"""CSPDarknet backbone (reference: core/yolox/models/darknet.py:270-354; built at core/exp.py:582 as
``CSPDarknet(C, 0.33, 0.5, stem=Focus)``)."""
import torch.nn as nn

from .network_blocks import BaseConv, CSPLayer, Focus, SPPBottleneck


class CSPDarknet(nn.Module):
    def __init__(self, in_channel, dep_mul, wid_mul, out_features=("dark3", "dark4", "dark5"), depthwise=False,
                 act="silu", stem=Focus):
        super().__init__()
        assert out_features, "please provide output features of Darknet"
        if depthwise:
            raise NotImplementedError("depthwise convolutions are outside the hot path (SURVEY.md section 8)")
        self.out_features = out_features
        c = int(wid_mul * 64)
        d = max(round(dep_mul * 3), 1)
        self.stem = stem(in_channel, c, ksize=3, act=act)
        self.dark2 = nn.Sequential(BaseConv(c, c * 2, 3, 2, act=act),
                                   CSPLayer(c * 2, c * 2, n=d, depthwise=depthwise, act=act))
        self.dark3 = nn.Sequential(BaseConv(c * 2, c * 4, 3, 2, act=act),
                                   CSPLayer(c * 4, c * 4, n=d * 3, depthwise=depthwise, act=act))
        self.dark4 = nn.Sequential(BaseConv(c * 4, c * 8, 3, 2, act=act),
                                   CSPLayer(c * 8, c * 8, n=d * 3, depthwise=depthwise, act=act))
        self.dark5 = nn.Sequential(BaseConv(c * 8, c * 16, 3, 2, act=act),
                                   SPPBottleneck(c * 16, c * 16, activation=act),
                                   CSPLayer(c * 16, c * 16, n=d, shortcut=False, depthwise=depthwise, act=act))

    def forward(self, x):
        outputs = {}
        x = self.stem(x)
        outputs["stem"] = x
        for name in ("dark2", "dark3", "dark4", "dark5"):
            x = getattr(self, name)(x)
            outputs[name] = x
        return [outputs[k] for k in self.out_features]
