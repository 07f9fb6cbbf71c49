"""BFM stem of the ``yolox_taf_bfm`` recipes: ``Temporal_Active_Focus_connect``
(reference: core/Others/Temporal_Active_Focus.py:62-127; selected at core/exp.py:591, built by
``CSPDarknet(C, 0.33, 0.5, stem=Temporal_Active_Focus_connect)`` with ``ksize=3, act="silu"``, core/exp.py:582).

The TAF tensor has ``C = 2 * T`` channels (polarity-minor, ``T`` FIFO slots).  ``log2(T)`` weight-normed grouped
1x1 convolutions + ReLU halve the number of time groups each; the first ``embed_dim = 4`` channels of every
stage are concatenated (``4 * log2(T)`` channels), pass a residual two-layer 1x1 MLP (x4 expansion, SiLU; the
Dropout2d layers are identities in eval mode), then the usual Focus space-to-depth + BaseConv.

Parameter names equal the reference's (``convs.{i}.weight_g / weight_v / bias`` from the old-style
``nn.utils.weight_norm``, ``trans_up``, ``trans_down``, ``conv.conv`` / ``conv.bn``).  On a ROCm device in eval
mode the whole per-pixel part runs as ONE fused kernel of the detector plan (csrc/detector.hip, k_bfm_stem).
"""
import warnings
from math import log2

import torch
import torch.nn as nn

from .network_blocks import Focus, get_activation


class Temporal_Active_Focus_connect(Focus):
    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act="silu"):
        time_channels = int(in_channels / 2)
        embed_dim = 4
        reduce_times = int(log2(time_channels))
        super().__init__(embed_dim * reduce_times, out_channels, ksize, stride, act)
        self.embed_dim = embed_dim
        self.convs = nn.ModuleList()
        self.relu = nn.ReLU()
        for i in range(reduce_times):
            input_dim = 2 if i == 0 else embed_dim
            with warnings.catch_warnings():  # the old-style weight_norm keeps the reference's weight_g / weight_v names
                warnings.simplefilter("ignore", FutureWarning)
                self.convs.append(nn.utils.weight_norm(
                    nn.Conv2d(int(input_dim * time_channels), int(embed_dim * time_channels / 2), 1,
                              groups=int(time_channels / 2))))
            time_channels = time_channels / 2
        self.trans_up = nn.Conv2d(embed_dim * reduce_times, embed_dim * reduce_times * 4, 1)
        self.act = get_activation(act)
        self.drop = nn.Dropout2d(0.1)
        self.trans_down = nn.Conv2d(embed_dim * reduce_times * 4, embed_dim * reduce_times, 1)
        for c in self.convs:  # Temporal_Active_Focus.py:86-93
            c.weight_v.data.normal_(0, 0.01)

    def mix(self, x):
        """The per-pixel part: (B, C, H, W) -> (B, 4 * log2(T), H, W)."""
        xout = []
        for conv in self.convs:
            x = self.relu(conv(x))
            xout.append(x[:, :self.embed_dim])
        x = torch.cat(xout, dim=1)
        y = self.drop(self.act(self.trans_up(x)))
        y = self.drop(self.trans_down(y))
        return x + y

    def forward(self, x):
        return self.conv(self.space_to_depth(self.mix(x[..., 0])))
