"""Building blocks of the YOLOX detector (reference: core/yolox/models/network_blocks.py).

Same attribute names as the reference so checkpoints interchange: ``BaseConv.{conv,bn,act}``,
``Bottleneck.{conv1,conv2}``, ``CSPLayer.{conv1,conv2,conv3,m}``, ``SPPBottleneck.{conv1,m,conv2}``,
``Focus.conv``.
"""
import torch
import torch.nn as nn


class SiLU(nn.Module):
    @staticmethod
    def forward(x):
        return x * torch.sigmoid(x)


def get_activation(name="silu", inplace=True):
    # network_blocks.py:19-30
    if name == "silu":
        return nn.SiLU(inplace=inplace)
    if name == "relu":
        return nn.ReLU(inplace=inplace)
    if name == "lrelu":
        return nn.LeakyReLU(0.1, inplace=inplace)
    if name == "gelu":
        return nn.GELU()
    raise AttributeError(f"Unsupported act type: {name}")


class BaseConv(nn.Module):
    """Conv2d(k, s, pad=(k-1)//2, bias=False) -> BatchNorm2d -> activation (network_blocks.py:33-65)."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride,
                              padding=(ksize - 1) // 2, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = get_activation(act, inplace=True)

    def forward(self, x, into=None):
        if self.training and x.is_cuda:
            # train step on the GPU: forward, data / weight gradients and BatchNorm + SiLU run in the gfx950 kernels of
            # csrc/train_ops.hip (FRLW_NATIVE_TRAIN=0: torch autograd / MIOpen, for A/B timing)
            from . import train_ops
            if train_ops.eligible(x, self.conv, self.bn, self.act):
                return train_ops.base_conv_train(x, self.conv, self.bn, into=into)
        if into is not None:  # (callers ask train_ops.csp_join_eligible first)
            raise RuntimeError("BaseConv: a destination slice needs the native training path")
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    """1x1 -> 3x3, residual iff shortcut and cin == cout (network_blocks.py:89-111)."""

    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise convolutions are outside the hot path (SURVEY.md section 8)")
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def forward(self, x, into=None):
        if self.use_add and self.training and x.is_cuda:
            from . import train_ops
            if train_ops.bottleneck_eligible(x, self.conv1, self.conv2):  # shortcut and its gradient inside the blocks' launches
                return train_ops.bottleneck_train(x, self.conv1, self.conv2, into=into)
        if into is not None and self.use_add:
            raise RuntimeError("Bottleneck: a destination slice needs the native training path")
        y = self.conv2(self.conv1(x), into=into) if into is not None else self.conv2(self.conv1(x))
        return y + x if self.use_add else y


class SPPBottleneck(nn.Module):
    """1x1, max-pool 5/9/13 (stride 1, same pad), concat(4), 1x1 (network_blocks.py:131-153)."""

    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), activation="silu"):
        super().__init__()
        hidden = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=activation)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.conv2 = BaseConv(hidden * (len(kernel_sizes) + 1), out_channels, 1, stride=1, act=activation)

    def forward(self, x):
        x = self.conv1(x)
        if self.training and x.is_cuda:
            from . import train_ops
            if train_ops.spp_pools_eligible(x, self.m):  # pools + concat + their gradient as one native op each way
                return self.conv2(train_ops.spp_pools(x))
            x = x.contiguous()  # ATen's channels_last max-pool backward is 0.5 ms per pool (measured); NCHW is not
        return self.conv2(torch.cat([x] + [m(x) for m in self.m], dim=1))


class CSPLayer(nn.Module):
    """Two 1x1 branches, n bottlenecks (expansion 1.0) on the first, concat, 1x1 (network_blocks.py:156-194)."""

    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])

    def forward(self, x):
        if self.training and x.is_cuda:
            from . import train_ops
            if train_ops.csp_join_eligible(x, self):
                # conv1 | conv2 as one autograd node (dx of the two branches summed in an epilogue); conv2 and the last Bottleneck
                # write their activations straight into the two halves of the buffer conv3 reads: no concatenation launch
                hidden = self.conv1.conv.out_channels
                buf = torch.empty((x.shape[0], 2 * hidden, x.shape[2], x.shape[3]), dtype=torch.float32, device=x.device,
                                  memory_format=torch.channels_last)
                x_1, x_2 = train_ops.pair_train(x, self.conv1, self.conv2, into_b=buf[:, hidden:])
                for i, blk in enumerate(self.m):
                    x_1 = blk(x_1, into=buf[:, :hidden] if i == len(self.m) - 1 else None)
                return self.conv3(train_ops.join_slices(buf, x_1, x_2))
            if train_ops.pair_eligible(x, self.conv1, self.conv2):  # one autograd node: dx of the two branches summed in an epilogue
                x_1, x_2 = train_ops.pair_train(x, self.conv1, self.conv2)
                return self.conv3(torch.cat((self.m(x_1), x_2), dim=1))
        return self.conv3(torch.cat((self.m(self.conv1(x)), self.conv2(x)), dim=1))


class Focus(nn.Module):
    """Space-to-depth in the order TL, BL, TR, BR, then BaseConv(4C -> out) (network_blocks.py:196-221).

    The input carries one trailing singleton dimension that is dropped here (``x[..., 0]``, :221).
    """

    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act="silu"):
        super().__init__()
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, act=act)

    @staticmethod
    def space_to_depth(x):
        return torch.cat((x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]), dim=1)

    def forward(self, x):
        x = x[..., 0]
        if self.training and x.is_cuda and not x.requires_grad and x.dtype == torch.float32 and x.shape[-1] % 2 == 0 \
                and x.shape[-2] % 2 == 0:
            from . import train_ops
            if train_ops.native_enabled():  # one kernel, channels_last out: what the native BaseConv reads without a copy
                return self.conv(train_ops.focus_nhwc(x))
        return self.conv(self.space_to_depth(x))
