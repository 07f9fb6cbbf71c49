"""Autograd wrapper of the training-mode BaseConv kernels (csrc/train_ops.hip): Conv2d(bias=False) +
BatchNorm2d(batch statistics) + SiLU as ONE ``torch.autograd.Function`` on channels_last tensors
(reference: core/yolox/models/network_blocks.py:33-65; the train step of core/exp.py:283-315).

``base_conv_train(x, conv, bn)`` is what ``BaseConv.forward`` calls in training mode on a ROCm tensor
(FRLW_NATIVE_TRAIN=0 switches back to torch autograd / MIOpen for A/B timing).  The function saves only the
convolution input and output; ``u = gamma * zhat + beta`` is recomputed in the backward.  Running statistics
are updated like ``nn.BatchNorm2d`` (momentum, unbiased variance, num_batches_tracked).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from .. import _lib

_SCRATCH = {}
_WCACHE = {}  # id(weight parameter) -> (weakref, operand cache): forward + data-gradient layouts, rebuilt every forward


def _weight_cache(lib, weight, Cin, Cout, k):
    import weakref
    key = id(weight)
    hit = _WCACHE.get(key)
    n = lib.frlw_baseconv_weight_cache_floats(Cin, Cout, k)
    if hit is None or hit[0]() is not weight or hit[1].numel() < n or hit[1].device != weight.device:
        buf = torch.empty(int(n), dtype=torch.float32, device=weight.device)
        _WCACHE[key] = (weakref.ref(weight, lambda _r, key=key: _WCACHE.pop(key, None)), buf)
        return buf
    return hit[1]


def _scratch(dev, key, numel, dtype):
    k = (dev.index, key, dtype)
    t = _SCRATCH.get(k)
    if t is None or t.numel() < numel:
        t = torch.empty(max(int(numel), 1), dtype=dtype, device=dev)
        _SCRATCH[k] = t
    return t


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _nhwc(x):
    """(B, C, H, W) logical tensor with NHWC storage (no copy when it already is channels_last)."""
    return x.contiguous(memory_format=torch.channels_last)


def pad32(n):
    return (n + 31) // 32 * 32


def native_enabled():
    return os.environ.get("FRLW_NATIVE_TRAIN", "1") != "0"


class _BaseConvTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, stride, eps, run_mean, run_var, momentum):
        ctx.set_materialize_grads(False)
        lib = _lib.load()
        dev = x.device
        x = _nhwc(x.float())
        B, Cin, H, W = x.shape
        Cout, _, k, _ = weight.shape
        pad = (k - 1) // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        w = weight.detach().float().contiguous()
        g = gamma.detach().float().contiguous()
        b = beta.detach().float().contiguous()
        z = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        y = torch.empty_like(z)
        stats = torch.empty((3, Cout), dtype=torch.float32, device=dev)  # mean, biased variance, invstd
        sc = _scratch(dev, "block", lib.frlw_baseconv_train_scratch_bytes(B, H, W, Cin, Cout, k, stride), torch.uint8)
        wc = _weight_cache(lib, weight, Cin, Cout, k)  # both GEMM operands of this weight, laid out once per step
        _lib.check(lib.frlw_baseconv_train_fwd(x.data_ptr(), w.data_ptr(), g.data_ptr(), b.data_ptr(), C.c_float(eps), B, H, W,
                                               Cin, Cout, k, stride, z.data_ptr(), y.data_ptr(), stats[0].data_ptr(),
                                               stats[1].data_ptr(), stats[2].data_ptr(),
                                               run_mean.data_ptr() if run_mean is not None else None,
                                               run_var.data_ptr() if run_var is not None else None, C.c_float(momentum),
                                               wc.data_ptr(), sc.data_ptr(), sc.numel(), _stream(dev)), "baseconv_train_fwd")
        ctx.wcache = wc
        ctx.wversion = weight._version
        ctx.weight_ref = weight
        ctx.save_for_backward(x, z, w, g, b, stats)
        ctx.geom = (B, Cin, H, W, Cout, k, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, z, w, g, b, stats = ctx.saved_tensors
        B, Cin, H, W, Cout, k, stride = ctx.geom
        dev = dy.device
        dy = _nhwc(dy.float())
        dz = torch.empty_like(z)
        dx = (torch.empty((B, Cin, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
              if ctx.needs_input_grad[0] else None)
        dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=dev)
        dgb = torch.empty((2, Cout), dtype=torch.float32, device=dev)
        sc = _scratch(dev, "block", lib.frlw_baseconv_train_scratch_bytes(B, H, W, Cin, Cout, k, stride), torch.uint8)
        # the operand cache belongs to the forward of THIS graph only while the weight (and the cache) are untouched
        # since: a second forward of the same layer before this backward would have overwritten it with the same
        # weights' layout (fine), an in-place weight update in between would not (then lay out again)
        fresh = ctx.weight_ref._version == ctx.wversion and _WCACHE.get(id(ctx.weight_ref), (None, None))[1] is ctx.wcache
        _lib.check(lib.frlw_baseconv_train_bwd(dy.data_ptr(), x.data_ptr(), z.data_ptr(), w.data_ptr(), g.data_ptr(),
                                               b.data_ptr(), stats[0].data_ptr(), stats[2].data_ptr(), B, H, W, Cin, Cout, k,
                                               stride, dz.data_ptr(), dx.data_ptr() if dx is not None else None,
                                               dw.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                                               ctx.wcache.data_ptr() if fresh else None, sc.data_ptr(),
                                               sc.numel(), _stream(dev)), "baseconv_train_bwd")
        return dx, dw, dgb[0], dgb[1], None, None, None, None, None


def base_conv_train(x, conv, bn):
    """silu(bn(conv(x))) with batch statistics; the running statistics are updated in the same launch sequence like
    nn.BatchNorm2d.forward does (momentum, unbiased variance, num_batches_tracked)."""
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = 0.0
    if track:
        bn.num_batches_tracked += 1
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    return _BaseConvTrain.apply(x, conv.weight, bn.weight, bn.bias, conv.stride[0], bn.eps,
                                bn.running_mean if track else None, bn.running_var if track else None, float(momentum))


def eligible(x, conv, bn, act):
    return (native_enabled() and x.is_cuda and isinstance(act, torch.nn.SiLU) and conv.bias is None
            and conv.groups == 1 and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.kernel_size[0] % 2 == 1
            and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
            and conv.padding[0] == (conv.kernel_size[0] - 1) // 2 and conv.dilation[0] == 1)
