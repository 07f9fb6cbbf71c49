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

from .. import _lib, _pins

_SCRATCH = {}
_WCACHE = {}  # id(weight parameter) -> (weakref, operand cache): forward + data-gradient layouts, rebuilt every forward
_WCACHE_GEOM = {}  # id(weight parameter) -> what the data-gradient layout in the cache was laid out for: (version, parity)


PRECISIONS = {"f32": 0, "bf16x3": 1}


def conv_precision():
    """Arithmetic of the native contractions of the train step: ``FRLW_CONV_PRECISION`` = ``f32`` (the default: float32 MFMA,
    exact products) or ``bf16x3`` (opt-in: float32 products from three bf16 MFMAs, <= 2^-16 relative error per product,
    float32 accumulation, 1.45 x the steps per second)."""
    v = os.environ.get("FRLW_CONV_PRECISION", "f32").strip().lower()
    if v not in PRECISIONS:
        raise ValueError(f"FRLW_CONV_PRECISION={v!r}: expected one of {sorted(PRECISIONS)}")
    return PRECISIONS[v]


def layer_precision(Cout, k, stride):
    """Per layer: the split data-gradient operand of a stride-2 3x3 layer is grouped by parity class in blocks of Cout rows,
    which must be whole k-tiles of 16 (every width of the shipped networks is); other widths keep the float32 MFMA."""
    p = conv_precision()
    return p if (p == 0 or not (k == 3 and stride == 2) or Cout % 16 == 0) else 0


def _weight_cache(lib, weight, Cin, Cout, k):
    import weakref
    key = id(weight)
    hit = _WCACHE.get(key)
    n = lib.frlw_baseconv_weight_cache_floats(Cin, Cout, k, 1)  # room for either precision
    if hit is None or hit[0]() is not weight or hit[1].numel() < n or hit[1].device != weight.device:
        buf = torch.empty(int(n), dtype=torch.float32, device=weight.device)
        if hit is not None:
            _pins.retire(hit[1])  # a live HIP graph may still launch kernels on the old cache
        _WCACHE[key] = (weakref.ref(weight, lambda _r, key=key: _WCACHE.pop(key, None)), buf)
        return buf
    return hit[1]


_PAIRS = {}  # id(first weight of a stacked pair) -> (weakref of it, weakref of the second weight); set by the pair's forward


def _pair_second(w):
    """The second weight of the stacked pair whose first weight is ``w`` (None: ``w`` is an ordinary weight)."""
    refs = _PAIRS.get(id(w))
    if refs is None:
        return None
    w2 = refs[1]()
    if refs[0]() is not w or w2 is None:  # (an id reused by another tensor, or a pair whose second weight is gone)
        _PAIRS.pop(id(w), None)
        return None
    return w2


def _ver(w, w2=None):
    return w._version if w2 is None else (w._version, w2._version)


_WREADY = {}  # id(weight parameter) -> (version, parity, cache data_ptr) the batched layout below has laid the cache out for
_PLANS = {}  # id(model) -> (weakref, item table on the device, total elements, [(weight, parity, cache, weight pointer)])


def _layout_plan(model):
    """One table entry per natively trained BaseConv weight of `model` whose operand cache exists (= that has run one
    forward): frlw_weight_layout_item_t {w, w_fwd, w_dgrad, Cout, Cin, k, parity, precision, reserved, first, w2, split, reserved2};
    the first weight of a stacked pair stands for both (Cout = the pair's channels, w2 / split = the second weight)."""
    import struct
    import weakref
    lib = _lib.load()
    rows, blob, first = [], b"", 0
    for mod in model.modules():
        conv, bn = getattr(mod, "conv", None), getattr(mod, "bn", None)
        if not isinstance(conv, torch.nn.Conv2d) or not isinstance(bn, torch.nn.BatchNorm2d):
            continue
        w = conv.weight
        hit, geom = _WCACHE.get(id(w)), _WCACHE_GEOM.get(id(w))
        if hit is None or geom is None or hit[0]() is not w or not w.is_cuda or not w.is_contiguous() or w.dtype != torch.float32:
            continue
        Cout, Cin, k, _ = w.shape
        w2 = _pair_second(w)
        split = 0
        if w2 is not None:
            if not (w2.is_cuda and w2.is_contiguous() and w2.dtype == torch.float32 and isinstance(geom[0], tuple)):
                continue
            split, Cout = Cout, Cout + w2.shape[0]
        elif isinstance(geom[0], tuple):
            continue  # the cache was last laid out for a pair that is gone
        cache = hit[1]
        prec = geom[2]
        n_f, n_d = lib.frlw_conv_operand_floats(k * k * Cin, Cout, prec), lib.frlw_conv_operand_floats(k * k * Cout, Cin, prec)
        if cache.numel() < n_f + n_d:
            continue
        blob += struct.pack("<QQQiiiiiiqQii", w.data_ptr(), cache.data_ptr(), cache.data_ptr() + 4 * n_f, Cout, Cin, k, geom[1], prec, 0, first,
                            w2.data_ptr() if w2 is not None else 0, split, 0)
        first += n_f + n_d
        rows.append((w, (geom[1], prec), cache, w.data_ptr(), w2, w2.data_ptr() if w2 is not None else 0))
    if not rows:
        return None
    import numpy as np
    table = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(rows[0][0].device)
    return (weakref.ref(model), table, first, rows)


def layout_all_weights(model):
    """Lay out the GEMM operands of EVERY BaseConv weight of `model` in one launch (call it once per step, before the
    forward: the per-layer forwards then find their cache ready and skip their own layout kernel -- 74 launches of ~5 us).
    Does nothing until the layers have run once (their caches are created by the first forward), or off the GPU."""
    if not native_enabled():
        return False
    plan = _PLANS.get(id(model))
    stale = plan is None or plan[0]() is not model
    if not stale:
        for w, parity, cache, ptr, w2, ptr2 in plan[3]:  # parity = (parity class, precision) the entry was written for
            hit, geom = _WCACHE.get(id(w)), _WCACHE_GEOM.get(id(w))
            if (hit is None or hit[1] is not cache or geom is None or geom[1:] != parity or w.data_ptr() != ptr
                    or _pair_second(w) is not w2 or (w2 is not None and w2.data_ptr() != ptr2) or isinstance(geom[0], tuple) != (w2 is not None)):
                stale = True
                break
    if stale:
        if plan is not None:
            _pins.retire(plan[1])  # the item table a captured layout launch reads
        plan = _layout_plan(model)
        if plan is None:
            _PLANS.pop(id(model), None)
            return False
        _PLANS[id(model)] = plan
    _, table, total, rows = plan
    lib = _lib.load()
    _lib.check(lib.frlw_conv_weight_layouts_batch(table.data_ptr(), len(rows), total, _stream(table.device)), "weight_layouts_batch")
    for w, parity, cache, _ptr, w2, _ptr2 in rows:
        _WREADY[id(w)] = (_ver(w, w2), parity, cache.data_ptr())
        _WCACHE_GEOM[id(w)] = (_ver(w, w2), *parity)
    return True


def _bump_versions(*tensors):
    """Advance the in-place version counters of buffers a kernel has written through raw pointers (no launch)."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


def _scratch(dev, key, numel, dtype):
    k = (dev.index, key, dtype)
    t = _SCRATCH.get(k)
    if t is None or t.numel() < numel:
        _pins.retire(t)
        t = torch.empty(max(int(numel), 1), dtype=dtype, device=dev)
        _SCRATCH[k] = t
    return t


def _sk_counters(dev):
    """The arrival counters of the in-kernel split-K reduction (frlw_evd.h: frlw_baseconv_train_fwd): 1024 ints per device,
    zero when created, kept at zero by the kernels themselves; one buffer for all layers (launches of a stream are ordered)."""
    k = (dev.index, "splitk_counters", torch.int32)
    t = _SCRATCH.get(k)
    if t is None:
        t = torch.zeros(1024, dtype=torch.int32, device=dev)
        _SCRATCH[k] = t
    return t


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _nhwc(x):
    """(B, C, H, W) logical tensor with NHWC storage (no copy when it already is channels_last)."""
    return x.contiguous(memory_format=torch.channels_last)


def _rows_in_place(t):
    """(tensor, floats between two pixel rows) of a (B, C, H, W) gradient for the kernels that read (M, C) rows: a channel
    slice of a wider channels_last tensor -- what the backward of ``torch.cat`` hands to each of its inputs -- is read where it
    lies (row stride = the wide tensor's channels) instead of being copied dense first; anything else becomes dense NHWC."""
    B, Cc, H, W = t.shape
    st = t.stride()
    cs = st[3]
    if (st[1] == 1 and cs >= Cc and cs % 4 == 0 and st[2] == W * cs and (B == 1 or st[0] == H * W * cs)
            and t.data_ptr() % 16 == 0 and cs != Cc):
        return t, cs
    t = _nhwc(t)
    return t, 0


def pad32(n):
    return (n + 31) // 32 * 32


def native_enabled():
    return os.environ.get("FRLW_NATIVE_TRAIN", "1") != "0"


class _Rec:
    """What one BaseConv forward leaves for its backward beside the saved tensors (x, z, w, gamma, beta, stats)."""
    __slots__ = ("geom", "wcache", "wparity", "wversion", "weight_ref", "weight2_ref", "split")


def _slice_ok(out, shape):
    return out.shape == shape and out.dtype == torch.float32 and out.stride()[1] == 1 and out.data_ptr() % 16 == 0


def _fwd_one(x, weight, gamma, beta, stride, eps, run_mean, run_var, momentum, tracked, residual=None, out=None, pair=None):
    """One frlw_baseconv_train_fwd call: (y, tensors to save, _Rec).  ``x``: float32 NHWC storage.  ``residual``: added to y in
    the pass that writes it; ``out``: a (B, Cout, Ho, Wo) channel slice of a wider channels_last tensor y is written into.
    ``pair`` = (weight2, gamma2, beta2, (eps2, run_mean2, run_var2, momentum2, tracked2), out2): a SECOND BaseConv reading the same
    x, stacked along the output channels (frlw_baseconv_fuse_t::split) -- returns ((y, y2), tensors to save, _Rec)."""
    lib = _lib.load()
    dev = x.device
    B, Cin, H, W = x.shape
    C1, _, k, _ = weight.shape
    Cout = C1 + (pair[0].shape[0] if pair is not None else 0)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    w = weight.detach().float().contiguous()
    g = gamma.detach().float().contiguous()
    b = beta.detach().float().contiguous()
    z = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    fuse = None
    extra = ()
    y_shape = (B, C1, Ho, Wo)
    if residual is not None or out is not None:
        res, res_rs = (None, 0) if residual is None else _rows_in_place(residual)
        y_rs = 0
        if out is not None:
            assert _slice_ok(out, y_shape)
            y_rs = out.stride()[3]
        fuse = _lib.FrlwBaseconvFuse(residual=None if res is None else res.data_ptr(), residual_row_stride=res_rs, y_row_stride=y_rs)
    y = out if out is not None else torch.empty(y_shape, dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    if pair is not None:
        weight2, gamma2, beta2, cfg2, out2 = pair
        w2, g2, b2 = weight2.detach().float().contiguous(), gamma2.detach().float().contiguous(), beta2.detach().float().contiguous()
        y2_shape = (B, Cout - C1, Ho, Wo)
        assert out2 is None or _slice_ok(out2, y2_shape)
        y2 = out2 if out2 is not None else torch.empty(y2_shape, dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        fuse = fuse if fuse is not None else _lib.FrlwBaseconvFuse()
        fuse.split, fuse.w2, fuse.gamma2, fuse.beta2 = C1, w2.data_ptr(), g2.data_ptr(), b2.data_ptr()
        fuse.running_mean2 = cfg2[1].data_ptr() if cfg2[1] is not None else None
        fuse.running_var2 = cfg2[2].data_ptr() if cfg2[2] is not None else None
        fuse.num_batches_tracked2 = cfg2[4].data_ptr() if cfg2[4] is not None else None
        fuse.y2, fuse.y2_row_stride = y2.data_ptr(), (y2.stride()[3] if out2 is not None else 0)
        assert cfg2[0] == eps and cfg2[3] == momentum and (cfg2[1] is None) == (run_mean is None)
        extra = (w2, g2, b2)
        import weakref
        _PAIRS[id(weight)] = (weakref.ref(weight), weakref.ref(weight2))
    stats = torch.empty((3, Cout), dtype=torch.float32, device=dev)  # mean, biased variance, invstd
    sc = _scratch(dev, "block", lib.frlw_baseconv_train_scratch_bytes(B, H, W, Cin, Cout, k, stride), torch.uint8)
    wc = _weight_cache(lib, weight, Cin, Cout, k)  # both GEMM operands of this weight, laid out once per step
    parity = int(lib.frlw_conv2d_dgrad_parity(k, stride, H, W))
    prec = layer_precision(Cout, k, stride)
    # laid out already by layout_all_weights() for exactly this weight version, parity class, precision and buffer?
    weight2 = pair[0] if pair is not None else None
    if pair is None:
        _PAIRS.pop(id(weight), None)  # this weight runs on its own (again): its cache is laid out for itself
    ver = _ver(weight, weight2)
    ready = (_WREADY.get(id(weight)) == (ver, (parity, prec), wc.data_ptr()) and w.data_ptr() == weight.data_ptr()
             and (weight2 is None or extra[0].data_ptr() == weight2.data_ptr()))
    _lib.check(lib.frlw_baseconv_train_fwd(x.data_ptr(), None if ready else w.data_ptr(), g.data_ptr(), b.data_ptr(), C.c_float(eps), B, H, W,
                                           Cin, Cout, k, stride, z.data_ptr(), y.data_ptr(), stats[0].data_ptr(),
                                           stats[1].data_ptr(), stats[2].data_ptr(),
                                           run_mean.data_ptr() if run_mean is not None else None,
                                           run_var.data_ptr() if run_var is not None else None, C.c_float(momentum),
                                           tracked.data_ptr() if tracked is not None else None,
                                           wc.data_ptr(), sc.data_ptr(), sc.numel(), _sk_counters(dev).data_ptr(),
                                           C.byref(fuse) if fuse is not None else None, prec, _stream(dev)), "baseconv_train_fwd")
    rec = _Rec()
    rec.wcache = wc
    # the data-gradient half of the cache depends on the parity class of (k, stride, H, W): a second forward of the same
    # layer on an input of another parity (shared layer, multi-scale graph) re-lays it -- remember what THIS forward wrote
    rec.wparity = (parity, prec)
    _WCACHE_GEOM[id(weight)] = (ver, parity, prec)
    rec.wversion = ver
    rec.weight_ref = weight
    rec.weight2_ref = weight2
    rec.split = C1 if pair is not None else 0
    rec.geom = (B, Cin, H, W, Cout, k, stride)
    return (y if pair is None else (y, y2)), (x, z, w, g, b, stats) + extra, rec


def _bwd_one(rec, saved, dy, need_dx, dx_add=None, dy2=None):
    """One frlw_baseconv_train_bwd call: (dx or None, dw, dgamma, dbeta).  ``dx_add``: the gradient another consumer of the same
    input has produced already, added to dx in the data gradient's epilogue (stride-1 layers).  A stacked pair (rec.split): ``dy`` /
    ``dy2`` are the two blocks' upstream gradients, dw / dgamma / dbeta come back stacked (the second block's rows from rec.split)."""
    lib = _lib.load()
    x, z, w, g, b, stats = saved[:6]
    B, Cin, H, W, Cout, k, stride = rec.geom
    dev = dy.device
    dy, dy_rs = _rows_in_place(dy.float())
    dz = torch.empty_like(z)
    dx = (torch.empty((B, Cin, H, W), dtype=torch.float32, device=dev, memory_format=torch.channels_last) if need_dx else None)
    dw = torch.empty((Cout, Cin, k, k), dtype=torch.float32, device=dev)
    dgb = torch.empty((2, Cout), dtype=torch.float32, device=dev)
    fuse = None
    if dx_add is not None and dx is not None:
        add, add_rs = _rows_in_place(dx_add.float())
        fuse = _lib.FrlwBaseconvFuse(dx_add=add.data_ptr(), dx_add_row_stride=add_rs)
    if rec.split:
        w2, g2, b2 = saved[6:9]
        dy2, dy2_rs = _rows_in_place(dy2.float())
        fuse = fuse if fuse is not None else _lib.FrlwBaseconvFuse()
        fuse.split, fuse.w2, fuse.gamma2, fuse.beta2 = rec.split, w2.data_ptr(), g2.data_ptr(), b2.data_ptr()
        fuse.dy2, fuse.dy2_row_stride = dy2.data_ptr(), dy2_rs
    sc = _scratch(dev, "block", lib.frlw_baseconv_train_scratch_bytes(B, H, W, Cin, Cout, k, stride), torch.uint8)
    # the operand cache belongs to the forward of THIS graph only while the weight (and the cache) are untouched
    # since: a second forward of the same layer before this backward would have overwritten it with the same
    # weights' layout (fine), an in-place weight update in between would not (then lay out again)
    fresh = (_ver(rec.weight_ref, rec.weight2_ref) == rec.wversion and _WCACHE.get(id(rec.weight_ref), (None, None))[1] is rec.wcache
             and _WCACHE_GEOM.get(id(rec.weight_ref)) == (rec.wversion, *rec.wparity))
    _lib.check(lib.frlw_baseconv_train_bwd(dy.data_ptr(), dy_rs, x.data_ptr(), z.data_ptr(), w.data_ptr(), g.data_ptr(),
                                           b.data_ptr(), stats[0].data_ptr(), stats[2].data_ptr(), B, H, W, Cin, Cout, k,
                                           stride, dz.data_ptr(), dx.data_ptr() if dx is not None else None,
                                           dw.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr(),
                                           rec.wcache.data_ptr() if fresh else None, sc.data_ptr(),
                                           sc.numel(), _sk_counters(dev).data_ptr(), C.byref(fuse) if fuse is not None else None,
                                           rec.wparity[1], _stream(dev)), "baseconv_train_bwd")
    return dx, dw, dgb[0], dgb[1]


class _Into:
    """A destination slice handed to an autograd Function without autograd seeing it as an input: the (B, C, H, W) channel slice
    of a wider channels_last buffer a block writes its activation into (the concatenation buffer of a CSPLayer)."""
    __slots__ = ("t",)

    def __init__(self, t):
        # an ALIAS of the slice (same storage, offset, shape, strides), not a view: the tensor a Function returns must not be an
        # autograd view of a buffer that other Functions write other slices of (autograd's view + in-place logic would take over)
        self.t = torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())


class _BaseConvTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, stride, eps, run_mean, run_var, momentum, tracked=None, into=None):
        ctx.set_materialize_grads(False)
        y, saved, ctx.rec = _fwd_one(_nhwc(x.float()), weight, gamma, beta, stride, eps, run_mean, run_var, momentum, tracked,
                                     out=into.t if into is not None else None)
        ctx.save_for_backward(*saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, dw, dg, db = _bwd_one(ctx.rec, ctx.saved_tensors, dy, ctx.needs_input_grad[0])
        return dx, dw, dg, db, None, None, None, None, None, None, None


class _JoinSlices(torch.autograd.Function):
    """The concatenation that never ran: ``parts`` are channel slices of ``whole`` (in order, together all of it), written there by
    the blocks that produced them; forward hands out the buffer, backward hands every producer its slice of the gradient (read
    in place by the blocks' backward kernels, like the slices torch.cat's backward makes)."""

    @staticmethod
    def forward(ctx, whole, *parts):
        w = whole.t
        lo = 0
        for p in parts:
            assert p.data_ptr() == w.data_ptr() + 4 * lo and p.stride() == w.stride() and p.shape[0] == w.shape[0] and p.shape[2:] == w.shape[2:]
            lo += p.shape[1]
        assert lo == w.shape[1]
        ctx.widths = [p.shape[1] for p in parts]
        return w

    @staticmethod
    def backward(ctx, g):
        out, lo = [], 0
        for wd in ctx.widths:
            out.append(g[:, lo:lo + wd] if g is not None else None)
            lo += wd
        return (None, *out)


def join_slices(whole, *parts):
    return _JoinSlices.apply(_Into(whole), *parts)


class _BottleneckTrain(torch.autograd.Function):
    """conv2(conv1(x)) + x of a Bottleneck with shortcut (network_blocks.py:89-111) as one autograd node: the shortcut is added
    by the pass that writes conv2's activation, and its gradient by the epilogue of conv1's data gradient -- the two
    elementwise launches autograd would make (forward add, gradient accumulation) are gone; same additions, same bits."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, cfg1, cfg2, into=None):
        ctx.set_materialize_grads(False)
        x = _nhwc(x.float())
        h, s1, ctx.rec1 = _fwd_one(x, w1, g1, b1, 1, *cfg1)
        y, s2, ctx.rec2 = _fwd_one(h, w2, g2, b2, 1, *cfg2, residual=x, out=into.t if into is not None else None)
        ctx.save_for_backward(*s1, *s2)
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * 10
        t = ctx.saved_tensors
        dh, dw2, dg2, db2 = _bwd_one(ctx.rec2, t[6:], dy, True)
        dx, dw1, dg1, db1 = _bwd_one(ctx.rec1, t[:6], dh, True, dx_add=dy)
        return dx, dw1, dg1, db1, dw2, dg2, db2, None, None, None


class _PairTrain(torch.autograd.Function):
    """Two BaseConvs reading the same x (conv1 | conv2 of a CSPLayer, network_blocks.py:191-193; the first cls / reg convolution
    of a head level, yolo_head.py:160-164) as one autograd node: the second data gradient adds the first in its epilogue
    instead of autograd launching the accumulation."""

    @staticmethod
    def forward(ctx, x, wa, ga, ba, wb, gb, bb, stride, cfga, cfgb, into_b=None):
        ctx.set_materialize_grads(False)
        x = _nhwc(x.float())
        ya, sa, ctx.reca = _fwd_one(x, wa, ga, ba, stride, *cfga)
        yb, sb, ctx.recb = _fwd_one(x, wb, gb, bb, stride, *cfgb, out=into_b.t if into_b is not None else None)
        ctx.save_for_backward(*sa, *sb)
        return ya, yb

    @staticmethod
    def backward(ctx, dya, dyb):
        t = ctx.saved_tensors
        need = ctx.needs_input_grad[0]
        dx, ga, gb = None, (None, None, None), (None, None, None)
        if dyb is not None:
            dx, *gb = _bwd_one(ctx.recb, t[6:], dyb, need)
        if dya is not None:
            dx, *ga = _bwd_one(ctx.reca, t[:6], dya, need, dx_add=dx)
        return (dx, *ga, *gb, None, None, None, None)


class _PairStackTrain(torch.autograd.Function):
    """Two stride-1 BaseConvs with the same kernel size reading the same x as ONE stacked block (frlw_baseconv_fuse_t::split): one
    convolution with both blocks' output channels, one statistics / BatchNorm + SiLU pass writing the two activations to their
    own destinations, and in the backward one BatchNorm backward reading the two upstream gradients where they lie, ONE data
    gradient (the sum over both blocks by construction) and ONE weight gradient -- ten launches per pair and step less than
    _PairTrain, and thin layers twice as wide.  Per element the contraction is summed in the same order as in the separate
    blocks unless the stacked layer splits its contraction differently; the batch statistics are float64 sums over other slabs:
    results agree with _PairTrain's to rounding (tests: 1e-5), not bit for bit."""

    @staticmethod
    def forward(ctx, x, wa, ga, ba, wb, gb, bb, stride, cfga, cfgb, into_b=None):
        ctx.set_materialize_grads(False)
        x = _nhwc(x.float())
        (ya, yb), saved, ctx.rec = _fwd_one(x, wa, ga, ba, stride, *cfga,
                                            pair=(wb, gb, bb, cfgb, into_b.t if into_b is not None else None))
        ctx.save_for_backward(*saved)
        return ya, yb

    @staticmethod
    def backward(ctx, dya, dyb):
        if dya is None and dyb is None:
            return (None,) * 11
        t = ctx.saved_tensors
        B, Cin, H, W, Cout, k, stride = ctx.rec.geom
        h = ctx.rec.split
        z = t[1]
        if dya is None:
            dya = torch.zeros((B, h, z.shape[2], z.shape[3]), dtype=torch.float32, device=z.device, memory_format=torch.channels_last)
        if dyb is None:
            dyb = torch.zeros((B, Cout - h, z.shape[2], z.shape[3]), dtype=torch.float32, device=z.device, memory_format=torch.channels_last)
        dx, dw, dg, db = _bwd_one(ctx.rec, t, dya, ctx.needs_input_grad[0], dy2=dyb)
        return dx, dw[:h], dg[:h], db[:h], dw[h:], dg[h:], db[h:], None, None, None, None


def _bn_cfg(bn):
    """(eps, running_mean, running_var, momentum, num_batches_tracked on the device or None) of one forward, like
    nn.BatchNorm2d.forward; a cumulative-average module (momentum None) bumps its counter on the host here."""
    track = bn.track_running_stats and bn.running_mean is not None
    momentum = 0.0
    tracked = None
    if track:
        if bn.momentum is None:  # cumulative average: the factor needs the counter's value on the host (one sync)
            bn.num_batches_tracked += 1
            momentum = 1.0 / float(bn.num_batches_tracked)
        else:  # the counter is bumped on the device by the statistics kernel: no extra launch per layer
            momentum = bn.momentum
            tracked = bn.num_batches_tracked if bn.num_batches_tracked.is_cuda else None
            if tracked is None:
                bn.num_batches_tracked += 1
    return (bn.eps, bn.running_mean if track else None, bn.running_var if track else None, float(momentum), tracked), track


def _bn_done(bn, track):
    if track:
        # the kernels wrote the running statistics through raw pointers: tell autograd's version counters, which is what the
        # eval engine's weight signature (yolox/model.py) watches -- also when only a submodule is in training mode
        _bump_versions(bn.running_mean, bn.running_var)


def base_conv_train(x, conv, bn, into=None):
    """silu(bn(conv(x))) with batch statistics; the running statistics are updated in the same launch sequence like
    nn.BatchNorm2d.forward does (momentum, unbiased variance, num_batches_tracked).  ``into``: the channel slice of a wider
    channels_last buffer the activation is written into (and returned as)."""
    cfg, track = _bn_cfg(bn)
    y = _BaseConvTrain.apply(x, conv.weight, bn.weight, bn.bias, conv.stride[0], *cfg, _Into(into) if into is not None else None)
    _bn_done(bn, track)
    return y


def fuse_enabled():
    """FRLW_TRAIN_FUSE=0: every BaseConv its own autograd node again (A/B timing; the results are the same bits)."""
    return os.environ.get("FRLW_TRAIN_FUSE", "1") != "0"


def bottleneck_train(x, c1, c2, into=None):
    """``c2(c1(x)) + x`` for two BaseConv modules (Bottleneck with shortcut) as one autograd node."""
    cfg1, t1 = _bn_cfg(c1.bn)
    cfg2, t2 = _bn_cfg(c2.bn)
    y = _BottleneckTrain.apply(x, c1.conv.weight, c1.bn.weight, c1.bn.bias, c2.conv.weight, c2.bn.weight, c2.bn.bias, cfg1, cfg2,
                               _Into(into) if into is not None else None)
    _bn_done(c1.bn, t1)
    _bn_done(c2.bn, t2)
    return y


def pair_train(x, ca, cb, into_b=None):
    """``(ca(x), cb(x))`` for two stride-1 BaseConv modules reading the same input, as one autograd node."""
    cfga, ta = _bn_cfg(ca.bn)
    cfgb, tb = _bn_cfg(cb.bn)
    node = _PairStackTrain if pair_stackable(ca, cb, cfga, cfgb) else _PairTrain
    ya, yb = node.apply(x, ca.conv.weight, ca.bn.weight, ca.bn.bias, cb.conv.weight, cb.bn.weight, cb.bn.bias,
                              ca.conv.stride[0], cfga, cfgb, _Into(into_b) if into_b is not None else None)
    _bn_done(ca.bn, ta)
    _bn_done(cb.bn, tb)
    return ya, yb


def stack_enabled():
    """FRLW_TRAIN_STACK=0: a pair stays two blocks whose data gradients are summed in an epilogue (_PairTrain; A/B timing)."""
    return os.environ.get("FRLW_TRAIN_STACK", "1") != "0"


def pair_stackable(ca, cb, cfga, cfgb):
    """Both blocks as ONE stacked block: same kernel size and input, channel counts multiples of 4, the same BatchNorm settings
    (eps, momentum, both or neither tracking running statistics, counters both on the device or both absent)."""
    return (stack_enabled() and ca.conv.kernel_size == cb.conv.kernel_size and ca.conv.out_channels % 4 == 0
            and cb.conv.out_channels % 4 == 0 and cfga[0] == cfgb[0] and cfga[3] == cfgb[3]
            and (cfga[1] is None) == (cfgb[1] is None) and (cfga[4] is None) == (cfgb[4] is None)
            and layer_precision(ca.conv.out_channels + cb.conv.out_channels, ca.conv.kernel_size[0], 1) == layer_precision(ca.conv.out_channels, ca.conv.kernel_size[0], 1))


def pair_eligible(x, ca, cb):
    return (fuse_enabled() and ca.training and cb.training and eligible(x, ca.conv, ca.bn, ca.act) and eligible(x, cb.conv, cb.bn, cb.act)
            and ca.conv.stride[0] == 1 and cb.conv.stride[0] == 1 and ca.conv.in_channels == cb.conv.in_channels)


def csp_join_eligible(x, csp):
    """The CSPLayer's concatenation as destination slices: conv2 and the last Bottleneck write straight into the buffer conv3
    reads (no torch.cat launch).  Every BaseConv on the way must take the native path."""
    hidden = csp.conv1.conv.out_channels
    if not (pair_eligible(x, csp.conv1, csp.conv2) and len(csp.m) >= 1 and hidden % 4 == 0 and csp.conv2.conv.out_channels == hidden):
        return False
    for blk in csp.m:
        c1, c2 = getattr(blk, "conv1", None), getattr(blk, "conv2", None)
        if c1 is None or c2 is None or not (blk.training and c1.training and c2.training and eligible(x, c1.conv, c1.bn, c1.act)
                                            and eligible(x, c2.conv, c2.bn, c2.act) and c2.conv.out_channels == hidden
                                            and c1.conv.stride[0] == 1 and c2.conv.stride[0] == 1):
            return False
    return True


def bottleneck_eligible(x, c1, c2):
    return (fuse_enabled() and c1.training and c2.training and eligible(x, c1.conv, c1.bn, c1.act) and eligible(x, c2.conv, c2.bn, c2.act)
            and c1.conv.stride[0] == 1 and c2.conv.stride[0] == 1 and c2.conv.in_channels == c1.conv.out_channels
            and c2.conv.out_channels == c1.conv.in_channels)


class _PredLevel(torch.autograd.Function):
    """cat[reg_pred(reg_feat), obj_pred(reg_feat), cls_pred(cls_feat)] of one head level (yolo_head.py:160-186, training
    branch) and its gradients in csrc/pred_ops.hip: one streaming pass forward, one backward."""

    @staticmethod
    def forward(ctx, reg_feat, cls_feat, w_reg, b_reg, w_obj, b_obj, w_cls, b_cls):
        lib = _lib.load()
        dev = reg_feat.device
        reg_feat, cls_feat = _nhwc(reg_feat.float()), _nhwc(cls_feat.float())
        B, Cc, H, W = reg_feat.shape
        nc = w_cls.shape[0]
        ws = [t.detach().float().contiguous() for t in (w_reg, b_reg, w_obj, b_obj, w_cls, b_cls)]
        out = torch.empty((B, H, W, 5 + nc), dtype=torch.float32, device=dev)
        _lib.check(lib.frlw_pred_fwd(reg_feat.data_ptr(), cls_feat.data_ptr(), B * H * W, Cc, nc, ws[0].data_ptr(), ws[1].data_ptr(),
                                     ws[2].data_ptr(), ws[3].data_ptr(), ws[4].data_ptr(), ws[5].data_ptr(), out.data_ptr(),
                                     _stream(dev)), "pred_fwd")
        ctx.save_for_backward(reg_feat, cls_feat, ws[0], ws[2], ws[4])
        ctx.shapes = (w_reg.shape, b_reg.shape, w_obj.shape, b_obj.shape, w_cls.shape, b_cls.shape)
        return out.permute(0, 3, 1, 2)  # (B, 5 + nc, H, W) view

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        reg_feat, cls_feat, w_reg, w_obj, w_cls = ctx.saved_tensors
        dev = reg_feat.device
        B, Cc, H, W = reg_feat.shape
        nc = w_cls.shape[0]
        M = B * H * W
        g = dout.float().permute(0, 2, 3, 1).contiguous()  # (B, H, W, 5 + nc) rows
        d_reg = torch.empty_like(reg_feat, memory_format=torch.channels_last)
        d_cls = torch.empty_like(cls_feat, memory_format=torch.channels_last)
        dw = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in ctx.shapes]
        n = lib.frlw_pred_bwd_scratch_floats(M, Cc, nc)
        sc = _scratch(dev, "pred", n, torch.float32)
        _lib.check(lib.frlw_pred_bwd(reg_feat.data_ptr(), cls_feat.data_ptr(), g.data_ptr(), M, Cc, nc, w_reg.data_ptr(),
                                     w_obj.data_ptr(), w_cls.data_ptr(), d_reg.data_ptr(), d_cls.data_ptr(), dw[0].data_ptr(),
                                     dw[1].data_ptr(), dw[2].data_ptr(), dw[3].data_ptr(), dw[4].data_ptr(), dw[5].data_ptr(),
                                     sc.data_ptr(), sc.numel(), _stream(dev)), "pred_bwd")
        return d_reg, d_cls, dw[0], dw[1], dw[2], dw[3], dw[4], dw[5]


class _SppPools(torch.autograd.Function):
    """cat[x, maxpool5(x), maxpool9(x), maxpool13(x)] of SPPBottleneck (network_blocks.py:139-151) and its gradient in
    csrc/pred_ops.hip (arg-max kept from the forward, gather backward: deterministic)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _nhwc(x.float())
        B, Cc, H, W = x.shape
        out = torch.empty((B, H, W, 4 * Cc), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, H, W, 3, Cc), dtype=torch.int16, device=x.device)
        _lib.check(lib.frlw_spp_train_fwd(x.data_ptr(), B, H, W, Cc, out.data_ptr(), arg.data_ptr(), _stream(x.device)), "spp_fwd")
        ctx.save_for_backward(arg)
        ctx.geom = (B, Cc, H, W)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (arg,) = ctx.saved_tensors
        B, Cc, H, W = ctx.geom
        g = _nhwc(dout.float())
        dx = torch.empty((B, H, W, Cc), dtype=torch.float32, device=g.device)
        _lib.check(lib.frlw_spp_train_bwd(g.data_ptr(), arg.data_ptr(), B, H, W, Cc, dx.data_ptr(), _stream(g.device)), "spp_bwd")
        return dx.permute(0, 3, 1, 2)


def spp_pools_eligible(x, pools):
    return (native_enabled() and x.is_cuda and x.dim() == 4 and x.shape[2] * x.shape[3] <= 512
            and [(m.kernel_size, m.stride, m.padding) for m in pools] == [(5, 1, 2), (9, 1, 4), (13, 1, 6)])


def spp_pools(x):
    return _SppPools.apply(x)


def focus_nhwc(x):
    """Focus space-to-depth (network_blocks.py:205-217) of an input that needs no gradient: (B, C, H, W) ->
    (B, 4C, H/2, W/2) logical tensor with NHWC storage, channel blocks TL, BL, TR, BR."""
    lib = _lib.load()
    x = x.contiguous()
    B, Cc, H, W = x.shape
    out = torch.empty((B, H // 2, W // 2, 4 * Cc), dtype=torch.float32, device=x.device)
    rc = lib.frlw_focus_nhwc(x.data_ptr(), B, Cc, H, W, out.data_ptr(), _stream(x.device))
    if rc == _lib.FRLW_ERR_UNSUPPORTED:  # very wide frames: the row does not fit the LDS transpose
        from .network_blocks import Focus
        return Focus.space_to_depth(x)
    _lib.check(rc, "focus_nhwc")
    return out.permute(0, 3, 1, 2)


def pred_level(reg_feat, cls_feat, reg_pred, obj_pred, cls_pred):
    """The three prediction convolutions of a level, concatenated along channels (native kernels)."""
    return _PredLevel.apply(reg_feat, cls_feat, reg_pred.weight, reg_pred.bias, obj_pred.weight, obj_pred.bias,
                            cls_pred.weight, cls_pred.bias)


def pred_eligible(reg_feat, cls_feat, reg_pred, obj_pred, cls_pred):
    c = reg_feat.shape[1]
    return (native_enabled() and reg_feat.is_cuda and reg_feat.dim() == 4 and reg_feat.shape == cls_feat.shape
            and c % 4 == 0 and c <= 512 and 1 <= cls_pred.out_channels <= 11 and reg_pred.out_channels == 4
            and obj_pred.out_channels == 1 and all(p.kernel_size == (1, 1) and p.bias is not None and p.stride == (1, 1)
                                                   for p in (reg_pred, obj_pred, cls_pred)))


def eligible(x, conv, bn, act):
    return (native_enabled() and x.is_cuda and isinstance(act, torch.nn.SiLU) and conv.bias is None
            and conv.groups == 1 and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0
            and conv.kernel_size[0] == conv.kernel_size[1] and conv.kernel_size[0] % 2 == 1
            and conv.stride[0] == conv.stride[1] and conv.stride[0] in (1, 2)
            and conv.padding[0] == (conv.kernel_size[0] - 1) // 2 and conv.dilation[0] == 1)
