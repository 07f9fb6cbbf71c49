"""YOLOX detector eval forward on MI355X: the plan builder over the gfx950 kernels of csrc/detector.hip.

``DetectorEngine(model)`` walks the reference-named module tree (``frlw_evd_amd.yolox.model.model`` =
core/model.py, backbone CSPDarknet, neck YOLOPAFPN, head YOLOXHead), folds every BatchNorm into its
convolution, re-lays the weights as (k*k*Cin, Cout) GEMM operands and records one launch per layer in a
native plan (``frlw_det_*`` in include/frlw_evd.h).  ``torch.cat`` never happens: producers write into
channel slices of the consumer's NHWC buffer.  There is no fallback: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

ACT_NONE, ACT_SILU, ACT_SIGMOID = 0, 1, 2


class View:
    """A channel slice of an NHWC buffer: (buffer index, pixel stride, channel offset, C, H, W)."""

    def __init__(self, buf, cs, co, c, h, w):
        self.buf, self.cs, self.co, self.c, self.h, self.w = buf, cs, co, c, h, w

    def slice(self, co, c):
        return View(self.buf, self.cs, self.co + co, c, self.h, self.w)


def fold_bn(conv, bn):
    """Conv2d(bias=False) + BatchNorm2d(eval) -> (weight, bias) of one biased convolution."""
    w = conv.weight.detach().double()
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    b = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    return (w * scale[:, None, None, None]).float(), b.float()


def gemm_weight(w):
    """(Cout, Cin, k, k) -> (k*k*Cin, Npad) row-major, row (ky*k + kx)*Cin + ci, Npad = Cout rounded up to 32."""
    cout, cin, k, _ = w.shape
    npad = (cout + 31) // 32 * 32
    m = torch.zeros((k * k * cin, npad), dtype=torch.float32)
    m[:, :cout] = w.permute(2, 3, 1, 0).reshape(k * k * cin, cout)
    return m.contiguous(), npad


def default_precision():
    """Convolution arithmetic of new engines: ``FRLW_CONV_PRECISION`` = ``f32`` (the default: float32 MFMA, every product
    exact -- the reference's own arithmetic) or ``bf16x3`` (opt-in: float32 products from three bf16 MFMAs on hi / lo split
    operands, ~1e-5 of the float32 result, 1.6 x the frames: include/frlw_evd.h, frlw_det_set_precision)."""
    v = os.environ.get("FRLW_CONV_PRECISION", "f32").strip().lower()
    if v not in PRECISIONS:
        raise ValueError(f"FRLW_CONV_PRECISION={v!r}: expected one of {sorted(PRECISIONS)}")
    return v


PRECISIONS = {"f32": 0, "bf16x3": 1}


class DetectorEngine:
    def __init__(self, net, device=None, precision=None):
        self.lib = _lib.load()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.net = net
        self.precision = precision or default_precision()
        self.handle = self.lib.frlw_det_create()
        _lib.check(self.lib.frlw_det_set_precision(self.handle, PRECISIONS[self.precision]), "frlw_det_set_precision")
        self._keep = []      # device weights referenced by the plan
        self._shapes = []    # per buffer: floats per image (index 0 = the NCHW input, set per call)
        self._bufs = {}      # batch size -> list of tensors
        self._built_for = None
        self.n_conv = 0
        self.flops_per_image = 0
        self.ops_meta = []   # one entry per plan op, in launch order (profiling aid)

    def __del__(self):
        try:
            if self.handle:
                self.lib.frlw_det_destroy(self.handle)
        except Exception:
            pass

    # ---- plan construction ---------------------------------------------------------------------
    def _new_buf(self, h, w, c):
        self._shapes.append(h * w * c)
        return View(len(self._shapes) - 1, c, 0, c, h, w)

    def _dev(self, t):
        t = t.to(self.device).contiguous()
        self._keep.append(t)
        return C.c_void_p(t.data_ptr())

    def _operand(self, wm):
        """The [K][Npad] float32 GEMM operand on the device, in the form the plan's arithmetic reads."""
        if self.precision == "f32" or self.device.type != "cuda":  # (a plan on "cpu" is never run: tools read its op list)
            return self._dev(wm)
        K, npad = wm.shape
        w = wm.to(self.device).contiguous()
        out = torch.empty(self.lib.frlw_conv_split_operand_bytes(K, npad), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.frlw_conv_split_operand(w.data_ptr(), K, npad, out.data_ptr(),
                                                    torch.cuda.current_stream(self.device).cuda_stream), "frlw_conv_split_operand")
        self._keep.append(out)
        return C.c_void_p(out.data_ptr())

    def _conv_raw(self, weight, bias, src, dst, k, stride, act, res=None, dst_bs=0, sig_from=0, groups=1):
        """``groups`` > 1: ``weight`` is (Cout, Cin / groups, k, k) as torch lays out a grouped convolution."""
        wm, npad = gemm_weight(weight)
        cout, cin = weight.shape[0], weight.shape[1]
        assert cin * groups == src.c and cout == dst.c and cout % groups == 0, (cin, src.c, cout, dst.c, groups)
        group_n = cout // groups if groups > 1 else 0
        rb, rcs, rco = (res.buf, res.cs, res.co) if res is not None else (-1, 0, 0)
        rc = self.lib.frlw_det_add_conv(self.handle, src.buf, src.cs, src.co, cin, src.h, src.w, self._operand(wm),
                                        self._dev(bias) if bias is not None else None, cout, npad, k, stride,
                                        dst.buf, dst.cs, dst.co, dst_bs, rb, rcs, rco, act, sig_from, group_n)
        _lib.check(rc, "frlw_det_add_conv")
        pad = (k - 1) // 2
        ho, wo = (src.h + 2 * pad - k) // stride + 1, (src.w + 2 * pad - k) // stride + 1
        assert (ho, wo) == (dst.h, dst.w), ((ho, wo), (dst.h, dst.w))
        self.n_conv += 1
        self.flops_per_image += 2 * ho * wo * cout * cin * k * k
        self.ops_meta.append(("conv", ho * wo, cout, k * k * cin, 2 * ho * wo * cout * cin * k * k))

    def _baseconv(self, bc, src, dst, res=None):
        w, b = fold_bn(bc.conv, bc.bn)
        if not isinstance(bc.act, torch.nn.SiLU):
            raise NotImplementedError("only SiLU BaseConv is on the hot path")
        self._conv_raw(w, b, src, dst, bc.conv.kernel_size[0], bc.conv.stride[0], ACT_SILU, res)

    def _csp(self, csp, src, dst):
        """CSPLayer (network_blocks.py:156-194): conv1 -> n bottlenecks || conv2, concat, conv3.

        conv1 and conv2 are both 1x1 on the same input: ONE convolution with the two weight sets stacked writes
        [conv1(x) | conv2(x)] into the concat buffer (twice the N per launch for these small, launch-bound layers).
        The bottleneck chain starts from the first half and its last convolution writes back over it (its input is
        `mid`, the residual is read and written element by element at the same address), so the buffer ends up as
        cat[m(conv1(x)), conv2(x)] without a copy."""
        hidden = csp.conv1.conv.out_channels
        cat = self._new_buf(src.h, src.w, 2 * hidden)
        w1, b1 = fold_bn(csp.conv1.conv, csp.conv1.bn)
        w2, b2 = fold_bn(csp.conv2.conv, csp.conv2.bn)
        for bc in (csp.conv1, csp.conv2):
            if not isinstance(bc.act, torch.nn.SiLU):
                raise NotImplementedError("only SiLU BaseConv is on the hot path")
        self._conv_raw(torch.cat([w1, w2], 0), torch.cat([b1, b2], 0), src, cat, 1, 1, ACT_SILU)
        n = len(csp.m)
        cur = cat.slice(0, hidden)
        for i, bott in enumerate(csp.m):
            mid = self._new_buf(src.h, src.w, bott.conv1.conv.out_channels)
            self._baseconv(bott.conv1, cur, mid)
            nxt = cat.slice(0, hidden) if i == n - 1 else self._new_buf(src.h, src.w, hidden)
            self._baseconv(bott.conv2, mid, nxt, res=cur if bott.use_add else None)
            cur = nxt
        self._baseconv(csp.conv3, cat, dst)

    def _bfm_front(self, stem, x_in, cin, H, W):
        """Temporal_Active_Focus_connect up to its BaseConv (core/Others/Temporal_Active_Focus.py:62-127):
        weight norm applied here, all 1x1 layers packed for k_bfm_stem."""
        if not isinstance(stem.act, torch.nn.SiLU):
            raise NotImplementedError("the fused BFM stem kernel implements the SiLU MLP of the shipped recipes")
        parts = []
        for conv in stem.convs:
            w = torch._weight_norm(conv.weight_v.detach(), conv.weight_g.detach(), 0)  # g * v / |v| per output channel
            parts += [w.reshape(w.shape[0], -1).float().flatten(), conv.bias.detach().float().flatten()]
        for lin in (stem.trans_up, stem.trans_down):
            parts += [lin.weight.detach().reshape(lin.weight.shape[0], -1).float().flatten(),
                      lin.bias.detach().float().flatten()]
        packed = torch.cat([p.cpu() for p in parts]).contiguous()
        want = self.lib.frlw_det_bfm_weight_count(cin)
        if want == 0:
            raise NotImplementedError(f"BFM stem with {cin} input channels (TAF K = {cin // 2}) is not supported")
        assert packed.numel() == want, (packed.numel(), want)
        er = stem.trans_down.out_channels
        f = self._new_buf(H // 2, W // 2, 4 * er)
        _lib.check(self.lib.frlw_det_add_bfm_stem(self.handle, x_in, cin, H, W, self._dev(packed), packed.numel(), f.buf),
                   "bfm stem")
        self.ops_meta.append(("bfm", H * W, 4 * er, 0, 0))
        return f

    def build(self, in_shape):
        """in_shape = (C, H, W) of one image (the network input without the trailing singleton dims)."""
        net = self.net
        bb, neck, head = net.backbone, net.neck, net.head
        cin, H, W = in_shape
        assert H % 32 == 0 and W % 32 == 0, "the detector needs H, W multiples of 32 (settings.py:22-25)"
        lib = self.lib
        self._shapes = [cin * H * W]
        x_in = 0
        # ---- backbone (darknet.py:270-354)
        if hasattr(bb.stem, "trans_up"):  # BFM stem (yolox_taf_bfm): fused per-pixel mix, Focus layout out
            f = self._bfm_front(bb.stem, x_in, cin, H, W)
        else:
            f = None
        c = bb.stem.conv.conv.out_channels
        stem = self._new_buf(H // 2, W // 2, c)
        fused = False
        if f is None:  # Focus + stem convolution as one kernel (the space-to-depth image is never written)
            sc = bb.stem.conv
            if isinstance(sc.act, torch.nn.SiLU) and sc.conv.kernel_size == (3, 3) and sc.conv.stride == (1, 1):
                w, b = fold_bn(sc.conv, sc.bn)
                wm, npad = gemm_weight(w)
                if npad == 32:
                    rc = lib.frlw_det_add_focus_stem(self.handle, x_in, cin, H, W, self._operand(wm), self._dev(b), c, stem.buf,
                                                     stem.cs, stem.co)
                    if rc != _lib.FRLW_ERR_UNSUPPORTED:
                        _lib.check(rc, "focus_stem")
                        fused = True
                        fl = 2 * (H // 2) * (W // 2) * c * 4 * cin * 9
                        self.n_conv += 1
                        self.flops_per_image += fl
                        self.ops_meta.append(("fstem", H * W // 4, c, 36 * cin, fl))
            if not fused:
                f = self._new_buf(H // 2, W // 2, 4 * cin)
                _lib.check(lib.frlw_det_add_focus(self.handle, x_in, cin, H, W, f.buf), "focus")
                self.ops_meta.append(("focus", H * W // 4, 4 * cin, 0, 0))
        if not fused:
            self._baseconv(bb.stem.conv, f, stem)
        h2, w2 = H // 4, W // 4
        d2a = self._new_buf(h2, w2, 2 * c)
        self._baseconv(bb.dark2[0], stem, d2a)
        d2 = self._new_buf(h2, w2, 2 * c)
        self._csp(bb.dark2[1], d2a, d2)
        # the three backbone outputs are written straight into the neck's concat buffers
        c3, c4, c5 = 4 * c, 8 * c, 16 * c
        h3, w3, h4, w4, h5, w5 = H // 8, W // 8, H // 16, W // 16, H // 32, W // 32
        cat_p3 = self._new_buf(h3, w3, 2 * c3)   # [upsample(fpn_out1) | dark3]
        cat_p4 = self._new_buf(h4, w4, 2 * c4)   # [upsample(fpn_out0) | dark4]
        cat_n3 = self._new_buf(h4, w4, 2 * c3)   # [bu_conv2(pan_out2) | fpn_out1]
        cat_n4 = self._new_buf(h5, w5, 2 * c4)   # [bu_conv1(pan_out1) | fpn_out0]
        d3a = self._new_buf(h3, w3, c3)
        self._baseconv(bb.dark3[0], d2, d3a)
        d3 = cat_p3.slice(c3, c3)
        self._csp(bb.dark3[1], d3a, d3)
        d4a = self._new_buf(h4, w4, c4)
        self._baseconv(bb.dark4[0], d3, d4a)
        d4 = cat_p4.slice(c4, c4)
        self._csp(bb.dark4[1], d4a, d4)
        d5a = self._new_buf(h5, w5, c5)
        self._baseconv(bb.dark5[0], d4, d5a)
        spp = bb.dark5[1]
        hid = spp.conv1.conv.out_channels
        assert [m.kernel_size for m in spp.m] == [5, 9, 13]
        sppcat = self._new_buf(h5, w5, 4 * hid)
        self._baseconv(spp.conv1, d5a, sppcat.slice(0, hid))
        _lib.check(lib.frlw_det_add_spp_pool(self.handle, sppcat.buf, sppcat.cs, hid, h5, w5), "spp")
        self.ops_meta.append(("spp", h5 * w5, hid, 0, 0))
        d5b = self._new_buf(h5, w5, c5)
        self._baseconv(spp.conv2, sppcat, d5b)
        d5 = self._new_buf(h5, w5, c5)
        self._csp(bb.dark5[2], d5b, d5)
        # ---- neck (yolo_pafpn.py:77-113)
        fpn_out0 = cat_n4.slice(c4, c4)
        self._baseconv(neck.lateral_conv0, d5, fpn_out0)
        n_ops = lib.frlw_det_num_ops(self.handle)
        _lib.check(lib.frlw_det_add_upsample(self.handle, fpn_out0.buf, fpn_out0.cs, fpn_out0.co, c4, h5, w5,
                                             cat_p4.buf, cat_p4.cs, 0), "upsample")
        if lib.frlw_det_num_ops(self.handle) > n_ops:  # (no launch of its own when the producing convolution's epilogue writes it)
            self.ops_meta.append(("upsample", h4 * w4, c4, 0, 0))
        f_out0 = self._new_buf(h4, w4, c4)
        self._csp(neck.C3_p4, cat_p4, f_out0)
        fpn_out1 = cat_n3.slice(c3, c3)
        self._baseconv(neck.reduce_conv1, f_out0, fpn_out1)
        n_ops = lib.frlw_det_num_ops(self.handle)
        _lib.check(lib.frlw_det_add_upsample(self.handle, fpn_out1.buf, fpn_out1.cs, fpn_out1.co, c3, h4, w4,
                                             cat_p3.buf, cat_p3.cs, 0), "upsample")
        if lib.frlw_det_num_ops(self.handle) > n_ops:
            self.ops_meta.append(("upsample", h3 * w3, c3, 0, 0))
        pan_out2 = self._new_buf(h3, w3, c3)
        self._csp(neck.C3_p3, cat_p3, pan_out2)
        self._baseconv(neck.bu_conv2, pan_out2, cat_n3.slice(0, c3))
        pan_out1 = self._new_buf(h4, w4, c4)
        self._csp(neck.C3_n3, cat_n3, pan_out1)
        self._baseconv(neck.bu_conv1, pan_out1, cat_n4.slice(0, c4))
        pan_out0 = self._new_buf(h5, w5, c5)
        self._csp(neck.C3_n4, cat_n4, pan_out0)
        # ---- head (yolo_head.py:162-231): the prediction convs write straight into (B, A, 5 + nc)
        nc = head.num_classes
        F = 5 + nc
        levels = [pan_out2, pan_out1, pan_out0]
        A = sum(v.h * v.w for v in levels)
        raw = self._new_buf(1, A, F)
        self.raw_buf, self.A, self.F = raw.buf, A, F
        off = 0
        import os
        lanes = os.environ.get("FRLW_DET_LANES", "0") != "0"  # head levels on side streams: measured no gain (5.17 vs 5.08 ms), off
        if lanes:
            _lib.check(lib.frlw_det_add_fork(self.handle), "fork")
        preds = []  # the levels' prediction ops are added behind the last tower: consecutive OP_PRED ops run as ONE launch
        for k, v in enumerate(levels):
            _lib.check(lib.frlw_det_set_lane(self.handle, k if (lanes and k <= 2) else 0), "lane")
            hs = self._new_buf(v.h, v.w, 256)
            self._baseconv(head.stems[k], v, hs)
            # the two towers end in the halves of one 512-channel buffer [reg_feat | cls_feat], so the three biased 1x1
            # prediction convolutions are ONE launch with a block weight matrix [[W_reg, 0], [W_obj, 0], [0, W_cls]]
            # On the two coarser levels the towers run side by side: their first convolutions read the same stem output
            # (ONE convolution with the two weight sets stacked along N), their second ones are ONE grouped convolution
            # (group g reads and writes the g-th 256 channels) -- twice the workgroups per launch, half the launches and
            # split-K reductions (36 -> 30 us per tower convolution on the 8 x 10 maps).  The finest level fills the chip
            # with 1280 workgroups per tower already; stacked it ran 2 % slower (measured), so it stays as it was.
            both = self._new_buf(v.h, v.w, 512)
            towers = (head.reg_convs[k], head.cls_convs[k])
            for t in towers:
                for bc in t:
                    if not isinstance(bc.act, torch.nn.SiLU):
                        raise NotImplementedError("only SiLU BaseConv is on the hot path")
            if k == 0:
                for tower, half in ((towers[0], both.slice(0, 256)), (towers[1], both.slice(256, 256))):
                    t1 = self._new_buf(v.h, v.w, 256)
                    self._baseconv(tower[0], hs, t1)
                    self._baseconv(tower[1], t1, half)
            else:
                t1 = self._new_buf(v.h, v.w, 512)
                f0 = [fold_bn(t[0].conv, t[0].bn) for t in towers]
                self._conv_raw(torch.cat([f0[0][0], f0[1][0]], 0), torch.cat([f0[0][1], f0[1][1]], 0), hs, t1, 3, 1, ACT_SILU)
                f1 = [fold_bn(t[1].conv, t[1].bn) for t in towers]
                self._conv_raw(torch.cat([f1[0][0], f1[1][0]], 0), torch.cat([f1[0][1], f1[1][1]], 0), t1, both, 3, 1, ACT_SILU,
                               groups=2)
            b_p = torch.cat([head.reg_preds[k].bias.detach(), head.obj_preds[k].bias.detach(),
                             head.cls_preds[k].bias.detach()], 0).float().cpu()
            w_rows = torch.cat([head.reg_preds[k].weight.detach(), head.obj_preds[k].weight.detach(),
                                head.cls_preds[k].weight.detach()], 0).float().cpu().reshape(F, 256).contiguous()
            preds.append((k, v, both, w_rows, b_p, off))
            off += v.h * v.w
        _lib.check(lib.frlw_det_set_lane(self.handle, 0), "lane")
        if lanes:
            _lib.check(lib.frlw_det_add_join(self.handle), "join")
        for k, v, both, w_rows, b_p, off_k in preds:
            rc = lib.frlw_det_add_pred(self.handle, both.buf, both.cs, both.co, 256, v.h * v.w, self._dev(w_rows), self._dev(b_p),
                                       F, raw.buf, off_k, A * F)
            if rc == _lib.FRLW_ERR_UNSUPPORTED:  # very many classes: the block-matrix convolution [[W_reg, 0], [W_obj, 0], [0, W_cls]]
                w_p = torch.zeros((F, 512, 1, 1), dtype=torch.float32)
                w_p[0:4, 0:256] = head.reg_preds[k].weight.detach().float().cpu()
                w_p[4:5, 0:256] = head.obj_preds[k].weight.detach().float().cpu()
                w_p[5:F, 256:512] = head.cls_preds[k].weight.detach().float().cpu()
                dst = View(raw.buf, F, off_k * F, F, v.h, v.w)
                self._conv_raw(w_p, b_p, both, dst, 1, 1, ACT_SIGMOID, dst_bs=A * F, sig_from=4)
            else:
                _lib.check(rc, "frlw_det_add_pred")
                self.flops_per_image += 2 * v.h * v.w * F * 256
                # (consecutive prediction ops are ONE launch: one row of the per-launch table for all of them)
                if self.ops_meta and self.ops_meta[-1][0] == "pred" and k > preds[0][0]:
                    last = self.ops_meta[-1]
                    self.ops_meta[-1] = ("pred", last[1] + v.h * v.w, F, 512, last[4] + 2 * v.h * v.w * F * 256)
                else:
                    self.ops_meta.append(("pred", v.h * v.w, F, 512, 2 * v.h * v.w * F * 256))
        self.n_forward_ops = lib.frlw_det_num_ops(self.handle)
        # ---- decode + NMS (yolo_head.py:258-303)
        self.dec_buf = self._new_buf(1, A, F).buf
        self.dets_buf = self._new_buf(1, A, 6).buf
        self.counts_buf = self._new_buf(1, 1, 1 + A).buf  # per image: [count, scratch of A ints]
        # per image: candidate count, sorted corner boxes, the "i suppresses j" bit matrix (k_nms_matrix / k_nms_sweep)
        self.nms_buf = self._new_buf(1, 1, int(lib.frlw_det_nms_workspace_floats(A))).buf
        n = len(levels)
        arr = C.c_int * n
        _lib.check(lib.frlw_det_add_decode_nms(self.handle, raw.buf, A, nc, n, arr(*[v.h for v in levels]),
                                               arr(*[v.w for v in levels]), arr(*[int(s) for s in head.strides]),
                                               C.c_float(head.obj_threshold), C.c_float(head.nms_threshold),
                                               self.dec_buf, self.dets_buf, self.counts_buf, self.nms_buf), "decode")
        head.hw = [(v.h, v.w) for v in levels]
        # split-K scratch: 8 splits x (< 256 tiles of 64 x 64), independent of the batch size
        self.scratch_floats = 8 * 256 * 64 * 64 + 1024
        self._shapes.append(0)
        self.scratch_buf = len(self._shapes) - 1
        _lib.check(lib.frlw_det_set_scratch(self.handle, self.scratch_buf, self.scratch_floats), "scratch")
        self._built_for = tuple(in_shape)

    # ---- execution -----------------------------------------------------------------------------
    def _buffers(self, B):
        bufs = self._bufs.get(B)
        if bufs is None:
            bufs = [None] + [torch.empty(B * n, dtype=torch.float32, device=self.device) for n in self._shapes[1:]]
            bufs[self.counts_buf] = torch.zeros(B * (1 + self.A), dtype=torch.int32, device=self.device)
            # (zeros: the last 1024 words of every lane's region are the arrival counters of the in-kernel split-K reduction)
            bufs[self.scratch_buf] = torch.zeros(3 * self.scratch_floats, dtype=torch.float32, device=self.device)
            self._bufs[B] = bufs
        return bufs

    def _run(self, x, first, last):
        if not x.is_cuda:
            raise RuntimeError("DetectorEngine needs a ROCm tensor: there is no CPU path in the engine")
        if x.dim() == 5:  # (B, C, H, W, 1): the singleton Focus drops (network_blocks.py:221)
            x = x[..., 0]
        x = x.contiguous().float()
        B = x.shape[0]
        if self._built_for is None:
            self.build(tuple(x.shape[1:]))
        assert tuple(x.shape[1:]) == self._built_for, "input shape differs from the one the plan was built for"
        bufs = self._buffers(B)
        ptrs = (C.c_void_p * len(bufs))(*[C.c_void_p(x.data_ptr())] + [C.c_void_p(t.data_ptr()) for t in bufs[1:]])
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(self.lib.frlw_det_run(self.handle, B, ptrs, len(bufs), first, last, stream), "frlw_det_run")
        self._last_input = x  # keep alive until the stream has consumed it
        return bufs, B

    def raw_outputs(self, x):
        """(B, A, 5 + nc) = cat[reg, sigmoid(obj), sigmoid(cls)] per anchor (the pre-NMS tensor)."""
        if self._built_for is None:
            self.build(tuple((x[..., 0] if x.dim() == 5 else x).shape[1:]))
        bufs, B = self._run(x, 0, self.n_forward_ops)
        return bufs[self.raw_buf].view(B, self.A, self.F)

    def _nms_large(self, dec):
        """yolo_head.py:276-301 for one image from its decoded (A, 5 + nc) rows."""
        from .yolox.yolo_head import nms_reference
        head = self.net.head
        o = dec[dec[:, 4] > head.obj_threshold]
        xyxy = torch.cat([o[:, 0:1] - o[:, 2:3] / 2, o[:, 1:2] - o[:, 3:4] / 2, o[:, 0:1] + o[:, 2:3] / 2,
                          o[:, 1:2] + o[:, 3:4] / 2], dim=-1)
        o = o[nms_reference(xyxy, o[:, 4], head.nms_threshold)]
        nc = head.num_classes
        return torch.cat([o[:, 0:4], torch.argmax(o[:, 5:5 + nc], 1)[:, None].to(o.dtype),
                          (o[:, 4] * torch.max(o[:, 5:5 + nc], 1)[0])[:, None]], dim=1)

    def detect(self, x, return_decoded=False):
        """Full eval forward: list of (n_i, 6) [cx, cy, w, h, cls, obj * max cls] per image."""
        bufs, B = self._run(x, 0, -1)
        counts = bufs[self.counts_buf].view(B, 1 + self.A)[:, 0].cpu().tolist()  # the reference loops on the host too
        # ONE copy out of the engine's (reused) buffer for the whole batch -- the rows up to the largest count -- and per-image
        # views of it (a clone per image was 32 launches of 5 us behind every forward)
        top = max(max(counts), 1)
        dets = bufs[self.dets_buf].view(B, self.A, 6)[:, :top].clone()
        out = []
        for b, n in enumerate(counts):
            if n < 0:
                # more than 8192 candidates (what k_decode_sort holds in LDS): only possible for inputs beyond the
                # 1 Mpx detector shape (6720 anchors).  Such an image takes the box-by-box procedure on the decoded
                # rows the kernel left in HBM (ROCm tensors, torch ops; same arithmetic and visiting order)
                out.append(self._nms_large(bufs[self.dec_buf].view(B, self.A, self.F)[b]))
                continue
            out.append(dets[b, :n] if n > 0 else torch.zeros((1, 6), device=self.device))
        if return_decoded:
            return out, bufs[self.dec_buf].view(B, self.A, self.F)
        return out
