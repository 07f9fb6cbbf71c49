"""Training branch of the YOLOX head: SimOTA label assignment and the losses
(reference: core/yolox/models/yolo_head.py:237-256,305-707, core/yolox/models/losses.py:9-53,
core/yolox/utils/boxes.py:79-102).

Three forms of the same loss: the reference's per-image procedure with plain torch ops (``get_assignments`` -- CPU tensors,
and the yardstick of the GPU tests), the batched one (native SimOTA + masked torch terms, ``yolox_losses_batched``) and the
native one (``_YoloxLoss``: decode, assignment, terms and gradient in csrc/simota.hip), which is what ROCm tensors take.
Dtype behaviour is kept in all three: labels arrive as float64 (data/dataset.py:216), so IoUs, costs and the total loss
are float64 while predictions stay float32.
"""
import torch
import torch.nn.functional as F


def bboxes_iou_cxcywh(a, b):
    """Pairwise IoU of (N, 4) and (M, 4) boxes in cxcywh (boxes.py:79-102 with xyxy=False)."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    area_a = a[:, 2] * a[:, 3]  # == torch.prod(a[:, 2:], 1), without prod's cumulative-product backward
    area_b = b[:, 2] * b[:, 3]
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def iou_loss(pred, target):
    """1 - IoU^2 per box, cxcywh (losses.py:16-36, loss_type "iou")."""
    pred = pred.view(-1, 4)
    target = target.view(-1, 4)
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p = pred[:, 2] * pred[:, 3]  # == torch.prod(pred[:, 2:], 1) (boxes are (n, 4)); prod's backward scans
    area_g = target[:, 2] * target[:, 3]
    en = (tl < br).type(tl.type()).prod(dim=1)
    d = br - tl
    area_i = d[:, 0] * d[:, 1] * en
    iou = area_i / (area_p + area_g - area_i + 1e-16)
    return 1 - iou ** 2


_GRIDS = {}


def _level_grid(h, w, stride, device, dtype):
    """((1, h*w, 2) cell grid, (1, h*w) stride row) of one level, built once per (shape, device) and kept: the reference
    rebuilds both on the host and uploads them every step (yolo_head.py:242-246) -- two host-to-device copies per level
    that also keep the step from being captured into a HIP graph.  Constants: never written."""
    key = (h, w, float(stride), str(device), dtype)
    hit = _GRIDS.get(key)
    if hit is None:
        yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, h * w, 2).to(device=device, dtype=dtype)
        hit = _GRIDS[key] = (grid, torch.zeros(1, h * w).fill_(stride).to(device=device, dtype=dtype))
    return hit


def output_and_grid(output, stride):
    """(B, 5 + nc, h, w) raw level output -> decoded (B, h*w, 5 + nc) and its (1, h*w, 2) grid
    (yolo_head.py:237-256): xy = (xy + grid) * stride, wh = square(wh) * stride."""
    B, n_ch, h, w = output.shape
    grid = _level_grid(h, w, stride, output.device, output.dtype)[0]
    out = output.view(B, 1, n_ch, h, w).permute(0, 1, 3, 4, 2).reshape(B, h * w, n_ch)
    xy = (out[..., :2] + grid) * stride
    wh = torch.square(out[..., 2:4]) * stride
    return torch.cat([xy, wh, out[..., 4:]], dim=-1), grid


def in_boxes_info(gt, strides, x_shifts, y_shifts, radius):
    """Candidate anchors = centre inside a GT box OR inside the radius*stride square around its centre
    (yolo_head.py:586-669).  Returns (candidate mask (A,), in-box-AND-in-centre (G, n_candidates))."""
    s = strides[0]
    xc = (x_shifts[0] * s + 0.5 * s).unsqueeze(0)  # (1, A)
    yc = (y_shifts[0] * s + 0.5 * s).unsqueeze(0)
    cx, cy, w, h = (gt[:, i].unsqueeze(1) for i in range(4))  # (G, 1)
    deltas = torch.stack([xc - (cx - 0.5 * w), yc - (cy - 0.5 * h), (cx + 0.5 * w) - xc, (cy + 0.5 * h) - yc], 2)
    is_in_boxes = deltas.min(dim=-1).values > 0.0
    r = radius * s.unsqueeze(0)
    cdeltas = torch.stack([xc - (cx - r), yc - (cy - r), (cx + r) - xc, (cy + r) - yc], 2)
    is_in_centers = cdeltas.min(dim=-1).values > 0.0
    anchor = (is_in_boxes.sum(dim=0) > 0) | (is_in_centers.sum(dim=0) > 0)
    return anchor, is_in_boxes[:, anchor] & is_in_centers[:, anchor]


def dynamic_k_matching(cost, ious, gt_classes, fg_mask):
    """SimOTA (yolo_head.py:671-707): per GT k = clamp(floor(sum of its top-10 IoUs), 1) lowest-cost anchors;
    an anchor claimed by several GTs goes to the cheapest.  Updates fg_mask in place."""
    num_gt = cost.shape[0]
    matching = torch.zeros_like(cost)
    topk_ious, _ = torch.topk(ious, min(10, ious.size(1)), dim=1)
    dynamic_ks = torch.clamp(topk_ious.sum(1).int(), min=1)
    for g in range(num_gt):
        _, pos = torch.topk(cost[g], k=dynamic_ks[g].item(), largest=False)
        matching[g][pos] = 1.0
    multi = matching.sum(0) > 1
    if multi.sum() > 0:
        _, argmin = torch.min(cost[:, multi], dim=0)
        matching[:, multi] *= 0.0
        matching[argmin, multi] = 1.0
    fg_in = matching.sum(0) > 0.0
    num_fg = fg_in.sum().item()
    fg_mask[fg_mask.clone()] = fg_in
    matched_gt = matching[:, fg_in].argmax(0)
    return num_fg, gt_classes[matched_gt], (matching * ious).sum(0)[fg_in], matched_gt


@torch.no_grad()
def get_assignments(b, gt_boxes, gt_classes, preds_b, strides, x_shifts, y_shifts, cls_preds, obj_preds,
                    num_classes, radius):
    """yolo_head.py:482-584."""
    num_gt = gt_boxes.shape[0]
    fg_mask, in_box_and_center = in_boxes_info(gt_boxes, strides, x_shifts, y_shifts, radius)
    boxes = preds_b[fg_mask]
    cls_ = cls_preds[b][fg_mask]
    obj_ = obj_preds[b][fg_mask]
    n_in = boxes.shape[0]
    ious = bboxes_iou_cxcywh(gt_boxes, boxes)
    gt_onehot = F.one_hot(gt_classes.to(torch.int64), num_classes).float().unsqueeze(1).repeat(1, n_in, 1)
    iou_cost = -torch.log(ious + 1e-8)
    joint = (cls_.float().unsqueeze(0).repeat(num_gt, 1, 1).sigmoid_()
             * obj_.float().unsqueeze(0).repeat(num_gt, 1, 1).sigmoid_())
    cls_cost = F.binary_cross_entropy(joint.sqrt_(), gt_onehot, reduction="none").sum(-1)
    cost = cls_cost + 3.0 * iou_cost + 100000.0 * (~in_box_and_center)
    num_fg, matched_classes, matched_ious, matched_gt = dynamic_k_matching(cost, ious, gt_classes, fg_mask)
    return matched_classes, fg_mask, matched_ious, matched_gt, num_fg


_SIMOTA_WS = {}
_FORCE_LOOP = False  # tests flip this to compare the batched path with the per-image procedure on the GPU


@torch.no_grad()
def simota_assign(outputs, labels, x_shifts, y_shifts, strides_all, num_classes, radius):
    """Whole-batch SimOTA on the GPU (frlw_simota_assign, csrc/simota.hip): no per-image Python, no sync.

    Returns fg (B, A) bool, matched_gt (B, A) int32 (-1 = background), matched_iou (B, A) float64,
    num_fg (B) int32, nlabel (B) int32 -- all device tensors."""
    import ctypes as C
    from .. import _lib
    lib = _lib.load()
    B, A, P = outputs.shape
    G = labels.shape[1]
    dev = outputs.device
    preds = outputs.detach().float().contiguous()
    lab = labels.detach().to(device=dev, dtype=torch.float64).contiguous()
    xs = x_shifts.reshape(-1).float().contiguous()
    ys = y_shifts.reshape(-1).float().contiguous()
    st = strides_all.reshape(-1).float().contiguous()
    fg = torch.empty((B, A), dtype=torch.uint8, device=dev)
    matched_gt = torch.empty((B, A), dtype=torch.int32, device=dev)
    matched_iou = torch.empty((B, A), dtype=torch.float64, device=dev)
    num_fg = torch.empty((B,), dtype=torch.int32, device=dev)
    nlabel = torch.empty((B,), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    need = lib.frlw_simota_workspace_bytes(B, A, G)
    key = (dev.index, stream)
    ws = _SIMOTA_WS.get(key)
    if ws is None or ws.numel() < need:
        from .. import _pins
        _pins.retire(ws)  # a live HIP graph may still launch kernels on the old workspace
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _SIMOTA_WS[key] = ws
    _lib.check(lib.frlw_simota_assign(preds.data_ptr(), lab.data_ptr(), xs.data_ptr(), ys.data_ptr(), st.data_ptr(),
                                      B, A, G, num_classes, C.c_float(radius), fg.data_ptr(), matched_gt.data_ptr(),
                                      matched_iou.data_ptr(), num_fg.data_ptr(), nlabel.data_ptr(), ws.data_ptr(),
                                      ws.numel(), stream), "frlw_simota_assign")
    return fg.bool(), matched_gt, matched_iou, num_fg, nlabel


def yolox_losses_batched(outputs, x_shifts, y_shifts, strides_all, labels, num_classes, radius):
    """The loss of ``yolox_losses`` with the assignment done by ``simota_assign`` and the three terms summed
    over all anchors under the foreground mask instead of over gathered rows: same value (up to the order of
    the float sums), no host synchronisation.  ROCm tensors only."""
    B, A, _ = outputs.shape
    bbox_preds = outputs[:, :, :4]
    obj_preds = outputs[:, :, 4:5]
    cls_preds = outputs[:, :, 5:]
    fg, matched_gt, matched_iou, num_fg_b, nlabel = simota_assign(outputs, labels, x_shifts, y_shifts, strides_all,
                                                                  num_classes, radius)
    labels = labels.to(device=outputs.device, dtype=torch.float64)
    gi = matched_gt.clamp(min=0).long()
    reg_t = torch.gather(labels[:, :, 1:5], 1, gi.unsqueeze(-1).expand(-1, -1, 4))            # (B, A, 4) f64
    cls_id = torch.gather(labels[:, :, 0], 1, gi).to(torch.int64).clamp(0, num_classes - 1)  # (B, A)
    cls_t = F.one_hot(cls_id, num_classes) * matched_iou.unsqueeze(-1)                        # f64, yolo_head.py:383-385
    obj_t = fg.unsqueeze(-1).to(outputs.dtype)
    num_fg = num_fg_b.sum().clamp(min=1).to(torch.float64)
    bce = torch.nn.BCEWithLogitsLoss(reduction="none")
    zero = torch.zeros((), dtype=torch.float64, device=outputs.device)
    l_iou = iou_loss(bbox_preds.reshape(-1, 4), reg_t.reshape(-1, 4)).view(B, A)
    loss_iou = torch.where(fg, l_iou, zero).sum() / num_fg
    loss_obj = bce(obj_preds.reshape(-1, 1), obj_t.reshape(-1, 1)).sum() / num_fg
    l_cls = bce(cls_preds, cls_t)
    loss_cls = torch.where(fg.unsqueeze(-1), l_cls, zero.to(l_cls.dtype)).sum() / num_fg
    reg_weight = 5.0
    loss = reg_weight * loss_iou + loss_obj + loss_cls + 0.0
    return loss, reg_weight * loss_iou, loss_obj, loss_cls, 0.0, num_fg / nlabel.sum().clamp(min=1)


_LOSS_WS = {}
_FORCE_TORCH_LOSS = False  # tests flip this to compare the native loss (value and gradients) with the autograd one above


def _level_arrays(rows, strides):
    """HOST arrays of frlw_yolox_loss_*: device pointers, grid shapes and strides of the levels."""
    import ctypes as C
    n = len(rows)
    return ((C.c_void_p * n)(*[r.data_ptr() for r in rows]), (C.c_int32 * n)(*[r.shape[1] for r in rows]),
            (C.c_int32 * n)(*[r.shape[2] for r in rows]), (C.c_float * n)(*[float(s) for s in strides]))


class _YoloxLoss(torch.autograd.Function):
    """``get_losses`` (yolo_head.py:305-473) as one native forward and one native backward (csrc/simota.hip,
    frlw_yolox_loss_fwd / _bwd): returns the (6,) float64 tensor {loss, 5 * loss_iou, loss_obj, loss_cls, num_fg / num_gt, num_fg}."""

    @staticmethod
    def forward(ctx, labels, strides, num_classes, radius, *levels):
        import ctypes as C
        from .. import _lib, _pins
        lib = _lib.load()
        dev = levels[0].device
        # (B, h, w, 5 + nc) rows: what train_ops.pred_level produced (no copy); NCHW outputs of the torch convolutions are
        # re-laid once
        rows = [o.detach().float().permute(0, 2, 3, 1).contiguous() for o in levels]
        B, P = rows[0].shape[0], rows[0].shape[3]
        A = sum(r.shape[1] * r.shape[2] for r in rows)
        lab = labels.detach().to(device=dev, dtype=torch.float64).contiguous()
        G = lab.shape[1]
        preds = torch.empty((B, A, P), dtype=torch.float32, device=dev)
        fg = torch.empty((B, A), dtype=torch.uint8, device=dev)
        matched_gt = torch.empty((B, A), dtype=torch.int32, device=dev)
        matched_iou = torch.empty((B, A), dtype=torch.float64, device=dev)
        result = torch.empty((6,), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        need = lib.frlw_yolox_loss_workspace_bytes(B, A, G)
        key = (dev.index, stream)
        ws = _LOSS_WS.get(key)
        if ws is None or ws.numel() < need:
            _pins.retire(ws)  # a live HIP graph may still launch kernels on the old workspace
            ws = _LOSS_WS[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        raw, hs, wd, st = _level_arrays(rows, strides)
        _lib.check(lib.frlw_yolox_loss_fwd(raw, hs, wd, st, len(rows), B, num_classes, lab.data_ptr(), G, C.c_float(radius),
                                           preds.data_ptr(), fg.data_ptr(), matched_gt.data_ptr(), matched_iou.data_ptr(),
                                           result.data_ptr(), ws.data_ptr(), ws.numel(), stream), "frlw_yolox_loss_fwd")
        ctx.save_for_backward(lab, fg, matched_gt, matched_iou, result, *rows)
        ctx.meta = (tuple(float(s) for s in strides), num_classes)
        return result

    @staticmethod
    def backward(ctx, grad):
        import ctypes as C
        from .. import _lib
        lib = _lib.load()
        lab, fg, matched_gt, matched_iou, result, *rows = ctx.saved_tensors
        strides, num_classes = ctx.meta
        dev = rows[0].device
        grad = grad.detach().to(torch.float64).contiguous()
        grads = [torch.empty_like(r) for r in rows]
        raw, hs, wd, st = _level_arrays(rows, strides)
        out = (C.c_void_p * len(rows))(*[g.data_ptr() for g in grads])
        _lib.check(lib.frlw_yolox_loss_bwd(raw, hs, wd, st, len(rows), rows[0].shape[0], num_classes, lab.data_ptr(),
                                           lab.shape[1], fg.data_ptr(), matched_gt.data_ptr(), matched_iou.data_ptr(),
                                           result.data_ptr(), grad.data_ptr(), out,
                                           torch.cuda.current_stream(dev).cuda_stream), "frlw_yolox_loss_bwd")
        return (None, None, None, None, *[g.permute(0, 3, 1, 2) for g in grads])


NATIVE_MAX_ANCHORS = 150 * 1024 // 16  # 9600: what frlw_simota_assign / frlw_yolox_loss_fwd accept (FRLW_ERR_UNSUPPORTED above)


def yolox_losses_native(level_outputs, strides, labels, num_classes, radius):
    """The tuple of ``yolox_losses`` from the native loss (ROCm tensors): seven launches forward, one backward."""
    res = _YoloxLoss.apply(labels, tuple(strides), int(num_classes), float(radius), *level_outputs)
    return res[0], res[1], res[2], res[3], 0.0, res[4]


def yolox_losses(level_outputs, strides, labels, num_classes, radius):
    """``get_losses`` (yolo_head.py:305-473) on the raw per-level outputs cat[reg, obj, cls] (B, 5 + nc, h, w).

    Returns (loss, 5 * loss_iou, loss_obj, loss_cls, loss_l1 = 0.0, num_fg / num_gt).  On ROCm tensors the whole
    loss -- decode, assignment, the three terms and their gradient -- runs in the HIP library (``_YoloxLoss``); on CPU
    tensors -- the gloo tests and the golden checks -- it is the reference's per-image procedure."""
    n_anchors = sum(int(o.shape[2]) * int(o.shape[3]) for o in level_outputs)
    # (the native assignment keeps an image's IoU and cost rows of one box in LDS: 2 * A * 8 bytes <= 150 KB, csrc/simota.hip; a
    # 720 x 1280 input has 18 900 anchors -- such shapes take the reference's per-image procedure below instead of aborting)
    fits = n_anchors <= NATIVE_MAX_ANCHORS
    if level_outputs[0].is_cuda and not (_FORCE_LOOP or _FORCE_TORCH_LOSS) and len(level_outputs) <= 4 and fits:
        return yolox_losses_native(level_outputs, strides, labels, num_classes, radius)
    outs, xs, ys, ss = [], [], [], []
    for o, stride in zip(level_outputs, strides):
        dec, grid = output_and_grid(o, stride)
        outs.append(dec)
        xs.append(grid[:, :, 0])
        ys.append(grid[:, :, 1])
        ss.append(_level_grid(o.shape[2], o.shape[3], stride, o.device, o.dtype)[1])
    outputs = torch.cat(outs, 1)
    x_shifts, y_shifts, strides_all = torch.cat(xs, 1), torch.cat(ys, 1), torch.cat(ss, 1)
    if outputs.is_cuda and not _FORCE_LOOP and fits:
        return yolox_losses_batched(outputs, x_shifts, y_shifts, strides_all, labels, num_classes, radius)
    bbox_preds = outputs[:, :, :4]
    obj_preds = outputs[:, :, 4].unsqueeze(-1)
    cls_preds = outputs[:, :, 5:]
    nlabel = (labels.sum(dim=2) > 0).sum(dim=1)
    A = outputs.shape[1]
    cls_t, reg_t, obj_t, fg_masks = [], [], [], []
    num_fg, num_gts = 0.0, 0.0
    for b in range(outputs.shape[0]):
        num_gt = int(nlabel[b])
        num_gts += num_gt
        if num_gt == 0:
            cls_t.append(outputs.new_zeros((0, num_classes)))
            reg_t.append(outputs.new_zeros((0, 4)))
            obj_t.append(outputs.new_zeros((A, 1)))
            fg_masks.append(outputs.new_zeros(A).bool())
            continue
        gt_boxes = labels[b, :num_gt, 1:5]
        gt_classes = labels[b, :num_gt, 0]
        matched_classes, fg_mask, matched_ious, matched_gt, n_fg = get_assignments(
            b, gt_boxes, gt_classes, bbox_preds[b], strides_all, x_shifts, y_shifts, cls_preds, obj_preds,
            num_classes, radius)
        num_fg += n_fg
        cls_t.append(F.one_hot(matched_classes.to(torch.int64), num_classes) * matched_ious.unsqueeze(-1))
        obj_t.append(fg_mask.unsqueeze(-1).to(outputs.dtype))
        reg_t.append(gt_boxes[matched_gt])
        fg_masks.append(fg_mask)
    cls_t, reg_t, obj_t, fg_masks = torch.cat(cls_t, 0), torch.cat(reg_t, 0), torch.cat(obj_t, 0), torch.cat(fg_masks, 0)
    num_fg = max(num_fg, 1)
    bce = torch.nn.BCEWithLogitsLoss(reduction="none")
    loss_iou = iou_loss(bbox_preds.reshape(-1, 4)[fg_masks], reg_t).sum() / num_fg
    loss_obj = bce(obj_preds.reshape(-1, 1), obj_t).sum() / num_fg
    loss_cls = bce(cls_preds.reshape(-1, num_classes)[fg_masks], cls_t).sum() / num_fg
    reg_weight = 5.0
    loss = reg_weight * loss_iou + loss_obj + loss_cls + 0.0
    return loss, reg_weight * loss_iou, loss_obj, loss_cls, 0.0, num_fg / max(num_gts, 1)
