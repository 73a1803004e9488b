"""The supervised loss as two kernel launches around the per-scene IoU kernel (binding of
include/loss_hip.h; csrc/votenet_loss.hip, csrc/loss_core.h).

`get_labeled_loss_fused` fills the same end_points keys as votenet/losses.py:get_labeled_loss
(the mirror of models/loss_helper_labeled.py:300-370) from one statistics vector, and the gradient
of the loss with respect to every head output comes out of the same launch: the autograd node
below just hands those buffers over, scaled by the incoming gradient of `loss`.  Only
`end_points['loss']` (= 'detection_loss') is differentiable; the other entries are logging values.
"""
import ctypes
import importlib
import os

import torch

_c_int, _ll, _vp = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p


class VnLossTensor(ctypes.Structure):
    _fields_ = [("p", _vp), ("sb", _ll), ("sk", _ll), ("sc", _ll), ("sd", _ll)]


_PRED = ("agg_xyz", "obj", "center", "h_scores", "h_resn", "s_scores", "s_resn", "sem", "iou",
         "iou_jit", "seed_xyz", "vote_xyz", "jit_center", "jit_size", "jit_heading")
_GRADS = ("g_obj", "g_center", "g_h_scores", "g_h_resn", "g_s_scores", "g_s_resn", "g_sem", "g_iou",
          "g_iou_jit", "g_vote")


class VnLossArgs(ctypes.Structure):  # field order == include/loss_hip.h
    _fields_ = ([(n, _c_int) for n in ("B", "K", "G", "S", "VF", "N", "NH", "NS", "NC", "NI",
                                       "has_jitter", "consistency")] + [("grad_scale", ctypes.c_float)] +
                [(n, _vp) for n in ("center_label", "box_label_mask", "heading_class_label",
                                    "heading_residual_label", "size_class_label",
                                    "size_residual_label", "sem_cls_label", "vote_label",
                                    "vote_label_mask", "seed_inds")] +
                [("seed_inds_stride", _ll), ("mean_size", _vp)] +
                [(n, VnLossTensor) for n in _PRED] +
                [(n, _vp) for n in ("boxes", "gt_boxes", "iou_lab", "iou_assign", "stats",
                                    "objectness_label", "objectness_mask", "object_assignment")] +
                [(n, _vp) for n in _GRADS] + [("gt_nearest", _vp), ("partials", _vp)])


(ST_LOSS, ST_VOTE, ST_OBJ, ST_CENTER, ST_HCLS, ST_HREG, ST_SCLS, ST_SREG, ST_SEM, ST_BOX, ST_IOU,
 ST_JIT, ST_POS_RATIO, ST_NEG_RATIO, ST_OBJ_ACC, ST_OBJ_COUNT, ST_CLS_ACC, ST_PRED_IOU,
 ST_PRED_IOU_OBJ, ST_IOU_ACC, ST_IOU_ACC_OBJ, ST_JIT_ACC, ST_JIT_ACC_OBJ, ST_COUNT) = range(24)

_STAT_KEYS = {
    'vote_loss': ST_VOTE, 'objectness_loss': ST_OBJ, 'center_loss': ST_CENTER,
    'heading_cls_loss': ST_HCLS, 'heading_reg_loss': ST_HREG, 'size_cls_loss': ST_SCLS,
    'size_reg_loss': ST_SREG, 'sem_cls_loss': ST_SEM, 'box_loss': ST_BOX, 'iou_loss': ST_IOU,
    'pos_ratio': ST_POS_RATIO, 'neg_ratio': ST_NEG_RATIO, 'obj_acc': ST_OBJ_ACC,
    'obj_count': ST_OBJ_COUNT, 'cls_acc': ST_CLS_ACC, 'pred_iou_value': ST_PRED_IOU,
    'pred_iou_obj_value': ST_PRED_IOU_OBJ, 'iou_acc': ST_IOU_ACC, 'iou_acc_obj': ST_IOU_ACC_OBJ,
}
_JITTER_KEYS = {'jitter_iou_loss': ST_JIT, 'jitter_iou_acc': ST_JIT_ACC,
                'jitter_iou_acc_obj': ST_JIT_ACC_OBJ}


def enabled():
    return os.environ.get("VOTENET_FUSED_LOSS", "1") != "0"


_HOST_BUILD = None  # tests: ctypes handle of tests/loss_host.cpp (same arithmetic, host pointers)


def available(device):
    return device.type == "cuda" or _HOST_BUILD is not None


def _launch(name, args, device):
    if device.type != "cuda":
        if _HOST_BUILD is None:
            raise RuntimeError("the fused loss runs on the GPU only (no CPU path)")
        rc = getattr(_HOST_BUILD, name.replace("votenet_loss", "host_loss"))(ctypes.byref(args))
        assert rc == 0
        return
    _L = importlib.import_module("3dioumatch_amd._lib")
    with torch.cuda.device(device):
        _L.check(getattr(_L.lib, name)(ctypes.byref(args), _L.current_stream_ptr(device)), name)


def _scratch_floats(args, device):
    if device.type != "cuda":
        if _HOST_BUILD is None:
            raise RuntimeError("the fused loss runs on the GPU only (no CPU path)")
        return int(_HOST_BUILD.host_loss_scratch_floats(ctypes.byref(args)))
    _L = importlib.import_module("3dioumatch_amd._lib")
    return int(_L.lib.votenet_loss_scratch_floats(ctypes.byref(args)))


def _scene_iou(boxes, gt_boxes):
    """(best IoU (B,P) f32, first best same-scene GT index (B,P) i32)."""
    if boxes.is_cuda:
        cuda = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_cuda")
        best = torch.empty(boxes.shape[:2], dtype=torch.float32, device=boxes.device)
        idx = torch.empty(boxes.shape[:2], dtype=torch.int32, device=boxes.device)
        cuda.scene_best_iou3d_gpu(boxes, gt_boxes, best, idx)
        return best, idx
    from .losses import _scene_best_iou
    best, idx = _scene_best_iou(boxes, gt_boxes)
    return best.float().contiguous(), idx.int().contiguous()


def _view(t):
    s = list(t.stride()) + [0] * (4 - t.dim())
    return VnLossTensor(t.data_ptr(), s[0], s[1], s[2] if t.dim() > 2 else 0, s[3] if t.dim() > 3 else 0)


def _label(end_points, key, dtype):
    t = end_points[key]
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s" % (key, dtype))
    return t.contiguous()


class _FusedLabeledLoss(torch.autograd.Function):
    """stats, labels <- launches; backward: the stored gradients times d(loss)."""

    @staticmethod
    def forward(ctx, end_points, config, nb, obj, center, h_scores, h_resn, s_scores, s_resn, sem,
                iou, iou_jit, vote_xyz):
        out = _labeled_pass(end_points, config, nb, obj, center, h_scores, h_resn, s_scores, s_resn,
                            sem, iou, iou_jit, vote_xyz)
        stats, objectness_label, objectness_mask, object_assignment, pred_bbox, extra = out
        ctx.flat, ctx.shapes, ctx.sizes, ctx.has_jitter = extra
        ctx.mark_non_differentiable(objectness_label, objectness_mask, object_assignment, pred_bbox)
        return stats, objectness_label, objectness_mask, object_assignment, pred_bbox

    @staticmethod
    def backward(ctx, g_stats, *unused):
        scaled = ctx.flat * g_stats[ST_LOSS]
        out, off = [], 0
        for sh, n in zip(ctx.shapes, ctx.sizes):
            out.append(scaled[off:off + n].view(sh) if n else None)
            off += n
        g_obj, g_center, g_hs, g_hr, g_ss, g_sr, g_sem, g_iou, g_jit, g_vote = out
        return (None, None, None, g_obj, g_center, g_hs, g_hr, g_ss, g_sr, g_sem, g_iou,
                g_jit if ctx.has_jitter else None, g_vote)


def _labeled_pass(end_points, config, nb, obj, center, h_scores, h_resn, s_scores, s_resn, sem, iou,
                  iou_jit, vote_xyz, grad_dest=None, grad_scale=1.0):
    """The supervised loss' launches on the first nb scenes.  grad_dest: {name of _GRADS: device
    address} to write the gradient rows somewhere the caller owns (a buffer shared with the
    consistency loss); None: a buffer of its own (returned in `extra`)."""
    dev = center.device
    k, g = center.shape[1], end_points['center_label'].shape[1]
    s, vf = end_points['seed_xyz'].shape[1], vote_xyz.shape[1] // end_points['seed_xyz'].shape[1]
    has_jitter = iou_jit is not None
    rows = 2 * k if has_jitter else k
    a = VnLossArgs()
    a.B, a.K, a.G, a.S, a.VF, a.N = nb, k, g, s, vf, end_points['vote_label'].shape[1]
    a.NH, a.NS, a.NC, a.NI = h_scores.shape[2], s_scores.shape[2], sem.shape[2], iou.shape[2]
    a.has_jitter = 1 if has_jitter else 0
    a.grad_scale = float(grad_scale)
    keep = []  # tensors whose storage the struct points into

    def ptr(t):
        keep.append(t)
        return t.data_ptr()
    for key, dt in (('center_label', torch.float32), ('box_label_mask', torch.float32),
                    ('heading_class_label', torch.int64), ('heading_residual_label', torch.float32),
                    ('size_class_label', torch.int64), ('size_residual_label', torch.float32),
                    ('sem_cls_label', torch.int64), ('vote_label', torch.float32),
                    ('vote_label_mask', torch.int64)):
        setattr(a, key, ptr(_label(end_points, key, dt)))
    seed_inds = end_points['seed_inds']
    if seed_inds.dtype != torch.int32 or seed_inds.stride(1) != 1:
        seed_inds = seed_inds.int().contiguous()
    a.seed_inds, a.seed_inds_stride = ptr(seed_inds), seed_inds.stride(0)
    a.mean_size = ptr(config.mean_size(dev).contiguous())
    preds = {"agg_xyz": end_points['aggregated_vote_xyz'], "obj": obj, "center": center,
             "h_scores": h_scores, "h_resn": h_resn, "s_scores": s_scores, "s_resn": s_resn,
             "sem": sem, "iou": iou, "iou_jit": iou_jit if has_jitter else iou,
             "seed_xyz": end_points['seed_xyz'], "vote_xyz": vote_xyz}
    if has_jitter:
        preds.update(jit_center=end_points['jitter_center'], jit_size=end_points['jitter_size'],
                     jit_heading=end_points['jitter_heading'])
    else:
        preds.update(jit_center=center, jit_size=center, jit_heading=center)
    for name, t in preds.items():
        if t.dtype != torch.float32 or t.device != dev:
            raise RuntimeError("%s must be a float32 tensor on %s" % (name, dev))
        keep.append(t)
        setattr(a, name, _view(t))

    f32 = dict(dtype=torch.float32, device=dev)
    boxes = torch.empty((nb, rows, 7), **f32)
    gt_boxes = torch.empty((nb, g, 7), **f32)
    a.boxes, a.gt_boxes = ptr(boxes), ptr(gt_boxes)
    _launch("votenet_loss_decode", a, dev)
    iou_lab, iou_assign = _scene_iou(boxes, gt_boxes)
    a.iou_lab, a.iou_assign = ptr(iou_lab), ptr(iou_assign)

    stats = torch.empty(ST_COUNT, **f32)
    objectness_label = torch.empty((nb, k), dtype=torch.int64, device=dev)
    objectness_mask = torch.empty((nb, k), **f32)
    object_assignment = torch.empty((nb, k), dtype=torch.int64, device=dev)
    gt_nearest = torch.empty((nb, g), dtype=torch.int32, device=dev)
    a.stats, a.objectness_label = ptr(stats), ptr(objectness_label)
    a.objectness_mask, a.object_assignment = ptr(objectness_mask), ptr(object_assignment)
    a.gt_nearest = ptr(gt_nearest)
    a.partials = ptr(torch.empty(max(1, _scratch_floats(a, dev)), **f32))
    shapes = [(nb, k, 2), (nb, k, 3), (nb, k, a.NH), (nb, k, a.NH), (nb, k, a.NS),
              (nb, k, a.NS, 3), (nb, k, a.NC), (nb, k, a.NI),
              (nb, k, a.NI) if has_jitter else (0,), (nb, s * vf, 3)]
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    flat = None
    if grad_dest is None:
        flat = torch.empty(sum(sizes), **f32)  # every element is written by the kernel
        off = 0
        for name, n in zip(_GRADS, sizes):
            setattr(a, name, flat.data_ptr() + 4 * off)
            off += n
        keep.append(flat)
    else:
        for name in _GRADS:
            setattr(a, name, grad_dest[name])
    _launch("votenet_loss_forward_backward", a, dev)
    pred_bbox = boxes[:, :k]
    return (stats, objectness_label, objectness_mask, object_assignment, pred_bbox,
            (flat, shapes, sizes, has_jitter))


def supported(end_points, supervised_inds):
    return (supervised_inds is None or isinstance(supervised_inds, slice)) and \
        'iou_scores' in end_points and end_points['center'].dim() == 3


def get_labeled_loss_fused(end_points, dataset_config, supervised_inds=None):
    """Same contract as losses.get_labeled_loss for `supervised_inds` None (every scene) or
    slice(0, n) (labeled scenes first)."""
    nb = end_points['center'].shape[0] if supervised_inds is None else int(supervised_inds.stop)

    def sel(key):
        t = end_points[key]
        return t if t.shape[0] == nb else t[:nb]
    has_jitter = 'jitter_center' in end_points
    stats, objectness_label, objectness_mask, object_assignment, pred_bbox = _FusedLabeledLoss.apply(
        end_points, dataset_config, nb, sel('objectness_scores'), sel('center'), sel('heading_scores'),
        sel('heading_residuals_normalized'), sel('size_scores'), sel('size_residuals_normalized'),
        sel('sem_cls_scores'), sel('iou_scores'), sel('iou_scores_jitter') if has_jitter else None,
        sel('vote_xyz'))
    log = stats.detach()
    for key, i in _STAT_KEYS.items():
        end_points[key] = log[i]
    if has_jitter:
        for key, i in _JITTER_KEYS.items():
            end_points[key] = log[i]
    end_points['objectness_label'] = objectness_label
    end_points['objectness_mask'] = objectness_mask
    end_points['object_assignment'] = object_assignment
    end_points['pred_bbox'] = pred_bbox
    loss = stats[ST_LOSS]
    end_points['detection_loss'] = loss
    end_points['loss'] = loss
    return loss, end_points


_CONSISTENCY_GRADS = ("g_center", "g_h_scores", "g_h_resn", "g_s_scores", "g_s_resn", "g_sem")
_CONSISTENCY_KEYS = {
    'unlabeled_objectness_loss': ST_OBJ, 'unlabeled_pos_ratio': ST_POS_RATIO,
    'unlabeled_neg_ratio': ST_NEG_RATIO, 'unlabeled_center_loss': ST_CENTER,
    'unlabeled_heading_cls_loss': ST_HCLS, 'unlabeled_heading_reg_loss': ST_HREG,
    'unlabeled_size_cls_loss': ST_SCLS, 'unlabeled_size_reg_loss': ST_SREG,
    'unlabeled_sem_cls_loss': ST_SEM, 'unlabeled_box_loss': ST_BOX,
}


class _FusedConsistencyLoss(torch.autograd.Function):
    """The consistency loss on pseudo labels (losses_unlabeled.get_pseudo_detection_loss) with the
    kernels of the supervised loss (VnLossArgs.consistency = 1): ONE call, two launches, where the
    tensor version is ~90 small kernels forward and as many backward."""

    @staticmethod
    def forward(ctx, labels, config, agg_xyz, obj, center, h_scores, h_resn, s_scores, s_resn, sem):
        stats, objectness_label, objectness_mask, object_assignment, extra = _consistency_pass(
            labels, config, agg_xyz, obj, center, h_scores, h_resn, s_scores, s_resn, sem)
        ctx.flat, ctx.shapes, ctx.sizes = extra
        ctx.mark_non_differentiable(objectness_label, objectness_mask, object_assignment)
        return stats, objectness_label, objectness_mask, object_assignment

    @staticmethod
    def backward(ctx, g_stats, *unused):
        first = ctx.sizes[0]  # (the objectness rows are zeros: no gradient)
        scaled = ctx.flat[first:] * g_stats[ST_LOSS]
        out, off = [], 0
        for sh, n in zip(ctx.shapes[1:], ctx.sizes[1:]):
            out.append(scaled[off:off + n].view(sh))
            off += n
        return (None, None, None, None) + tuple(out)


def _consistency_pass(labels, config, agg_xyz, obj, center, h_scores, h_resn, s_scores, s_resn, sem,
                      grad_dest=None, grad_scale=1.0):
    """The consistency mode's launches; grad_dest / grad_scale as in _labeled_pass (names: g_obj and
    _CONSISTENCY_GRADS)."""
    dev = center.device
    nb, k = center.shape[:2]
    g = labels['center'].shape[1]
    a = VnLossArgs()
    a.B, a.K, a.G, a.S, a.VF, a.N = nb, k, g, 0, 0, 0
    a.NH, a.NS, a.NC, a.NI = h_scores.shape[2], s_scores.shape[2], sem.shape[2], 1
    a.has_jitter, a.consistency = 0, 1
    a.grad_scale = float(grad_scale)
    keep = []

    def ptr(t):
        keep.append(t)
        return t.data_ptr()
    for field, key, dt in (('center_label', 'center', torch.float32), ('box_label_mask', 'mask', torch.float32),
                           ('heading_class_label', 'heading_class', torch.int64),
                           ('heading_residual_label', 'heading_residual', torch.float32),
                           ('size_class_label', 'size_class', torch.int64),
                           ('size_residual_label', 'size_residual', torch.float32),
                           ('sem_cls_label', 'sem_cls', torch.int64)):
        t = labels[key]
        if t.dtype != dt or t.shape[0] != nb:
            raise RuntimeError("pseudo label %s must be %s with %d scenes" % (key, dt, nb))
        setattr(a, field, ptr(t.contiguous()))
    a.mean_size = ptr(config.mean_size(dev).contiguous())
    preds = {"agg_xyz": agg_xyz, "obj": obj, "center": center, "h_scores": h_scores, "h_resn": h_resn,
             "s_scores": s_scores, "s_resn": s_resn, "sem": sem}
    for name, t in preds.items():
        if t.dtype != torch.float32 or t.device != dev:
            raise RuntimeError("%s must be a float32 tensor on %s" % (name, dev))
        keep.append(t)
        setattr(a, name, _view(t))
    for name in ("iou", "iou_jit", "seed_xyz", "vote_xyz", "jit_center", "jit_size", "jit_heading"):
        setattr(a, name, _view(center))  # unused in this mode
    f32 = dict(dtype=torch.float32, device=dev)
    stats = torch.empty(ST_COUNT, **f32)
    objectness_label = torch.empty((nb, k), dtype=torch.int64, device=dev)
    objectness_mask = torch.empty((nb, k), **f32)
    object_assignment = torch.empty((nb, k), dtype=torch.int64, device=dev)
    gt_nearest = torch.empty((nb, g), dtype=torch.int32, device=dev)
    a.stats, a.objectness_label = ptr(stats), ptr(objectness_label)
    a.objectness_mask, a.object_assignment = ptr(objectness_mask), ptr(object_assignment)
    a.gt_nearest = ptr(gt_nearest)
    a.partials = ptr(torch.empty(max(1, _scratch_floats(a, dev)), **f32))
    shapes = [(nb, k, 2), (nb, k, 3), (nb, k, a.NH), (nb, k, a.NH), (nb, k, a.NS), (nb, k, a.NS, 3),
              (nb, k, a.NC)]
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    flat = None
    if grad_dest is None:
        flat = torch.empty(sum(sizes), **f32)  # every element is written by the kernel
        off = 0
        for name, n in zip(("g_obj",) + _CONSISTENCY_GRADS, sizes):
            setattr(a, name, flat.data_ptr() + 4 * off)
            off += n
        keep.append(flat)
    else:
        for name in ("g_obj",) + _CONSISTENCY_GRADS:
            setattr(a, name, grad_dest[name])
    _launch("votenet_loss_forward_backward", a, dev)
    return stats, objectness_label, objectness_mask, object_assignment, (flat, shapes, sizes)


def get_pseudo_detection_loss_fused(end_points, labeled_num, config):
    """Same contract as losses_unlabeled.get_pseudo_detection_loss (same end_points keys)."""
    tail = slice(labeled_num, None)
    mask = end_points['unlabeled_box_label_mask']
    center = end_points['unlabeled_center_label'][:, :, 0:3]
    # (the reference masks the centres of empty slots in place, loss_helper_unlabeled.py:150-152)
    center = torch.where((1 - mask).unsqueeze(-1).bool(), torch.full_like(center, -1000), center)
    end_points['unlabeled_center_label'] = center
    labels = {'center': center, 'mask': mask.float(),
              'heading_class': end_points['unlabeled_heading_class_label'],
              'heading_residual': end_points['unlabeled_heading_residual_label'],
              'size_class': end_points['unlabeled_size_class_label'],
              'size_residual': end_points['unlabeled_size_residual_label'],
              'sem_cls': end_points['unlabeled_sem_cls_label']}
    stats, obj_label, obj_mask, assignment = _FusedConsistencyLoss.apply(
        labels, config, end_points['aggregated_vote_xyz'][tail], end_points['objectness_scores'][tail],
        end_points['center'][tail], end_points['heading_scores'][tail],
        end_points['heading_residuals_normalized'][tail], end_points['size_scores'][tail],
        end_points['size_residuals_normalized'][tail], end_points['sem_cls_scores'][tail])
    log = stats.detach()
    for key, i in _CONSISTENCY_KEYS.items():
        end_points[key] = log[i]
    end_points['unlabeled_objectness_label'] = obj_label
    end_points['unlabeled_objectness_mask'] = obj_mask
    end_points['unlabeled_object_assignment'] = assignment
    loss = stats[ST_LOSS]
    end_points['unlabeled_detection_loss'] = loss
    return loss, end_points


class _FusedSemiLoss(torch.autograd.Function):
    """detection_loss + weight * unlabeled_detection_loss of the semi-supervised step
    (train.py:327-333) as ONE autograd node: the supervised loss' launches on the labeled scenes and
    the consistency mode's on the unlabeled ones write the rows of ONE gradient buffer per head
    output (the consistency rows already scaled by `weight`: VnLossArgs.grad_scale), so the backward
    is one multiplication -- where two nodes on two slices of every head output cost a zero-fill
    and a copy per slice (slice_backward) and an addition per output."""

    @staticmethod
    def forward(ctx, end_points, config, ln, weight, labels, obj, center, h_scores, h_resn, s_scores,
                s_resn, sem, iou, iou_jit, vote_xyz):
        dev = center.device
        bt, k = center.shape[:2]
        has_jitter = iou_jit is not None
        nh, ns, nc, ni = h_scores.shape[2], s_scores.shape[2], sem.shape[2], iou.shape[2]
        rows = [k * 2, k * 3, k * nh, k * nh, k * ns, k * ns * 3, k * nc, k * ni,
                k * ni if has_jitter else 0, vote_xyz.shape[1] * 3]      # floats per scene, _GRADS order
        shapes = [(bt, k, 2), (bt, k, 3), (bt, k, nh), (bt, k, nh), (bt, k, ns), (bt, k, ns, 3),
                  (bt, k, nc), (bt, k, ni), (bt, k, ni) if has_jitter else (0,), (bt, vote_xyz.shape[1], 3)]
        sizes = [bt * r for r in rows]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)  # (IoU / vote rows of the unlabeled scenes stay zero)
        base, off = {}, 0
        for name, n in zip(_GRADS, sizes):
            base[name] = flat.data_ptr() + 4 * off
            off += n
        head = lambda t: t[:ln]  # noqa: E731
        tail = lambda t: t[ln:]  # noqa: E731
        out_l = _labeled_pass(end_points, config, ln, head(obj), head(center), head(h_scores), head(h_resn),
                              head(s_scores), head(s_resn), head(sem), head(iou),
                              head(iou_jit) if has_jitter else None, head(vote_xyz), grad_dest=base)
        stats_l, lab_l, mask_l, assign_l, pred_bbox, _ = out_l
        row_of = dict(zip(_GRADS, rows))
        dest_u = {name: base[name] + 4 * ln * row_of[name] for name in ("g_obj",) + _CONSISTENCY_GRADS}
        stats_u, lab_u, mask_u, assign_u, _ = _consistency_pass(
            labels, config, tail(end_points['aggregated_vote_xyz']), tail(obj), tail(center), tail(h_scores),
            tail(h_resn), tail(s_scores), tail(s_resn), tail(sem), grad_dest=dest_u, grad_scale=weight)
        total = stats_l[ST_LOSS] + stats_u[ST_LOSS] * weight
        ctx.flat, ctx.shapes, ctx.sizes, ctx.has_jitter = flat, shapes, sizes, has_jitter
        ctx.mark_non_differentiable(stats_l, stats_u, lab_l, mask_l, assign_l, pred_bbox, lab_u, mask_u, assign_u)
        return total, stats_l, stats_u, lab_l, mask_l, assign_l, pred_bbox, lab_u, mask_u, assign_u

    @staticmethod
    def backward(ctx, g_total, *unused):
        scaled = ctx.flat * g_total
        out, off = [], 0
        for sh, n in zip(ctx.shapes, ctx.sizes):
            out.append(scaled[off:off + n].view(sh) if n else None)
            off += n
        g_obj, g_center, g_hs, g_hr, g_ss, g_sr, g_sem, g_iou, g_jit, g_vote = out
        return (None, None, None, None, None, g_obj, g_center, g_hs, g_hr, g_ss, g_sr, g_sem, g_iou,
                g_jit if ctx.has_jitter else None, g_vote)


def semi_loss_supported(end_points, labeled_num):
    dev = end_points['center'].device
    return (enabled() and os.environ.get("VOTENET_FUSED_SEMI_LOSS", "1") != "0" and available(dev)
            and os.environ.get("VOTENET_FUSED_CONSISTENCY", "1") != "0"
            and 'iou_scores' in end_points and end_points['center'].dim() == 3
            and 0 < labeled_num < end_points['center'].shape[0])


def get_semi_loss_fused(end_points, config, labeled_num, weight):
    """(loss, end_points) with loss = detection_loss + weight * unlabeled_detection_loss and every key
    get_labeled_loss_fused and get_pseudo_detection_loss_fused fill; the pseudo labels
    (`unlabeled_*_label`, `unlabeled_box_label_mask`) are already in end_points."""
    mask = end_points['unlabeled_box_label_mask']
    center = end_points['unlabeled_center_label'][:, :, 0:3]
    center = torch.where((1 - mask).unsqueeze(-1).bool(), torch.full_like(center, -1000), center)
    end_points['unlabeled_center_label'] = center
    labels = {'center': center, 'mask': mask.float(),
              'heading_class': end_points['unlabeled_heading_class_label'],
              'heading_residual': end_points['unlabeled_heading_residual_label'],
              'size_class': end_points['unlabeled_size_class_label'],
              'size_residual': end_points['unlabeled_size_residual_label'],
              'sem_cls': end_points['unlabeled_sem_cls_label']}
    has_jitter = 'jitter_center' in end_points
    (total, stats_l, stats_u, lab_l, mask_l, assign_l, pred_bbox, lab_u, mask_u, assign_u) = _FusedSemiLoss.apply(
        end_points, config, int(labeled_num), float(weight), labels, end_points['objectness_scores'],
        end_points['center'], end_points['heading_scores'], end_points['heading_residuals_normalized'],
        end_points['size_scores'], end_points['size_residuals_normalized'], end_points['sem_cls_scores'],
        end_points['iou_scores'], end_points['iou_scores_jitter'] if has_jitter else None,
        end_points['vote_xyz'])
    log = stats_l.detach()
    for key, i in _STAT_KEYS.items():
        end_points[key] = log[i]
    if has_jitter:
        for key, i in _JITTER_KEYS.items():
            end_points[key] = log[i]
    end_points['objectness_label'], end_points['objectness_mask'] = lab_l, mask_l
    end_points['object_assignment'], end_points['pred_bbox'] = assign_l, pred_bbox
    end_points['detection_loss'] = log[ST_LOSS]
    logu = stats_u.detach()
    for key, i in _CONSISTENCY_KEYS.items():
        end_points[key] = logu[i]
    end_points['unlabeled_objectness_label'], end_points['unlabeled_objectness_mask'] = lab_u, mask_u
    end_points['unlabeled_object_assignment'] = assign_u
    end_points['unlabeled_detection_loss'] = logu[ST_LOSS]
    end_points['loss'] = total
    return total, end_points
