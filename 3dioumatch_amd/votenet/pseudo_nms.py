"""Device-side NMS of the pseudo-label filter (binding of include/lhs_hip.h).

The reference copies the 64 best teacher predictions of every unlabeled scene to the host, builds
their 3-D boxes one by one in numpy and runs utils/nms.py:lhs_3d_faster_samecls
(models/loss_helper_unlabeled.py:447-487); here the same computation is one kernel launch with no
host round trip, so the semi-supervised step stays capturable in a HIP graph.
"""
import importlib

import torch

_L = importlib.import_module("3dioumatch_amd._lib")
_lib = _L.lib


def lhs_nms_samecls_gpu(center, size, heading, score, cls, thresh, old_type=False):
    """center (S,n,3) f32, size (S,n,3) f64, heading (S,n) f64, score (S,n) f32, cls (S,n) i64 ->
    picked (S,n) bool: True for the boxes lhs_3d_faster_samecls keeps (n <= 64)."""
    for t, dt, name in ((center, torch.float32, "center"), (size, torch.float64, "size"),
                        (heading, torch.float64, "heading"), (score, torch.float32, "score"),
                        (cls, torch.int64, "cls")):
        if not t.is_cuda or t.dtype != dt:
            raise RuntimeError("%s must be a %s GPU tensor" % (name, dt))
    s, n = score.shape
    if n > 64:
        raise RuntimeError("lhs_nms_samecls: at most 64 boxes per scene (MAX_NUM_OBJ)")
    picked = torch.zeros((s, n), dtype=torch.int32, device=score.device)
    # contiguous copies (if any) stay bound to locals until after the launch: a temporary's block
    # could be handed to the next allocation on this stream before the kernel has read it
    center, size, heading, score, cls = (t.contiguous() for t in (center, size, heading, score, cls))
    with torch.cuda.device(score.device):
        _L.check(_lib.lhs_nms_samecls(s, n, center.data_ptr(), size.data_ptr(), heading.data_ptr(),
                                      score.data_ptr(), cls.data_ptr(),
                                      float(thresh), 1 if old_type else 0, picked.data_ptr(),
                                      _L.current_stream_ptr(score.device)), "lhs_nms_samecls")
    return picked.bool()


def nms3d_aabb_gpu(center, size, heading, score, cls, thresh, old_type=False, same_class=True):
    """The evaluation path's per-scene NMS (utils/nms.py nms_3d_faster / nms_3d_faster_samecls on
    the camera-frame bounds of every proposal) -> picked (S,n) bool, n <= 1024 (one workgroup per scene: 256 lanes up to 256 boxes,
    1024 lanes beyond)."""
    for t, dt, name in ((center, torch.float32, "center"), (size, torch.float64, "size"),
                        (heading, torch.float64, "heading"), (score, torch.float32, "score"),
                        (cls, torch.int64, "cls")):
        if not t.is_cuda or t.dtype != dt:
            raise RuntimeError("%s must be a %s GPU tensor" % (name, dt))
    s, n = score.shape
    if n > 1024:
        raise RuntimeError("nms3d_aabb: at most 1024 boxes per scene")
    picked = torch.zeros((s, n), dtype=torch.int32, device=score.device)
    center, size, heading, score, cls = (t.contiguous() for t in (center, size, heading, score, cls))
    with torch.cuda.device(score.device):
        _L.check(_lib.lhs_nms3d_aabb(s, n, center.data_ptr(), size.data_ptr(), heading.data_ptr(),
                                     score.data_ptr(), cls.data_ptr(),
                                     float(thresh), 1 if old_type else 0, 1 if same_class else 0,
                                     picked.data_ptr(), _L.current_stream_ptr(score.device)),
                 "lhs_nms3d_aabb")
    return picked.bool()


import ctypes  # noqa: E402

_c_int, _c_float, _vp = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class LhsPseudoArgs(ctypes.Structure):  # field order == include/lhs_hip.h
    _fields_ = ([(n, _c_int) for n in ("S", "K", "NC", "NI", "NH", "NS")] +
                [(n, _c_float) for n in ("obj_threshold", "cls_threshold", "iou_threshold")] +
                [("use_nms", _c_int)] +
                [(n, _vp) for n in ("objectness", "sem_cls", "iou", "heading_scores", "heading_residuals",
                                    "size_scores", "size_residuals", "center", "vote_xyz", "mean_size",
                                    "flip_x", "flip_y", "rot_mat", "scale", "box_center", "box_size",
                                    "box_heading", "box_score", "passed", "negative", "false_xyz", "picked",
                                    "label_mask", "center_label", "false_center_label", "sem_cls_label",
                                    "heading_label", "size_label", "heading_residual_label",
                                    "size_residual_label", "iou_label", "pseudo_gt_ratio")])


def pseudo_labels_supported(pred_center, scale, f32=(), i64=(), sem_cls=None, iou_scores=None):
    """The two-launch form covers GPU tensors, 64 <= K <= 1024 proposals, a per-axis scale (S,1,3),
    float32 network outputs (`f32`), int64 flip flags (`i64`) on the same device and one IoU score
    per proposal or per class.  Anything else -- bool flags, an autocast teacher -- is the tensor
    implementation's (get_pseudo_labels), not an error."""
    dev = pred_center.device
    if not (pred_center.is_cuda and 64 <= pred_center.shape[1] <= 1024 and
            tuple(scale.shape) == (pred_center.shape[0], 1, 3)):
        return False
    if any(t.dtype != torch.float32 or t.device != dev for t in (pred_center, scale, *f32)):
        return False
    if any(t.dtype != torch.int64 or t.device != dev for t in i64):
        return False
    if sem_cls is not None and iou_scores is not None and iou_scores.shape[2] not in (1, sem_cls.shape[2]):
        return False
    return True


def pseudo_labels_gpu(objectness, sem_cls, iou_scores, heading_scores, heading_residuals, size_scores,
                      size_residuals, center, vote_xyz, mean_size, flip_x, flip_y, rot_mat, scale,
                      obj_threshold, cls_threshold, iou_threshold, nms=None):
    """get_pseudo_labels + trans_center / trans_size of losses_unlabeled.py as two launches around
    the NMS kernel (include/lhs_hip.h lhs_pseudo_select / lhs_pseudo_finish).  `nms`: None, or
    (iou threshold, old_type) for lhs_nms_samecls on the selected boxes.  Returns a dict with
    label_mask, center_label, false_center_label, sem_cls_label, heading_label,
    heading_residual_label, size_label, size_residual_label, iou_label, pseudo_gt_ratio."""
    dev = center.device
    s, k = center.shape[:2]
    f32 = dict(dtype=torch.float32, device=dev)
    i64 = dict(dtype=torch.int64, device=dev)
    keep = []

    def inp(t, dt):
        if t.dtype != dt or t.device != dev:
            raise RuntimeError("pseudo_labels_gpu: expected %s tensors on %s" % (dt, dev))
        t = t.detach().contiguous()
        keep.append(t)
        return t.data_ptr()
    a = LhsPseudoArgs()
    a.S, a.K = s, k
    a.NC, a.NI, a.NH, a.NS = sem_cls.shape[2], iou_scores.shape[2], heading_scores.shape[2], size_scores.shape[2]
    a.obj_threshold, a.cls_threshold, a.iou_threshold = float(obj_threshold), float(cls_threshold), float(iou_threshold)
    a.use_nms = 1 if nms is not None else 0
    for name, t in (("objectness", objectness), ("sem_cls", sem_cls), ("iou", iou_scores),
                    ("heading_scores", heading_scores), ("heading_residuals", heading_residuals),
                    ("size_scores", size_scores), ("size_residuals", size_residuals), ("center", center),
                    ("vote_xyz", vote_xyz), ("mean_size", mean_size), ("rot_mat", rot_mat), ("scale", scale)):
        setattr(a, name, inp(t, torch.float32))
    a.flip_x, a.flip_y = inp(flip_x, torch.int64), inp(flip_y, torch.int64)
    n = 64
    out = {"label_mask": torch.empty((s, n), **i64), "center_label": torch.empty((s, n, 3), **f32),
           "false_center_label": torch.empty((s, n, 3), **f32), "sem_cls_label": torch.empty((s, n), **i64),
           "heading_label": torch.empty((s, n), **i64), "size_label": torch.empty((s, n), **i64),
           "heading_residual_label": torch.empty((s, n), **f32),
           "size_residual_label": torch.empty((s, n, 3), **f32), "iou_label": torch.empty((s, n), **f32),
           "pseudo_gt_ratio": torch.empty((), **f32)}
    box_center = torch.empty((s, n, 3), **f32)
    box_size = torch.empty((s, n, 3), dtype=torch.float64, device=dev)
    box_heading = torch.empty((s, n), dtype=torch.float64, device=dev)
    box_score = torch.empty((s, n), **f32)
    flags = torch.empty((2, s, n), dtype=torch.int32, device=dev)
    false_xyz = torch.empty((s, n, 3), **f32)
    a.box_center, a.box_size, a.box_heading = box_center.data_ptr(), box_size.data_ptr(), box_heading.data_ptr()
    a.box_score, a.passed, a.negative = box_score.data_ptr(), flags[0].data_ptr(), flags[1].data_ptr()
    a.false_xyz = false_xyz.data_ptr()
    for name, t in out.items():
        setattr(a, name, t.data_ptr())
    stream = _L.current_stream_ptr(dev)
    with torch.cuda.device(dev):
        _L.check(_lib.lhs_pseudo_select(ctypes.byref(a), stream), "lhs_pseudo_select")
        picked = None
        if nms is not None:
            picked = torch.zeros((s, n), dtype=torch.int32, device=dev)
            _L.check(_lib.lhs_nms_samecls(s, n, box_center.data_ptr(), box_size.data_ptr(), box_heading.data_ptr(),
                                          box_score.data_ptr(), out["sem_cls_label"].data_ptr(), float(nms[0]),
                                          1 if nms[1] else 0, picked.data_ptr(), stream), "lhs_nms_samecls")
            a.picked = picked.data_ptr()
        _L.check(_lib.lhs_pseudo_finish(ctypes.byref(a), stream), "lhs_pseudo_finish")
    return out
