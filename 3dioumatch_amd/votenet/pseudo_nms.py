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
