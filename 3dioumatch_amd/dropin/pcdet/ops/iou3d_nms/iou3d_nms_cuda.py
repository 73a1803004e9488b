"""pcdet.ops.iou3d_nms.iou3d_nms_cuda -- drop-in for the reference's compiled extension
(pybind surface: OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17), backed by the
gfx950 HIP library through its C ABI (include/iou3d_hip.h).

Same five functions and calling convention: the CALLER allocates the outputs and the callee
writes in place (iou3d_nms.cpp:49-88); nms_* take a CPU `keep` tensor and return the kept
count (iou3d_nms.cpp:90-138).  Differences, all lenient:
  * bad inputs raise Python exceptions instead of exit(-1) (iou3d_nms.cpp:14-26);
  * launches go to the current torch stream, not the legacy default stream;
  * `keep` may be int64 (what the Python wrapper allocates, iou3d_nms_utils.py:97) or int32
    (what the fork's C++ reads, iou3d_nms.cpp:98); it is written in its own dtype;
  * the greedy NMS scan runs on the device; only `keep`/count come back to the host.
"""
import importlib.util
import os
import sys

import torch


def _load_lib():
    name = "_3dioumatch_amd_lib"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                         "..", "..", "..", "..", "_lib.py"))
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        del sys.modules[name]
        raise
    return mod


_L = _load_lib()
_lib = _L.lib


def _chk_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must be CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous tensor" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _pairs(fn, what, boxes_a, boxes_b, ans):
    _chk_gpu(boxes_a, "boxes_a"); _chk_gpu(boxes_b, "boxes_b"); _chk_gpu(ans, "ans")
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    if ans.numel() != na * nb:
        raise RuntimeError("ans must have %d x %d elements" % (na, nb))
    with torch.cuda.device(boxes_a.device):
        _L.check(fn(na, boxes_a.data_ptr(), nb, boxes_b.data_ptr(), ans.data_ptr(),
                    _L.current_stream_ptr(boxes_a.device)), what)
    return 1


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """(N,7),(M,7) -> ans_overlap (N,M) BEV intersection area, in place.  iou3d_nms.cpp:49-68"""
    return _pairs(_lib.iou3d_boxes_overlap_bev, "boxes_overlap_bev_gpu", boxes_a, boxes_b,
                  ans_overlap)


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """(N,7),(M,7) -> ans_iou (N,M) BEV IoU, in place.  iou3d_nms.cpp:70-88"""
    return _pairs(_lib.iou3d_boxes_iou_bev, "boxes_iou_bev_gpu", boxes_a, boxes_b, ans_iou)


def boxes_iou3d_fused_gpu(boxes_a, boxes_b, ans_iou3d):
    """Addition: the whole of boxes_iou3d_gpu (iou3d_nms_utils.py:48-81) in one kernel."""
    return _pairs(_lib.iou3d_boxes_iou3d, "boxes_iou3d_fused_gpu", boxes_a, boxes_b, ans_iou3d)


def scene_best_iou3d_gpu(boxes_a, boxes_b, best_iou, best_idx):
    """Addition: boxes_a (S,P,7), boxes_b (S,G,7) -> best_iou (S,P) f32, best_idx (S,P) i32 in
    place: per prediction the largest 3-D IoU with a box of the SAME scene and the first index
    attaining it (boxes_iou3d_gpu over all pairs + the block-diagonal max/gather of
    loss_helper_iou.py:98-111 in one kernel)."""
    _chk_gpu(boxes_a, "boxes_a"); _chk_gpu(boxes_b, "boxes_b"); _chk_gpu(best_iou, "best_iou")
    if boxes_a.dim() != 3 or boxes_b.dim() != 3 or boxes_a.shape[0] != boxes_b.shape[0] \
            or boxes_a.shape[2] != 7 or boxes_b.shape[2] != 7:
        raise RuntimeError("boxes must be (S,P,7) and (S,G,7)")
    s, p, g = boxes_a.shape[0], boxes_a.shape[1], boxes_b.shape[1]
    if best_iou.numel() != s * p or best_idx.numel() != s * p or best_idx.dtype != torch.int32 \
            or not best_idx.is_cuda or not best_idx.is_contiguous():
        raise RuntimeError("best_iou / best_idx must be contiguous (S,P) float32 / int32 GPU tensors")
    with torch.cuda.device(boxes_a.device):
        _L.check(_lib.iou3d_scene_best_iou3d(s, p, boxes_a.data_ptr(), g, boxes_b.data_ptr(),
                                             best_iou.data_ptr(), best_idx.data_ptr(),
                                             _L.current_stream_ptr(boxes_a.device)),
                 "scene_best_iou3d_gpu")
    return 1


def nms_device(boxes, thresh, normal=False):
    """The same NMS with the keep list LEFT ON THE DEVICE: boxes (N,7) sorted by score desc ->
    (keep (N,) int64 GPU tensor, kept count).  One 4-byte device -> host copy (the count) instead of
    the reference interface's blocking copy of the list into a CPU tensor (iou3d_nms.cpp:90-138)
    that iou3d_nms_utils.nms_gpu sends straight back to the device."""
    _chk_gpu(boxes, "boxes")
    n = boxes.shape[0]
    dev = boxes.device
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=dev), 0
    col_blocks = (n + 63) // 64
    with torch.cuda.device(dev):
        mask = torch.empty(n * col_blocks, dtype=torch.int64, device=dev)
        keep_dev = torch.empty(n, dtype=torch.int64, device=dev)
        num_dev = torch.empty(1, dtype=torch.int32, device=dev)
        _L.check(_lib.iou3d_nms(boxes.data_ptr(), n, float(thresh), 1 if normal else 0,
                                mask.data_ptr(), keep_dev.data_ptr(), num_dev.data_ptr(),
                                _L.current_stream_ptr(dev)), "nms_device")
        num = int(num_dev.item())  # synchronises
    return keep_dev, num


def _nms(boxes, keep, thresh, normal):
    _chk_gpu(boxes, "boxes")
    if keep.is_cuda or not keep.is_contiguous():
        raise RuntimeError("keep must be a contiguous CPU tensor")
    if keep.dtype not in (torch.int64, torch.int32):
        raise RuntimeError("keep must be an int64 or int32 tensor")
    n = boxes.shape[0]
    if n == 0:
        return 0
    if keep.numel() < n:
        raise RuntimeError("keep must have at least %d elements" % n)
    dev = boxes.device
    col_blocks = (n + 63) // 64
    with torch.cuda.device(dev):
        mask = torch.empty(n * col_blocks, dtype=torch.int64, device=dev)
        keep_dev = torch.empty(n, dtype=torch.int64, device=dev)
        num_dev = torch.empty(1, dtype=torch.int32, device=dev)
        _L.check(_lib.iou3d_nms(boxes.data_ptr(), n, float(thresh), 1 if normal else 0,
                                mask.data_ptr(), keep_dev.data_ptr(), num_dev.data_ptr(),
                                _L.current_stream_ptr(dev)), "nms_gpu")
        num = int(num_dev.item())  # synchronises, like the reference's blocking memcpy
        keep[:num] = keep_dev[:num].to("cpu").to(keep.dtype)
    return num


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """boxes (N,7) sorted by score desc, keep CPU tensor -> kept count (3-D IoU NMS).
    iou3d_nms.cpp:90-138"""
    return _nms(boxes, keep, nms_overlap_thresh, False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """Same with the axis-aligned BEV IoU.  iou3d_nms.cpp:141-190"""
    return _nms(boxes, keep, nms_overlap_thresh, True)


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """CPU tensors (N,7),(M,7) -> ans_iou (N,M), in place; single-threaded host code.
    iou3d_cpu.cpp:232-252"""
    for t, name in ((boxes_a, "boxes_a"), (boxes_b, "boxes_b"), (ans_iou, "ans_iou")):
        if t.is_cuda:
            raise RuntimeError("%s must be a CPU tensor" % name)
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous tensor" % name)
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be a float tensor" % name)
    _L.check(_lib.iou3d_boxes_iou_bev_cpu(boxes_a.shape[0], boxes_a.data_ptr(),
                                          boxes_b.shape[0], boxes_b.data_ptr(),
                                          ans_iou.data_ptr()), "boxes_iou_bev_cpu")
    return 1
