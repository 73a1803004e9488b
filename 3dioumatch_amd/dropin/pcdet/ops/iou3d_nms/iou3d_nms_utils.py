"""pcdet.ops.iou3d_nms.iou3d_nms_utils -- host-side mirror of the reference wrappers
(OpenPCDet/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:12-116): same function names, arguments and
return values; box rows are (x, y, z, dx, dy, dz, heading).

`boxes_iou3d_gpu` is the function 3DIoUMatch reaches through utils/box_util.py:140-149 and
models/loss_helper_iou.py:43,98,106.  The reference composes it from one overlap kernel plus
about ten elementwise torch kernels; here the same arithmetic (same order, same 1e-6 clamp)
runs inside one HIP kernel.
"""
import torch

from ...utils import common_utils
from . import iou3d_nms_cuda


def _check_boxes(boxes_a, boxes_b):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """BEV IoU on the CPU; accepts numpy arrays or CPU tensors and answers in kind."""
    boxes_a, is_numpy = common_utils.check_numpy_to_torch(boxes_a)
    boxes_b, is_numpy = common_utils.check_numpy_to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    _check_boxes(boxes_a, boxes_b)
    ans_iou = boxes_a.new_zeros((boxes_a.shape[0], boxes_b.shape[0]))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7),(M,7) device tensors -> (N,M) BEV IoU."""
    _check_boxes(boxes_a, boxes_b)
    ans_iou = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32,
                          device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) device tensors -> (N,M) 3-D IoU = BEV overlap x z-overlap over the union
    volume, denominator clamped at 1e-6."""
    _check_boxes(boxes_a, boxes_b)
    iou3d = torch.empty((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32,
                        device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou3d_fused_gpu(boxes_a.contiguous(), boxes_b.contiguous(), iou3d)
    return iou3d


def boxes_iou3d_scene_max_gpu(boxes_a, boxes_b):
    """(S,P,7),(S,G,7) device tensors -> (best IoU (S,P) f32, index of the best same-scene box
    (S,P) i64): what compute_iou_labels (loss_helper_iou.py:98-111) derives from the all-pairs
    matrix, computed for the same-scene pairs only."""
    assert boxes_a.shape[2] == boxes_b.shape[2] == 7
    best = torch.empty(boxes_a.shape[:2], dtype=torch.float32, device=boxes_a.device)
    idx = torch.empty(boxes_a.shape[:2], dtype=torch.int32, device=boxes_a.device)
    iou3d_nms_cuda.scene_best_iou3d_gpu(boxes_a.contiguous(), boxes_b.contiguous(), best, idx)
    return best, idx.long()


def _nms_common(fn, boxes, scores, thresh, pre_maxsize=None):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    device_form = getattr(iou3d_nms_cuda, "nms_device", None)
    if device_form is not None and fn in (iou3d_nms_cuda.nms_gpu, iou3d_nms_cuda.nms_normal_gpu):
        # the keep list stays on the device: one 4-byte copy (the count) instead of the list's round
        # trip through the CPU tensor of the reference interface (iou3d_nms_utils.py:90-104)
        keep, num_out = device_form(boxes, thresh, fn is iou3d_nms_cuda.nms_normal_gpu)
        return order[keep[:num_out]].contiguous(), None
    keep = torch.empty(boxes.size(0), dtype=torch.int64)
    num_out = fn(boxes, keep, thresh)
    return order[keep[:num_out].to(boxes.device)].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Greedy NMS with the rotated 3-D IoU; returns (kept indices into `boxes`, None)."""
    return _nms_common(iou3d_nms_cuda.nms_gpu, boxes, scores, thresh, pre_maxsize)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Greedy NMS with the axis-aligned BEV IoU; returns (kept indices, None)."""
    return _nms_common(iou3d_nms_cuda.nms_normal_gpu, boxes, scores, thresh)
