"""Average-precision evaluation with the oriented-box IoU on the device.

Host-side mirror of the reference utils/eval_det.py (voc_ap :29-61, eval_det_cls :82-157,
eval_det_multiprocessing :213-262 as called by models/ap_helper.py:APCalculator :382-435 with
get_iou_obb).  The reference walks the detections of a class in score order and, for each, calls
the pure-Python box3d_iou (utils/box_util.py:112-137) against every ground-truth box of its scan
-- by far the slowest part of an evaluation epoch, which is why it forks a 10-process pool.  The
IoUs do not depend on the walk, so here ALL detections of ALL classes go through one kernel launch
(include/iou3d_hip.h: iou3d_corners_best_match) that returns (ovmax, jmax) per detection; the
greedy true/false-positive marking is then a stable sort and a first-occurrence lookup per class.

Differences, by construction: detections with exactly equal confidence keep their input order (the
reference's np.argsort(-confidence) leaves tie order to the sort implementation); a ground-truth
class without any prediction gets AP = rec = prec = 0 (what eval_det_multiprocessing intends,
:257-261; its list indexing misassigns results in that case).
"""
import importlib

import numpy as np
import torch


def corners_iou3d_gpu(a, b):
    """a (n,8,3), b (m,8,3) float32 GPU corners -> (n,m) float64 box3d_iou(a[i], b[j])[0]."""
    _L = importlib.import_module("3dioumatch_amd._lib")
    for t, name in ((a, "a"), (b, "b")):
        if not t.is_cuda or t.dtype != torch.float32 or t.shape[1:] != (8, 3):
            raise RuntimeError("%s must be a (n,8,3) float32 GPU tensor" % name)
    n, m = a.shape[0], b.shape[0]
    out = torch.empty((n, m), dtype=torch.float64, device=a.device)
    a, b = a.contiguous(), b.contiguous()  # (bound to locals: they must outlive the launch)
    with torch.cuda.device(a.device):
        _L.check(_L.lib.iou3d_corners_iou3d(n, a.data_ptr(), m, b.data_ptr(),
                                            out.data_ptr(), _L.current_stream_ptr(a.device)),
                 "iou3d_corners_iou3d")
    return out


def corners_best_match_gpu(det, gt_begin, gt_count, gt):
    """det (nd,8,3) f32, gt (ng,8,3) f32, gt_begin / gt_count (nd,) i32, all on the GPU ->
    (ovmax (nd,) f64, jmax (nd,) i32): eval_det_cls' per-detection loop (eval_det.py:128-141)."""
    _L = importlib.import_module("3dioumatch_amd._lib")
    for t, dt, name in ((det, torch.float32, "det"), (gt, torch.float32, "gt"),
                        (gt_begin, torch.int32, "gt_begin"), (gt_count, torch.int32, "gt_count")):
        if not t.is_cuda or t.dtype != dt:
            raise RuntimeError("%s must be a %s GPU tensor" % (name, dt))
    nd = det.shape[0]
    ovmax = torch.empty(nd, dtype=torch.float64, device=det.device)
    jmax = torch.empty(nd, dtype=torch.int32, device=det.device)
    det, gt_begin, gt_count, gt = (t.contiguous() for t in (det, gt_begin, gt_count, gt))
    with torch.cuda.device(det.device):
        _L.check(_L.lib.iou3d_corners_best_match(
            nd, det.data_ptr(), gt_begin.data_ptr(), gt_count.data_ptr(), gt.data_ptr(), ovmax.data_ptr(),
            jmax.data_ptr(), _L.current_stream_ptr(det.device)), "iou3d_corners_best_match")
    return ovmax, jmax


def _best_match(det, gt_begin, gt_count, gt, device):
    """numpy in / numpy out around the kernel (tests substitute the oracle)."""
    dev = torch.device(device if device is not None else "cuda:0")
    if dev.type != "cuda":
        raise RuntimeError("eval_det needs a GPU device (there is no CPU path)")
    if len(gt) == 0:
        gt = np.zeros((1, 8, 3), np.float32)
    ov, jm = corners_best_match_gpu(torch.from_numpy(det).to(dev), torch.from_numpy(gt_begin).to(dev),
                                    torch.from_numpy(gt_count).to(dev), torch.from_numpy(gt).to(dev))
    return ov.cpu().numpy(), jm.cpu().numpy()


def voc_ap(rec, prec, use_07_metric=False):
    """eval_det.py:29-61."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]       # precision envelope, :50-52
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def _mark(score, ovmax, gt_id, npos, ovthresh, use_07_metric):
    """eval_det.py:119-157 for one class once (ovmax, matched ground-truth id) are known."""
    order = np.argsort(-score, kind="stable")
    ov, gid = ovmax[order], gt_id[order]
    hit = ov > ovthresh
    tp = np.zeros(len(order))
    if hit.any():                                        # first detection to claim a box is the TP
        pos = np.nonzero(hit)[0]
        _, first = np.unique(gid[pos], return_index=True)
        tp[pos[first]] = 1.
    fp = 1. - tp
    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def _flatten(pred, gt):
    """{class: {img: [(box, score)]}}, {class: {img: [box]}} -> flat arrays + per-class slices."""
    det, score, begin, count, gts = [], [], [], [], []
    det_slice, npos = {}, {}
    n_gt = 0
    for c in gt.keys():
        where = {}
        npos[c] = 0
        for img, boxes in gt[c].items():
            where[img] = (n_gt, len(boxes))
            gts.extend(boxes)
            n_gt += len(boxes)
            npos[c] += len(boxes)
        d0 = len(det)
        for img, lst in pred.get(c, {}).items():
            b, k = where.get(img, (0, 0))
            for box, s in lst:
                det.append(box)
                score.append(s)
                begin.append(b)
                count.append(k)
        det_slice[c] = slice(d0, len(det))
    arr = lambda x, dt, shape: (np.asarray(x, dt).reshape(shape) if len(x) else np.zeros(shape, dt))  # noqa: E731
    return (arr(det, np.float32, (-1, 8, 3)), arr(score, np.float64, (-1,)),
            arr(begin, np.int32, (-1,)), arr(count, np.int32, (-1,)),
            arr(gts, np.float32, (-1, 8, 3)), det_slice, npos)


def _group(pred_all, gt_all):
    """eval_det.py:227-243: {img: [(class, box, score)]} -> {class: {img: [...]}}."""
    pred, gt = {}, {}
    for img_id in pred_all.keys():
        for classname, bbox, score in pred_all[img_id]:
            pred.setdefault(classname, {}).setdefault(img_id, []).append((bbox, score))
            gt.setdefault(classname, {}).setdefault(img_id, [])
    for img_id in gt_all.keys():
        for classname, bbox in gt_all[img_id]:
            gt.setdefault(classname, {}).setdefault(img_id, []).append(bbox)
    return pred, gt


def eval_det_cls(pred, gt, ovthresh=0.25, use_07_metric=False, device=None):
    """One class: pred {img: [(corners, score)]}, gt {img: [corners]} -> rec, prec, ap
    (eval_det.py:82-157 with get_iou_func = get_iou_obb)."""
    det, score, begin, count, gts, _, npos = _flatten({0: pred}, {0: gt})
    ovmax, jmax = _best_match(det, begin, count, gts, device)
    return _mark(score, ovmax, begin.astype(np.int64) + jmax, npos[0], ovthresh, use_07_metric)


def eval_det(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, device=None):
    """pred_all {img: [(class, corners (8,3), score)]}, gt_all {img: [(class, corners)]} ->
    rec, prec, ap dicts keyed by class (eval_det.py:213-262)."""
    pred, gt = _group(pred_all, gt_all)
    det, score, begin, count, gts, det_slice, npos = _flatten(pred, gt)
    ovmax, jmax = _best_match(det, begin, count, gts, device)
    gt_id = begin.astype(np.int64) + jmax
    rec, prec, ap = {}, {}, {}
    for c in gt.keys():
        if c not in pred:
            rec[c], prec[c], ap[c] = 0, 0, 0
            continue
        s = det_slice[c]
        rec[c], prec[c], ap[c] = _mark(score[s], ovmax[s], gt_id[s], npos[c], ovthresh, use_07_metric)
    return rec, prec, ap
