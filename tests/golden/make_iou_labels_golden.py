"""Regenerate tests/golden/iou_labels_ref.npz -- BUILD container only (imports the reference's
Python from /root/reference; only seeded inputs and numeric outputs are stored).

The REFERENCE's compute_iou_labels (models/loss_helper_iou.py:52-112) on seeded end_points:
decoded predictions (heading / size class arg-max, residual gather, negated heading), GT boxes
with empty slots pushed to -1000, all-pairs 3-D IoU, block-diagonal max / arg-max.  Its
box3d_iou_batch_gpu is CUDA-only, so the IoU itself comes from the oracle, which is pinned
bit-for-bit to the reference's compiled iou3d_cpu.cpp (tests/golden/iou_bev_cpu_ref.npz).
Stored per dataset config (ScanNet: axis-aligned, SUN RGB-D: 12 heading bins):
  inputs  : the prediction tensors and label tensors fed in
  outputs : iou_labels (B,K) f32, objectness_label (B,K) i64, object_assignment (B,K) i64,
            pred_bbox (B,K,7), and the `reverse=True` matrix (B,G,K).
tests/test_train_step.py::test_iou_labels_* replays them through votenet/losses.py.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
sys.path.insert(0, ROOT)
sys.path.insert(0, TESTS)
sys.path.insert(0, HERE)
from oracle.oracle import Oracle  # noqa: E402
from oracle import standin as oracle_ext  # noqa: E402

REF = "/root/reference"
B, K, G = 3, 96, 64


def seeded_inputs(cfg, seed):
    """Predictions around the ground-truth boxes of a synthetic batch, so that the IoU labels
    cover the whole range (0 ... ~0.9) instead of being almost all zero."""
    datamod = importlib.import_module("3dioumatch_amd.votenet.data")
    batch = datamod.make_batch(B, 2048, cfg, seed=seed, num_objects=9)
    g = torch.Generator().manual_seed(seed + 1)
    nh, ns = cfg.num_heading_bin, cfg.num_size_cluster
    pick = torch.randint(0, 9, (B, K), generator=g)
    gt_center = torch.gather(batch["center_label"], 1, pick.unsqueeze(-1).expand(-1, -1, 3))
    center = gt_center + torch.randn(B, K, 3, generator=g) * 0.15
    center[:, ::7] += 3.0  # some far-away proposals: IoU 0, objectness 0
    votes = center + torch.randn(B, K, 3, generator=g) * 0.05
    s_cls = torch.gather(batch["size_class_label"], 1, pick)
    h_cls = torch.gather(batch["heading_class_label"], 1, pick)
    size_scores = torch.randn(B, K, ns, generator=g)
    size_scores.scatter_(2, s_cls.unsqueeze(-1), 5.0)
    heading_scores = torch.randn(B, K, nh, generator=g)
    heading_scores.scatter_(2, h_cls.unsqueeze(-1), 5.0)
    size_res = torch.randn(B, K, ns, 3, generator=g) * 0.1
    gt_sres = torch.gather(batch["size_residual_label"], 1, pick.unsqueeze(-1).expand(-1, -1, 3))
    size_res.scatter_add_(2, s_cls.view(B, K, 1, 1).expand(-1, -1, 1, 3), gt_sres.unsqueeze(2))
    size_res[:, 5::11] -= 3.0  # negative decoded sizes -> the 1e-6 clamp
    heading_res = torch.randn(B, K, nh, generator=g) * 0.05
    gt_hres = torch.gather(batch["heading_residual_label"], 1, pick)
    heading_res.scatter_add_(2, h_cls.unsqueeze(-1), gt_hres.unsqueeze(-1))
    sem = torch.randn(B, K, cfg.num_class, generator=g)
    obj = torch.randn(B, K, 2, generator=g)
    labels = {k: batch[k] for k in ("center_label", "box_label_mask", "heading_class_label",
                                    "heading_residual_label", "size_class_label",
                                    "size_residual_label")}
    preds = dict(pred_votes=votes, pred_center=center, pred_sem_cls=sem, pred_objectness=obj,
                 pred_heading_scores=heading_scores, pred_heading_residuals=heading_res,
                 pred_size_scores=size_scores, pred_size_residuals=size_res)
    return labels, preds


def main():
    o = Oracle(omp=True)
    sys.modules["pointnet2._ext"] = oracle_ext.make(o)
    iou_stub = types.ModuleType("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    iou_stub.boxes_iou3d_gpu = lambda a, b: torch.from_numpy(
        o.boxes_iou3d(a.detach().numpy(), b.detach().numpy()))
    iou_stub.boxes_iou3d_scene_max_gpu = None  # GPU-only entry point of the mirror, unused here
    for name in ("pcdet", "pcdet.ops", "pcdet.ops.iou3d_nms"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_utils"] = iou_stub
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    importlib.import_module("3dioumatch_amd")
    cfgmod = importlib.import_module("3dioumatch_amd.votenet.config")
    importlib.import_module("3dioumatch_amd.votenet.data")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "pointnet2"))
    from models.loss_helper_iou import compute_iou_labels  # noqa: E402  (the reference's)

    out = {}
    for tag, cfg in (("scannet", cfgmod.scannet_config()), ("sunrgbd", cfgmod.sunrgbd_config())):
        labels, preds = seeded_inputs(cfg, seed=41)
        inds = torch.arange(B)
        ep = {k: v.clone() for k, v in labels.items()}
        args = [preds[k].clone() for k in ("pred_votes", "pred_center", "pred_sem_cls",
                                           "pred_objectness", "pred_heading_scores",
                                           "pred_heading_residuals", "pred_size_scores",
                                           "pred_size_residuals")]
        iou, objl, assign = compute_iou_labels(ep, inds, *args, {"dataset_config": cfg})
        ep2 = {k: v.clone() for k, v in labels.items()}
        rev = compute_iou_labels(ep2, inds, *[a.clone() for a in args], {"dataset_config": cfg},
                                 reverse=True)
        for k, v in labels.items():
            out["%s_in::%s" % (tag, k)] = v.numpy()
        for k, v in preds.items():
            out["%s_in::%s" % (tag, k)] = v.numpy()
        out[tag + "_iou_labels"] = iou.numpy()
        out[tag + "_objectness_label"] = objl.numpy()
        out[tag + "_object_assignment"] = assign.numpy()
        out[tag + "_pred_bbox"] = ep["pred_bbox"].detach().numpy()
        out[tag + "_reverse"] = rev.numpy()
        print(tag, "mean IoU %.4f, > 0.25: %d of %d, objectness %d" %
              (float(iou.mean()), int((iou > 0.25).sum()), iou.numel(), int(objl.sum())))
    path = os.path.join(HERE, "iou_labels_ref.npz")
    np.savez_compressed(path, **out)
    print("iou_labels_ref.npz %.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
