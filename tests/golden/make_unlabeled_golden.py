"""Regenerate tests/golden/unlabeled_loss_ref.npz with the REFERENCE's own
models/loss_helper_unlabeled.py:get_unlabeled_loss (build container only):

    python tests/golden/make_unlabeled_golden.py

The reference module is imported from /root/reference and run on the CPU (Tensor.cuda patched to
the identity) on seeded student / teacher outputs; inputs and every output the mirror must
reproduce are stored.  Placeholders are registered only for imports that are not on this path
(the compiled extensions, plyfile-dependent pc_util, and models/ap_helper whose single function
used here, flip_axis_to_camera, is an axis permutation restated below).  Only data is stored.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

B_LAB, B_UNL, K = 2, 3, 128
OUT_KEYS = ["unlabeled_center_label", "unlabeled_box_label_mask", "unlabeled_sem_cls_label",
            "unlabeled_heading_class_label", "unlabeled_heading_residual_label",
            "unlabeled_size_class_label", "unlabeled_size_residual_label",
            "unlabeled_false_center_label", "unlabeled_iou_label", "unlabeled_objectness_label",
            "unlabeled_objectness_mask", "unlabeled_object_assignment"]
STAT_KEYS = ["pseudo_gt_ratio", "unlabeled_objectness_loss", "unlabeled_pos_ratio",
             "unlabeled_neg_ratio", "unlabeled_center_loss", "unlabeled_heading_cls_loss",
             "unlabeled_heading_reg_loss", "unlabeled_size_cls_loss", "unlabeled_size_reg_loss",
             "unlabeled_sem_cls_loss", "unlabeled_box_loss", "unlabeled_detection_loss"]


def make_inputs(cfg, seed):
    """Teacher outputs in which > 64 proposals per unlabeled scene pass the three thresholds (so the
    top-64 selection has no ties), clumped so that the NMS suppresses, and student outputs near them."""
    g = torch.Generator().manual_seed(seed)
    b = B_LAB + B_UNL
    nh, ns, nc = cfg.num_heading_bin, cfg.num_size_cluster, cfg.num_class
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    u = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    clumps = u(b, 6, 3) * 4 - 2
    which = torch.randint(0, 6, (b, K), generator=g)
    t_center = torch.gather(clumps, 1, which.unsqueeze(-1).expand(-1, -1, 3)) + r(b, K, 3) * 0.2
    confident = u(b, K) < 0.7
    ema = {
        "center": t_center,
        "aggregated_vote_xyz": t_center + r(b, K, 3) * 0.05,
        "objectness_scores": torch.stack([-(3 + u(b, K) * 3), 3 + u(b, K) * 3], 2)
        * torch.where(confident, 1.0, 0.1).unsqueeze(-1),
        "sem_cls_scores": r(b, K, nc) + 9 * torch.nn.functional.one_hot(
            torch.randint(0, 3, (b, K), generator=g), nc) * confident.unsqueeze(-1),
        "heading_scores": r(b, K, nh),
        "heading_residuals": r(b, K, nh) * 0.1,
        "size_scores": r(b, K, ns),
        "size_residuals": r(b, K, ns, 3) * 0.1,
        "iou_scores": r(b, K, nc) + 0.5,
    }
    ep = {
        "supervised_mask": torch.tensor([1] * B_LAB + [0] * B_UNL),
        "aggregated_vote_xyz": t_center + r(b, K, 3) * 0.15,
        "center": t_center + r(b, K, 3) * 0.2,
        "objectness_scores": r(b, K, 2),
        "heading_scores": r(b, K, nh),
        "heading_residuals_normalized": r(b, K, nh) * 0.3,
        "size_scores": r(b, K, ns),
        "size_residuals_normalized": r(b, K, ns, 3) * 0.3,
        "sem_cls_scores": r(b, K, nc),
        "flip_x_axis": torch.randint(0, 2, (b,), generator=g),
        "flip_y_axis": torch.randint(0, 2, (b,), generator=g),
        "rot_angle": (u(b) - 0.5) * (np.pi / 18),
        "scale": (u(b, 1, 3) * 0.3 + 0.85),
    }
    c, s = torch.cos(ep["rot_angle"]), torch.sin(ep["rot_angle"])
    z, o = torch.zeros(b), torch.ones(b)
    ep["rot_mat"] = torch.stack([c, -s, z, s, c, z, z, z, o], 1).view(b, 3, 3)
    return ep, ema


def main():
    for name in ("pcdet", "pcdet.ops", "pcdet.ops.iou3d_nms"):
        sys.modules[name] = types.ModuleType(name)
    iou_stub = types.ModuleType("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    iou_stub.boxes_iou3d_gpu = None
    sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_utils"] = iou_stub
    pc = types.ModuleType("pc_util")
    pc.bbox_corner_dist_measure = None
    sys.modules["pc_util"] = pc
    ap = types.ModuleType("models.ap_helper")

    def flip_axis_to_camera(pc):  # models/ap_helper.py:28-35
        pc2 = np.copy(pc)
        pc2[..., [0, 1, 2]] = pc2[..., [0, 2, 1]]
        pc2[..., 1] *= -1
        return pc2
    ap.flip_axis_to_camera = flip_axis_to_camera
    sys.modules["models.ap_helper"] = ap
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "utils"))
    ref = importlib.import_module("models.loss_helper_unlabeled")

    # the config module alone (the package __init__ would import the real extensions)
    spec = importlib.util.spec_from_file_location(
        "votenet_config", os.path.join(ROOT, "3dioumatch_amd", "votenet", "config.py"))
    cfgmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfgmod)
    out = {}
    for tag, cfg in (("scannet", cfgmod.scannet_config()), ("sunrgbd", cfgmod.sunrgbd_config())):
        class RefConfig(object):  # what the reference's dataset configs provide, numpy in float64
            num_class, num_heading_bin, num_size_cluster = cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster
            mean_size_arr = cfg.mean_size_arr.astype(np.float64)
            class2angle = staticmethod(cfg.class2angle)
            class2angle_gpu = staticmethod(cfg.class2angle_gpu)
            angle2class_gpu = staticmethod(cfg.angle2class_gpu)

            @staticmethod
            def class2size(pred_cls, residual):
                return cfg.mean_size_arr.astype(np.float64)[pred_cls, :] + residual
        config_dict = {"dataset_config": RefConfig, "unlabeled_batch_size": B_UNL, "dataset": tag,
                       "nms_iou": 0.25, "use_old_type_nms": False, "obj_threshold": 0.9,
                       "cls_threshold": 0.9, "use_lhs": True, "iou_threshold": 0.25,
                       "samecls_match": False, "view_stats": False}
        ep, ema = make_inputs(cfg, seed=7 if tag == "scannet" else 8)
        for k, v in ep.items():
            out["%s_in_ep::%s" % (tag, k)] = v.numpy()
        for k, v in ema.items():
            out["%s_in_ema::%s" % (tag, k)] = v.numpy()
        loss, ep = ref.get_unlabeled_loss(dict(ep), dict(ema), RefConfig, config_dict)
        out[tag + "_loss"] = loss.detach().numpy()
        for k in OUT_KEYS:
            out["%s_out::%s" % (tag, k)] = ep[k].detach().numpy()
        for k in STAT_KEYS:
            out["%s_stat::%s" % (tag, k)] = np.float64(ep[k])
        print(tag, "loss %.5f" % float(loss), "pseudo boxes per scene",
              ep["unlabeled_box_label_mask"].sum(1).tolist(), "pseudo_gt_ratio %.3f" % float(ep["pseudo_gt_ratio"]))
    path = os.path.join(HERE, "unlabeled_loss_ref.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
