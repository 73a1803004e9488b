"""Regenerate tests/golden/eval_parse_ref.npz with the REFERENCE's models/ap_helper.py
parse_predictions (build container only):

    python tests/golden/make_eval_golden.py

The reference module is imported from /root/reference and run on the CPU on seeded head outputs.
Placeholders are registered only for imports that the mirrored branch never touches (plotting /
dataset utilities that need packages absent here).  Only data is stored.
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
B, K = 2, 160


def make_inputs(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    nh, ns, nc = cfg.num_heading_bin, cfg.num_size_cluster, cfg.num_class
    clumps = torch.rand(B, 10, 3, generator=g) * 5 - 2.5
    which = torch.randint(0, 10, (B, K), generator=g)
    center = torch.gather(clumps, 1, which.unsqueeze(-1).expand(-1, -1, 3)) + r(B, K, 3) * 0.25
    return {"center": center, "heading_scores": r(B, K, nh), "heading_residuals": r(B, K, nh) * 0.1,
            "size_scores": r(B, K, ns), "size_residuals": r(B, K, ns, 3) * 0.1,
            "sem_cls_scores": r(B, K, nc) * 2, "objectness_scores": r(B, K, 2) * 3,
            "iou_scores": r(B, K, nc)}


def main():
    for name in ("utils.eval_det", "sunrgbd", "sunrgbd.sunrgbd_utils", "pc_util", "utils.pc_util",
                 "pcdet", "pcdet.ops", "pcdet.ops.iou3d_nms", "pcdet.ops.iou3d_nms.iou3d_nms_utils"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["utils.eval_det"].eval_det_multiprocessing = None
    sys.modules["utils.eval_det"].get_iou_obb = None
    sys.modules["sunrgbd.sunrgbd_utils"].extract_pc_in_box3d = None
    sys.modules["pc_util"].bbox_corner_dist_measure = None
    sys.modules["utils.pc_util"].random_sampling = None
    sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_utils"].boxes_iou3d_gpu = None
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "utils"))
    ap = importlib.import_module("models.ap_helper")
    spec = importlib.util.spec_from_file_location(
        "votenet_config", os.path.join(ROOT, "3dioumatch_amd", "votenet", "config.py"))
    cfgmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfgmod)
    out = {}
    variants = [("scannet", cfgmod.scannet_config(), True, True),
                ("sunrgbd", cfgmod.sunrgbd_config(), True, False),
                ("nocls", cfgmod.scannet_config(), False, False)]
    for tag, cfg, cls_nms, use_iou in variants:
        class RefConfig(object):
            num_class = cfg.num_class
            class2angle = staticmethod(cfg.class2angle)

            @staticmethod
            def class2size(pred_cls, residual):
                return cfg.mean_size_arr.astype(np.float64)[pred_cls, :] + residual
        config_dict = {"dataset_config": RefConfig, "remove_empty_box": False, "use_3d_nms": True,
                       "nms_iou": 0.25, "use_old_type_nms": False, "cls_nms": cls_nms,
                       "use_iou_for_nms": use_iou, "per_class_proposal": cls_nms, "conf_thresh": 0.05}
        ep = make_inputs(cfg, seed={"scannet": 21, "sunrgbd": 22, "nocls": 23}[tag])
        for k, v in ep.items():
            out["%s_in::%s" % (tag, k)] = v.numpy()
        ep2 = dict(ep)
        batch = ap.parse_predictions(ep2, config_dict)
        corners_all, _ = ap.predictions2corners3d(dict(ep), config_dict)
        out[tag + "_pred_mask"] = np.asarray(ep2["pred_mask"]).astype(np.int32)
        out[tag + "_corners"] = corners_all.astype(np.float32)
        for i, cur in enumerate(batch):  # the list as (class, proposal index, confidence) triples
            key = {corners_all[i, j].tobytes(): j for j in range(K - 1, -1, -1)}
            out["%s_cls_%d" % (tag, i)] = np.array([c for c, _, _ in cur], np.int64)
            out["%s_j_%d" % (tag, i)] = np.array([key[b.tobytes()] for _, b, _ in cur], np.int64)
            out["%s_conf_%d" % (tag, i)] = np.array([s for _, _, s in cur], np.float32)
        out[tag + "_flags"] = np.array([cls_nms, use_iou], np.int32)
        print(tag, "kept per scene", out[tag + "_pred_mask"].sum(1).tolist(), "list sizes",
              [len(c) for c in batch])
    path = os.path.join(HERE, "eval_parse_ref.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
