"""Regenerate tests/golden/lhs_nms_ref.npz from the REFERENCE's own code (build container only):

    python tests/golden/make_lhs_golden.py

utils/nms.py:lhs_3d_faster_samecls and utils/box_util.py:get_3d_box are pure numpy, so they are
imported from /root/reference and run on seeded boxes built exactly as
models/loss_helper_unlabeled.py:447-487 builds them.  Only data is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
import types  # noqa: E402

# utils/nms.py imports pc_util (which needs plyfile/trimesh, absent here) only for a function
# lhs_3d_faster_samecls never calls; utils/box_util.py imports the GPU IoU wrapper it does not
# use in get_3d_box.  Neither is on the path being pinned: empty placeholders let the two files load.
_pc = types.ModuleType("pc_util")
_pc.bbox_corner_dist_measure = None
sys.modules["pc_util"] = _pc
for _name in ("pcdet", "pcdet.ops", "pcdet.ops.iou3d_nms"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
_iou = types.ModuleType("pcdet.ops.iou3d_nms.iou3d_nms_utils")
_iou.boxes_iou3d_gpu = None
sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_utils"] = _iou
sys.path.insert(0, os.path.join(REF, "utils"))
import nms as ref_nms  # noqa: E402
import box_util as ref_box  # noqa: E402


def flip_axis_to_camera(pc):  # models/ap_helper.py:28-35 (that module imports heavy deps)
    pc2 = np.copy(pc)
    pc2[..., [0, 1, 2]] = pc2[..., [0, 2, 1]]
    pc2[..., 1] *= -1
    return pc2


def scene(g, n, clumps, oriented):
    centers = g.uniform(-3, 3, (clumps, 3)).astype(np.float32)
    which = g.integers(0, clumps, n)
    center = (centers[which] + g.normal(0, 0.25, (n, 3))).astype(np.float32)
    size = g.uniform(0.3, 1.6, (n, 3)).astype(np.float32)
    heading = (g.uniform(-np.pi, np.pi, n) if oriented else np.zeros(n)).astype(np.float32)
    score = (g.uniform(0, 1, n).astype(np.float32) * g.uniform(0, 1, n).astype(np.float32))
    cls = g.integers(0, 3, n).astype(np.int64)
    return center, size, heading, score, cls


def reference_pick(center, size, heading, score, cls, thresh, old_type):
    n = center.shape[0]
    corners = np.zeros((n, 8, 3), dtype=np.float32)
    cam = flip_axis_to_camera(center)
    for j in range(n):
        # float64 size / angle, float32 centre: the dtypes the reference's numpy decoding yields
        corners[j] = ref_box.get_3d_box(size[j].astype(np.float64), float(heading[j]), cam[j, :])
    boxes = np.zeros((n, 8))
    boxes[:, 0:3] = corners.min(axis=1)
    boxes[:, 3:6] = corners.max(axis=1)
    boxes[:, 6] = score
    boxes[:, 7] = cls
    pick = ref_nms.lhs_3d_faster_samecls(boxes, thresh, old_type)
    out = np.zeros(n, np.int32)
    out[np.asarray(pick, dtype=np.int64)] = 1
    return boxes[:, 0:6].astype(np.float32), out


def main():
    g = np.random.default_rng(2024)
    arrays = {}
    cases = [(64, 6, False, 0.25, False), (64, 3, True, 0.25, False), (64, 10, False, 0.1, False),
             (64, 2, False, 0.25, True), (17, 2, True, 0.25, False), (1, 1, False, 0.25, False),
             (64, 64, False, 0.25, False)]
    for k, (n, clumps, oriented, thresh, old) in enumerate(cases):
        c, s, h, sc, cl = scene(g, n, clumps, oriented)
        aabb, pick = reference_pick(c, s, h, sc, cl, thresh, old)
        for name, arr in (("center", c), ("size", s), ("heading", h), ("score", sc), ("cls", cl),
                          ("aabb", aabb), ("pick", pick),
                          ("thresh", np.float64(thresh)), ("old", np.int32(old))):
            arrays["c%d_%s" % (k, name)] = arr
    # evaluation-path NMS (nms_3d_faster / nms_3d_faster_samecls) on up to 256 proposals
    eval_cases = [(256, 8, False, 0.25, False, True), (256, 5, True, 0.25, False, True),
                  (200, 4, False, 0.25, False, False), (128, 3, False, 0.5, True, True),
                  (65, 2, True, 0.25, False, False)]
    for k, (n, clumps, oriented, thresh, old, same) in enumerate(eval_cases):
        c, s, h, sc, cl = scene(g, n, clumps, oriented)
        corners = np.zeros((n, 8, 3), dtype=np.float32)
        cam = flip_axis_to_camera(c)
        for j in range(n):
            corners[j] = ref_box.get_3d_box(s[j].astype(np.float64), float(h[j]), cam[j, :])
        boxes = np.zeros((n, 8))
        boxes[:, 0:3] = corners.min(axis=1)
        boxes[:, 3:6] = corners.max(axis=1)
        boxes[:, 6] = sc
        boxes[:, 7] = cl
        fn = ref_nms.nms_3d_faster_samecls if same else ref_nms.nms_3d_faster
        pick = np.zeros(n, np.int32)
        pick[np.asarray(fn(boxes if same else boxes[:, :7], thresh, old), dtype=np.int64)] = 1
        for name, arr in (("center", c), ("size", s), ("heading", h), ("score", sc), ("cls", cl),
                          ("pick", pick), ("thresh", np.float64(thresh)), ("old", np.int32(old)),
                          ("same", np.int32(same))):
            arrays["e%d_%s" % (k, name)] = arr
    arrays["num_eval_cases"] = np.int32(len(eval_cases))
    arrays["num_cases"] = np.int32(len(cases))
    path = os.path.join(HERE, "lhs_nms_ref.npz")
    np.savez_compressed(path, **arrays)
    print(path, os.path.getsize(path), "bytes;",
          [int(arrays["c%d_pick" % k].sum()) for k in range(len(cases))], "picked")


if __name__ == "__main__":
    main()
