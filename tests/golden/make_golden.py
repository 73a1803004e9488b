"""Regenerate tests/golden/*.npz.  Run in the BUILD container (needs /root/reference for part 2):

    python tests/golden/make_golden.py

Part 1 (ops_*.npz): inputs + outputs of the CPU restatement (oracle/pn2_oracle.c) for every
   pointnet2 op.  The reference cannot run these ops on a CPU and has no tests for them, so these
   vectors pin the RESTATEMENT (regression vectors), not the reference binary.
Part 2 (iou_bev_cpu_ref.npz): outputs of the REFERENCE's own compiled CPU code
   (oracle/_ref/libiou3d_ref.so, built from OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp
   by `make -C oracle ref`): BEV overlap (box_overlap) and BEV IoU (boxes_iou_bev_cpu) for
   oriented, axis-aligned and known-answer boxes.  These pin oracle/iou3d_oracle.c to the
   reference bit-for-bit.
Part 3 (iou3d_nms_oracle.npz): 3-D IoU matrix and NMS results of the oracle (no reference CPU
   form exists for the wrapper epilogue / NMS kernels).
Only DATA is stored (inputs and expected outputs).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

importlib.import_module("3dioumatch_amd")
synth = importlib.import_module("3dioumatch_amd.synth")
from oracle.oracle import Oracle, Reference, have_reference  # noqa: E402

o = Oracle()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


def part1():
    # ---- FPS: uniform, edge cases (skip rule + exact duplicates), odd sizes -> block sizes
    fps = {}
    cases = {
        "U": (synth.cloud_uniform(2, 4096, synth.cube_side(4096, 0.2, 32), seed=0), 512),
        "E": (synth.cloud_edge_cases(2, 1024, 1.0, seed=1), 128),
        "n100": (synth.cloud_edge_cases(1, 100, 1.0, seed=2, near_origin=4, duplicates=8), 40),
        "n600": (synth.cloud_edge_cases(1, 600, 1.0, seed=3, near_origin=4, duplicates=32), 200),
        "n1500": (synth.cloud_edge_cases(1, 1500, 1.0, seed=4, near_origin=8, duplicates=64), 300),
        "allskip": (np.zeros((1, 64, 3), np.float32), 8),
    }
    for k, (xyz, m) in cases.items():
        fps["xyz_" + k] = xyz
        fps["m_" + k] = np.int32(m)
        fps["idx_" + k] = o.furthest_point_sampling(xyz, m)
    save("ops_fps.npz", **fps)

    # ---- ball query + group (+grad) for the reference's five (radius, nsample) settings,
    #      clouds scaled to N <= 4096 (backbone_module.py:35-72, proposal_module.py:72-79)
    bq = {}
    settings = [(0.2, 64, 4096, 256), (0.4, 32, 2048, 256), (0.8, 16, 1024, 128),
                (1.2, 16, 512, 64), (0.3, 16, 1024, 64)]
    g = np.random.default_rng(7)
    for t, (r, ns, n, m) in enumerate(settings):
        xyz = synth.cloud_edge_cases(2, n, synth.cube_side(n, r, ns), seed=10 + t,
                                     near_origin=4, duplicates=16)
        cent_idx = o.furthest_point_sampling(xyz, m)
        new_xyz = np.take_along_axis(xyz, cent_idx[..., None].astype(np.int64), axis=1)
        new_xyz[:, -1] = -50.0  # one centroid with an empty ball -> all-zero row
        idx = o.ball_query(new_xyz, xyz, r, ns)
        feats = g.standard_normal((2, 5, n)).astype(np.float32)
        grouped = o.group_points(feats, idx)
        gout = g.standard_normal(grouped.shape).astype(np.float32)
        bq.update({"r_%d" % t: np.float32(r), "ns_%d" % t: np.int32(ns), "xyz_%d" % t: xyz,
                   "new_xyz_%d" % t: new_xyz, "idx_%d" % t: idx})
        if t in (1, 3):  # keep the file small: group fixtures for two settings
            bq.update({"feats_%d" % t: feats, "grouped_%d" % t: grouped, "gout_%d" % t: gout,
                       "ggrad_%d" % t: o.group_points_grad(gout, idx, n)})
    save("ops_ballquery_group.npz", **bq)

    # ---- gather (+grad)
    pts = g.standard_normal((2, 6, 500)).astype(np.float32)
    gi = g.integers(0, 500, (2, 77)).astype(np.int32)
    gi[0, :5] = gi[0, 5]  # repeated indices: the grad must accumulate
    gg = g.standard_normal((2, 6, 77)).astype(np.float32)
    save("ops_gather.npz", points=pts, idx=gi, out=o.gather_points(pts, gi), grad_out=gg,
         grad=o.gather_points_grad(gg, gi, 500))

    # ---- three_nn / three_interpolate (+ both backward behaviours)
    unk = synth.cloud_uniform(2, 700, 2.0, seed=20)
    kn = synth.cloud_uniform(2, 300, 2.0, seed=21)
    kn[0, 10] = kn[0, 3]  # duplicate known point: earliest index must win the tie
    unk[0, 0] = kn[0, 3]
    d2, ni = o.three_nn(unk, kn)
    d2s, nis = o.three_nn(unk[:, :9], kn[:, :2])  # fewer than 3 known points -> +inf / index 0
    dist = np.sqrt(d2)
    w = 1.0 / (dist + 1e-8)
    w = (w / w.sum(axis=2, keepdims=True)).astype(np.float32)
    feats = g.standard_normal((2, 7, 300)).astype(np.float32)
    interp = o.three_interpolate(feats, ni, w)
    gout = g.standard_normal(interp.shape).astype(np.float32)
    save("ops_interp.npz", unknown=unk, known=kn, dist2=d2, idx=ni, dist2_small=d2s,
         idx_small=nis, weight=w, feats=feats, interp=interp, grad_out=gout,
         grad=o.three_interpolate_grad(gout, ni, w, 300),
         grad_as_executed_by_reference=o.three_interpolate_grad_as_executed(gout, ni, w, 300))


def part2():
    if not have_reference():
        print("oracle/_ref not built (no /root/reference?) -- skipping part 2")
        return
    ref = Reference()
    out = {}
    for tag, (a, b) in {"oriented": synth.boxes_pair(64, seed=0),
                        "aligned": synth.boxes_pair(64, seed=1, axis_aligned=True),
                        "kat": synth.box_kats()}.items():
        out["a_" + tag] = a
        out["b_" + tag] = b
        out["overlap_" + tag] = ref.box_overlap(a, b)
        out["iou_bev_" + tag] = ref.boxes_iou_bev_cpu(a, b)
    save("iou_bev_cpu_ref.npz", **out)


def part3():
    a, b = synth.boxes_pair(96, seed=5)
    ka, kb = synth.box_kats()
    boxes, scores = synth.boxes_scored(300, seed=6, spread=5.0)
    out = {"a": a, "b": b, "iou3d": o.boxes_iou3d(a, b), "ka": ka, "kb": kb,
           "iou3d_kat": o.boxes_iou3d(ka, kb), "nms_boxes": boxes}
    for thr in (0.25, 0.5):
        keep, mask = o.nms(boxes, thr)
        keepn, maskn = o.nms_normal(boxes, thr)
        tag = str(thr).replace(".", "p")
        out.update({"keep_" + tag: keep, "mask_" + tag: mask, "keepn_" + tag: keepn,
                    "maskn_" + tag: maskn})
    save("iou3d_nms_oracle.npz", **out)


if __name__ == "__main__":
    part1()
    part2()
    part3()
