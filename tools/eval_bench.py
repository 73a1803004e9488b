"""Time the AP evaluation (votenet/eval_det.py) on a ScanNet-val sized synthetic detection set:
312 scans, 18 classes, per-class proposals (every kept box listed once per class, as
parse_predictions does with per_class_proposal), ~14 ground-truth boxes per scan.

    python tools/eval_bench.py [--scans 312] [--keep 60] [--cpu-sample 20000]

Prints one JSON line: detections, (detection, ground truth) pairs, the kernel's time (HIP events),
the whole eval_det call (host grouping + kernel + marking) and, as the CPU baseline, the oracle's
C restatement of the same matching (1 thread) on a bounded sample of the detections.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
D = importlib.import_module("3dioumatch_amd.votenet.eval_det")


def corners(rng, n, ctr):
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1.])
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1.])
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1.])
    l, h, w = (rng.uniform(0.3, 1.6, (n, 1)) for _ in range(3))
    ang = rng.uniform(-np.pi, np.pi, (n, 1))
    x, y, z = sx * l / 2, sy * h / 2, sz * w / 2
    c, s = np.cos(ang), np.sin(ang)
    return np.stack([c * x + s * z + ctr[:, 0:1], y + ctr[:, 1:2], -s * x + c * z + ctr[:, 2:3]],
                    -1).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=312)
    ap.add_argument("--keep", type=int, default=60)
    ap.add_argument("--classes", type=int, default=18)
    ap.add_argument("--cpu-sample", type=int, default=20000)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    pred_all, gt_all = {}, {}
    for s in range(args.scans):
        ng = int(rng.integers(6, 24))
        gctr = rng.uniform(-3, 3, (ng, 3)) * [1, 0.2, 1]
        gbox = corners(rng, ng, gctr)
        gcls = rng.integers(0, args.classes, ng)
        gt_all[s] = [(int(c), b) for c, b in zip(gcls, gbox)]
        pctr = gctr[rng.integers(0, ng, args.keep)] + rng.normal(0, 0.15, (args.keep, 3))
        pbox = corners(rng, args.keep, pctr)
        prob = rng.random((args.keep, args.classes)).astype(np.float32)
        pred_all[s] = [(c, pbox[j], prob[j, c]) for c in range(args.classes) for j in range(args.keep)]

    torch.zeros(1, device="cuda:0")
    t0 = time.perf_counter()
    pred, gt = D._group(pred_all, gt_all)
    det, score, begin, count, gts, det_slice, npos = D._flatten(pred, gt)
    t_group = time.perf_counter() - t0
    dd, db, dc, dg = (torch.from_numpy(a).cuda() for a in (det, begin, count, gts))
    for _ in range(2):
        D.corners_best_match_gpu(dd, db, dc, dg)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ov, jm = D.corners_best_match_gpu(dd, db, dc, dg)
    e1.record()
    torch.cuda.synchronize()
    kernel_ms = e0.elapsed_time(e1) / 10
    t0 = time.perf_counter()
    rec, prec, apv = D.eval_det(pred_all, gt_all, ovthresh=0.25, device="cuda:0")
    t_all = time.perf_counter() - t0

    from oracle.oracle import Oracle  # checker, timed here as the CPU baseline only
    k = min(args.cpu_sample, len(det))
    t0 = time.perf_counter()
    wov, wjm = Oracle().best_match(det[:k], begin[:k], count[:k], gts)
    t_cpu = time.perf_counter() - t0
    assert np.array_equal(wjm, jm.cpu().numpy()[:k])
    pairs = int(count.sum())
    print(json.dumps({
        "what": "AP evaluation, %d scans x %d classes x %d kept boxes" % (args.scans, args.classes, args.keep),
        "detections": int(len(det)), "pairs": pairs, "kernel_ms": round(kernel_ms, 3),
        "pairs_per_s_gpu": round(pairs / (kernel_ms * 1e-3)),
        "host_group_flatten_s": round(t_group, 3), "eval_det_total_s": round(t_all, 3),
        "cpu_oracle_1thread_pairs_per_s": round(int(count[:k].sum()) / t_cpu),
        "mAP": float(np.mean(list(apv.values())))}))


if __name__ == "__main__":
    main()
