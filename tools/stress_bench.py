"""BASELINE configs[4] (stress): B=32 clouds of N=80000 points, nsample=128, 1024 proposals,
1024x1024 IoU matrix + 3-D NMS -- per-operator device times on one GPU (a parity/scale case, not
the bench line).  Index outputs are checked against size-independent properties.

    python tools/stress_bench.py [out.json]
"""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
import bench  # noqa: E402

B, N, M, NS, K = 32, 80000, 2048, 128, 1024


def main():
    dev = torch.device("cuda:0")
    ext = importlib.import_module("pointnet2._ext")
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    synth = importlib.import_module("3dioumatch_amd.synth")
    xyz = torch.from_numpy(synth.cloud_uniform(B, N, synth.cube_side(N, 0.2, NS), seed=1)).to(dev)
    out = {"config": {"B": B, "N": N, "npoint": M, "nsample": NS, "proposals": K}}
    t = {}
    t["fps_80000_2048"] = bench.time_op(lambda: ext.furthest_point_sampling(xyz, M), iters=2, warm=1)
    inds = ext.furthest_point_sampling(xyz, M)
    assert int(inds.min()) >= 0 and int(inds.max()) < N and bool((inds[:, 0] == 0).all())
    assert all(len(set(row.tolist())) == M for row in inds[:2].cpu())  # distinct samples
    xyz_t = xyz.transpose(1, 2).contiguous()
    new_xyz = ext.gather_points(xyz_t, inds).transpose(1, 2).contiguous()
    t["ball_query_ns128"] = bench.time_op(lambda: ext.ball_query(new_xyz, xyz, 0.2, NS), iters=5, warm=1)
    idx = ext.ball_query(new_xyz, xyz, 0.2, NS)
    # every returned neighbour lies inside the ball; the centroid itself is always found
    grouped = ext.group_points(xyz_t, idx)
    d2 = ((grouped - new_xyz.transpose(1, 2).unsqueeze(-1)) ** 2).sum(1)
    assert float(d2.max()) < 0.2 * 0.2
    t["group_xyz"] = bench.time_op(lambda: ext.group_points(xyz_t, idx), iters=10, warm=2)
    feat = torch.randn(B, 1, N, device=dev)
    t["group_feat_c1"] = bench.time_op(lambda: ext.group_points(feat, idx), iters=10, warm=2)
    t["query_and_group_fused"] = bench.time_op(
        lambda: ext.query_and_group(new_xyz, xyz, feat, 0.2, NS, True), iters=5, warm=1)
    # ... and as a set-abstraction layer runs it: on the cell lists and query plans its own sampling
    # call leaves behind (no build between sampling and grouping; bench.py's `layer` form)
    linds, lists = ext.furthest_point_sampling_with_grid(xyz, M, 0.2)
    assert torch.equal(linds, inds)
    layer_form = "cell lists + query plans left by the sampling kernel"
    if lists is None:  # (N = 80 000 is outside the sampling kernel's by-product range: lists built once,
        lists = ext.build_grid(xyz, 0.2)  # ahead of the timed query; no query plans)
        layer_form = "cell lists built ahead of the query (pn2_grid_build), no query plans"
    else:
        lists.mark_centroids(new_xyz, linds)
    idx_l, grouped_l = ext.query_and_group(new_xyz, xyz, feat, 0.2, NS, True, None, lists)
    assert torch.equal(idx_l, idx)
    assert torch.allclose(grouped_l[:, :3], (grouped - new_xyz.transpose(1, 2).unsqueeze(-1)) / 0.2, rtol=0, atol=1e-5)
    t["query_and_group_layer"] = bench.time_op(
        lambda: ext.query_and_group(new_xyz, xyz, feat, 0.2, NS, True, None, lists), iters=5, warm=1)
    pair_bytes = 12 * B * N + 12 * B * M + 4 * B * M * NS + 2 * 4 * B * M * NS \
        + 4 * B * 3 * N + 4 * B * 3 * M * NS + 4 * B * N + 4 * B * M * NS
    pair_us = t["ball_query_ns128"] + t["group_xyz"] + t["group_feat_c1"]
    out["pair"] = {"algorithmic_bytes": pair_bytes, "us": round(pair_us, 1),
                   "GBps": round(pair_bytes / pair_us / 1e3, 1),
                   "fused_us": round(t["query_and_group_fused"], 1),
                   "fused_GBps": round(pair_bytes / t["query_and_group_fused"] / 1e3, 1),
                   "fused_frac_of_8TBs": round(pair_bytes / t["query_and_group_fused"] / 8e6, 4),
                   "layer_form": layer_form,
                   "layer_us": round(t["query_and_group_layer"], 1),
                   "layer_GBps": round(pair_bytes / t["query_and_group_layer"] / 1e3, 1),
                   "layer_frac_of_8TBs": round(pair_bytes / t["query_and_group_layer"] / 8e6, 4)}
    a, b = synth.boxes_pair(K, seed=3)
    a_d, b_d = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    t["iou3d_1024x1024"] = bench.time_op(lambda: ut.boxes_iou3d_gpu(a_d, b_d), iters=10, warm=2)
    iou = ut.boxes_iou3d_gpu(a_d, b_d)
    assert float(iou.min()) >= 0 and float(iou.max()) <= 1 + 1e-4
    scores = torch.rand(K, device=dev)
    import time
    for _ in range(3):  # warm-up: the first calls pay for lazily loaded sort kernels
        keep, _ = ut.nms_gpu(a_d, scores, 0.25)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        keep, _ = ut.nms_gpu(a_d, scores, 0.25)
    torch.cuda.synchronize()
    t["nms3d_1024_incl_host"] = (time.perf_counter() - t0) / 10 * 1e6
    kept = a_d[keep]
    rest = ut.boxes_iou3d_gpu(kept, kept)
    rest.fill_diagonal_(0)
    assert float(rest.max()) <= 0.25 + 1e-4  # survivors do not overlap above the threshold
    out["kernels_us"] = {k: round(v, 1) for k, v in t.items()}
    out["nms_kept"] = int(keep.numel())
    print(json.dumps(out))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
