"""Round-6 timings of the reference-surface stragglers: group_inverse at SA2,
group_points_grad at SA1 (C = 4), device NMS of 1024 boxes incl. host, iou3d 2048 x 512.
    python tools/r6_ops_time.py [out.json]"""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
ext = importlib.import_module("pointnet2._ext")
ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
synth = importlib.import_module("3dioumatch_amd.synth")
import bench  # noqa: E402

dev = torch.device("cuda:0")
B, N = 8, 40000
out = {}
xyz = torch.from_numpy(synth.cloud_uniform(B, N, synth.cube_side(N, 0.2, 64), seed=1)).to(dev)
inds = ext.furthest_point_sampling(xyz, 2048)
new_xyz = ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
idx = ext.ball_query(new_xyz, xyz, 0.2, 64)
g4 = torch.rand(B, 4, 2048, 64, device=dev)
out["group_grad_sa1_c4_us"] = round(bench.time_op(lambda: ext.group_points_grad(g4, idx, N)), 2)
out["ball_query_sa1_cold_us"] = round(bench.time_op(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64)), 2)
inds2 = ext.furthest_point_sampling(new_xyz, 1024)
new2 = ext.gather_points(new_xyz.transpose(1, 2).contiguous(), inds2).transpose(1, 2).contiguous()
idx2 = ext.ball_query(new2, new_xyz, 0.4, 32)
out["group_inverse_sa2_us"] = round(bench.time_op(lambda: ext.group_inverse(idx2, 2048)), 2)
a, b = synth.boxes_pair(2048, seed=3)
a_d, b_d = torch.from_numpy(a).to(dev), torch.from_numpy(b[:512]).to(dev)
out["iou3d_2048x512_us"] = round(bench.time_op(lambda: ut.boxes_iou3d_gpu(a_d, b_d)), 2)
K = 1024
a, b = synth.boxes_pair(K, seed=3)
a_d = torch.from_numpy(a).to(dev)
scores = torch.rand(K, device=dev)
for _ in range(3):
    keep, _ = ut.nms_gpu(a_d, scores, 0.25)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    keep, _ = ut.nms_gpu(a_d, scores, 0.25)
torch.cuda.synchronize()
out["nms3d_1024_incl_host_us"] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
out["nms_kept"] = int(keep.numel())
# the device part alone: mask + scan, events around eager launches
cu = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_cuda")
order = scores.sort(0, descending=True)[1]
sorted_boxes = a_d[order].contiguous()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cu.nms_device(sorted_boxes, 0.25)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    cu.nms_device(sorted_boxes, 0.25)
e1.record()
torch.cuda.synchronize()
out["nms3d_1024_device_calls_us"] = round(e0.elapsed_time(e1) * 100, 1)
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
