"""Run every hot-path operator a few times at the BASELINE config-2 shapes -- meant to be
wrapped by rocprofv3 (`--kernel-trace --stats`) so that per-kernel device durations can be read
without any host launch overhead in them:

    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ops -- \
        python tools/op_bench.py [iters]
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
ext = importlib.import_module("pointnet2._ext")
ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
synth = importlib.import_module("3dioumatch_amd.synth")

B, N = 8, 40000
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
xyz = torch.from_numpy(synth.cloud_uniform(B, N, synth.cube_side(N, 0.2, 64), seed=1)).to(dev)
flipped = xyz.transpose(1, 2).contiguous()
feat = torch.rand(B, 1, N, device=dev)
for it in range(iters):
    inds = ext.furthest_point_sampling(xyz, 2048)
    new_xyz = ext.gather_points(flipped, inds).transpose(1, 2).contiguous()
    idx = ext.ball_query(new_xyz, xyz, 0.2, 64)
    gx = ext.group_points(flipped, idx)
    gf = ext.group_points(feat, idx)
    ext.query_and_group(new_xyz, xyz, feat, 0.2, 64, True)
    inds2 = ext.furthest_point_sampling(new_xyz, 1024)
    new2 = ext.gather_points(new_xyz.transpose(1, 2).contiguous(), inds2).transpose(1, 2).contiguous()
    idx2 = ext.ball_query(new2, new_xyz, 0.4, 32)
    f128 = torch.rand(B, 128, 2048, device=dev)
    g2 = ext.group_points(f128, idx2)
    ext.group_points_grad(torch.rand_like(g2), idx2, 2048)
    ext.furthest_point_sampling(new2, 512)
    ext.furthest_point_sampling(new2[:, :512].contiguous(), 256)
    grid = torch.rand(B, 32768, 3, device=dev) * 3
    seeds = torch.rand(B, 1024, 3, device=dev) * 3
    d2, nidx = ext.three_nn(grid, seeds)
    w = torch.rand(B, 32768, 3, device=dev)
    f256 = torch.rand(B, 256, 1024, device=dev)
    out = ext.three_interpolate(f256, nidx, w)
    a, b = synth.boxes_pair(2048, seed=3)
    ut.boxes_iou3d_gpu(torch.from_numpy(a).to(dev), torch.from_numpy(b[:512]).to(dev))
    ut.boxes_iou3d_gpu(torch.from_numpy(a[:256]).to(dev), torch.from_numpy(b[:256]).to(dev))
torch.cuda.synchronize()
print("op_bench done", iters)
