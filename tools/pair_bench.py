"""Run only the north-star kernel pair (ball_query + group_points xyz + group_points feat at
B=8, N=40000, m=2048, nsample=64) a few times -- to be wrapped by rocprofv3 for PMC passes:

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p1 -o pair -- python tools/pair_bench.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p2 -o pair -- python tools/pair_bench.py
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
ext = importlib.import_module("pointnet2._ext")
synth = importlib.import_module("3dioumatch_amd.synth")

B, N = 8, 40000
dev = torch.device("cuda:0")
xyz = torch.from_numpy(synth.cloud_uniform(B, N, synth.cube_side(N, 0.2, 64), seed=1)).to(dev)
flipped = xyz.transpose(1, 2).contiguous()
feat = torch.rand(B, 1, N, device=dev)
inds = ext.furthest_point_sampling(xyz, 2048)
new_xyz = ext.gather_points(flipped, inds).transpose(1, 2).contiguous()
torch.cuda.synchronize()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    idx = ext.ball_query(new_xyz, xyz, 0.2, 64)
    gx = ext.group_points(flipped, idx)
    gf = ext.group_points(feat, idx)
torch.cuda.synchronize()
print("pair_bench done")
