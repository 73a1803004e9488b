"""Time of the product furthest-point-sampling entry points on the train step's shapes (bench
scene, B = 8): SA1 40000 -> 2048 (bucketed tier, with and without the cell-list by-product) and
the register tiers of SA2..SA4.

    python tools/micro/fps_time.py
"""
import json
import os
import sys
from importlib import import_module

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

ext = import_module("3dioumatch_amd.dropin.pointnet2._ext")
data_mod = import_module("3dioumatch_amd.votenet.data")
cfg_mod = import_module("3dioumatch_amd.votenet.config")
dev = torch.device("cuda:0")
B = 8
scene = data_mod.make_batch(B, 40000, cfg_mod.scannet_config(), seed=100)["point_clouds"][:, :, :3].contiguous().to(dev)
res = {}
res["sa1_40000_2048_us"] = round(bench.time_op(lambda: ext.furthest_point_sampling(scene, 2048), iters=5, warm=2), 1)
res["sa1_40000_2048_with_cell_lists_us"] = round(
    bench.time_op(lambda: ext.furthest_point_sampling_with_grid(scene, 2048, 0.2), iters=5, warm=2), 1)
inds = ext.furthest_point_sampling(scene, 2048).long()
cur = torch.gather(scene, 1, inds[:, :, None].expand(-1, -1, 3)).contiguous()
for n, m in ((2048, 1024), (1024, 512), (512, 256)):
    res["fps_%d_%d_us" % (n, m)] = round(bench.time_op(lambda: ext.furthest_point_sampling(cur, m), iters=10, warm=2), 1)
    sel = ext.furthest_point_sampling(cur, m).long()
    cur = torch.gather(cur, 1, sel[:, :, None].expand(-1, -1, 3)).contiguous()
sun = data_mod.make_batch(16, 20000, cfg_mod.sunrgbd_config(), seed=100)["point_clouds"][:, :, :3].contiguous().to(dev)
res["sunrgbd_16x20000_2048_us"] = round(bench.time_op(lambda: ext.furthest_point_sampling(sun, 2048), iters=5, warm=2), 1)
res["checksum"] = int(inds.sum().item())
print(json.dumps(res))
