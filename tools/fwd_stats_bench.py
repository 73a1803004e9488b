"""Forward shared-MLP layers AS THE TRAIN STEP RUNS THEM (GEMM + BatchNorm statistics from the
epilogue, + the max over nsample of an SA module's last layer) at the config-2 shapes; us per call
and TFLOP/s.  `python tools/fwd_stats_bench.py`"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
K = importlib.import_module("pointnet2._mlp_ext")
import bench  # noqa: E402

dev = torch.device("cuda:0")
B = 8
LAYERS = [("sa1", (2048, 64), [(64, 64, False), (128, 64, True)]),
          ("sa2", (1024, 32), [(128, 131, False), (128, 128, False), (256, 128, True)]),
          ("sa3", (512, 16), [(128, 259, False), (128, 128, False), (256, 128, True)]),
          ("sa4", (256, 16), [(128, 259, False), (128, 128, False), (256, 128, True)]),
          ("grid", (256 * 4, 36), [(128, 259, False), (128, 128, False), (128, 128, True)])]
total = 0.0
for name, (mp, ns), mk in LAYERS:
    for li, (m, k, pool) in enumerate(mk):
        w = torch.randn(m, k, device=dev) / k ** 0.5
        x = torch.randn(B, k, mp, ns, device=dev)
        coeff = (torch.rand(k, device=dev) + 0.5, torch.rand(k, device=dev))
        g = torch.rand(m, device=dev) + 0.5
        bt = torch.rand(m, device=dev)
        rm, rv = torch.zeros(m, device=dev), torch.ones(m, device=dev)
        us = bench.time_op(lambda: K.gemm_forward_bn(w, x, coeff, g, bt, rm, rv, 0.1, 1e-5, pool=pool),
                           iters=5, warm=2)
        fl = 2.0 * m * k * B * mp * ns
        total += us
        print("%-5s L%d M=%3d K=%3d cols=%8d pool=%d | %8.1f us %6.1f TF" % (name, li, m, k, B * mp * ns, pool, us,
                                                                         fl / us / 1e6))
print("total (us): %.1f" % total)
