"""The weight gradient of the IoU branch's first layer (128 x 259 over 8 x 16384 columns, input
without gradient): the persistent weight-gradient half of the fused backward kernel (what the step
runs) against the stand-alone weight-gradient kernel, gradient operand formed on the fly or given.
    python tools/iou_l1_wgrad_bench.py
"""
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
b, m, k, r = 8, 128, 259, 16384
w = torch.randn(m, k, device=dev) / k ** 0.5
x = torch.randn(b, k, r, device=dev)
y = torch.randn(b, m, r, device=dev)
dz = torch.randn(b, m, r, device=dev)
vec = lambda n: torch.rand(n, device=dev) + 0.5  # noqa: E731
gamma, scale, shift, mean, invstd = vec(m), vec(m), vec(m), vec(m), vec(m)
_, _, coef = K.bn_relu_backward_stats(y, dz, gamma, scale, shift, mean, invstd, True)
fly = (y, dz, scale, shift, mean, invstd, coef)
with K.deferred_weight_reductions():
    t_fused = bench.time_op(lambda: K.gemm_backward_fused(w, x, None, fly, None, None, need_dx=False), iters=10, warm=2)
    t_wgrad = bench.time_op(lambda: K.gemm_wgrad(m, k, x, None, fly=fly), iters=10, warm=2)

    def two():
        dy, _, _ = K.bn_relu_backward(y, dz, gamma, scale, shift, mean, invstd, True)
        K.gemm_wgrad(m, k, x, None, dy=dy)
    t_dy = bench.time_op(two, iters=10, warm=2)
    t_stats = bench.time_op(lambda: K.bn_relu_backward_stats(y, dz, gamma, scale, shift, mean, invstd, True), iters=10, warm=2)
a = K.gemm_backward_fused(w, x, None, fly, None, None, need_dx=False)[1]
c = K.gemm_wgrad(m, k, x, None, fly=fly)
print("fused wgrad-only %.1f us | stand-alone wgrad (fly) %.1f us | apply + stand-alone wgrad (dy) %.1f us (of which sums+apply ~%.1f + ...) | max diff %.2e of %.2e"
      % (t_fused, t_wgrad, t_dy, t_stats, float((a - c).abs().max()), float(a.abs().max())))
