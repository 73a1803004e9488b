"""Backward GEMMs of the set-abstraction layers at the train step's shapes: the fused one-pass
kernel (mlp_gemm_backward_fused, which also leaves the BatchNorm-backward sums of the layer below)
against the two separate on-the-fly GEMMs plus that layer's statistics pass.

    python tools/bwd_bench.py [--json out.json]
Per layer: microseconds, algorithmic bytes (fused: (2M+2K) floats per column; pooled: (M+2K)),
flops (4*M*K per column), and the fractions of 8 TB/s and of the 157 TFLOP/s fp32 MFMA peak.
"""
import importlib
import json
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
LAYERS = [("sa1_l2", 64, 64, 2048, 64, False), ("sa1_l3", 128, 64, 2048, 64, True),
          ("sa2_l1", 128, 131, 1024, 32, False), ("sa2_l2", 128, 128, 1024, 32, False),
          ("sa2_l3", 256, 128, 1024, 32, True), ("sa3_l1", 128, 259, 512, 16, False),
          ("sa3_l2", 128, 128, 512, 16, False), ("sa3_l3", 256, 128, 512, 16, True),
          ("sa4_l1", 128, 259, 256, 16, False), ("sa4_l2", 128, 128, 256, 16, False),
          ("sa4_l3", 256, 128, 256, 16, True)]
only = [a for a in sys.argv[1:] if not a.startswith("--") and not a.endswith(".json")]
res = {}
for name, m, k, groups, ns, pooled in LAYERS:
    if only and name not in only:
        continue
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(m, k, generator=g) / k ** 0.5).to(dev)
    x = torch.randn(B, k, groups, ns, device=dev)
    y = torch.randn(B, m, groups, ns, device=dev)
    gamma, beta = torch.rand(m, device=dev) + 0.5, torch.randn(m, device=dev) * 0.3
    rm, rv = torch.zeros(m, device=dev), torch.ones(m, device=dev)
    mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, rm, rv, 0.1, 1e-5, True)
    if pooled:
        _, argmax, ymax = K.bn_relu_pool(y, scale, shift)
        dpooled = torch.randn(B, m, groups, device=dev)
        _, _, coef = K.bn_relu_pool_backward_stats(y, dpooled, argmax, ymax, gamma, scale, shift, mean,
                                                   invstd, True)
        kw = dict(pooled=(y, dpooled, argmax, scale, shift, mean, invstd, coef))
    else:
        dz = torch.randn(B, m, groups, ns, device=dev)
        _, _, coef = K.bn_relu_backward_stats(y, dz, gamma, scale, shift, mean, invstd, True)
        kw = dict(fly=(y, dz, scale, shift, mean, invstd, coef))
    xc = xs = None
    if k % 32 == 0:
        xg, xb = torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev) * 0.3
        xm, xi, xsc, xsh = K.bn_coefficients(x, xg, xb, torch.zeros(k, device=dev), torch.ones(k, device=dev),
                                             0.1, 1e-5, True)
        xc, xs = (xsc, xsh), (xm, xi, xg, True)
    cols = B * groups * ns
    byts = 4 * cols * ((m if pooled else 2 * m) + 2 * k)
    flops = 4 * m * k * cols
    fused = bench.time_op(lambda: K.gemm_backward_fused(w, x, xc, xstats=xs, **kw), iters=10)
    two = bench.time_op(lambda: (K.gemm_dgrad(w, **kw), K.gemm_wgrad(m, k, x, xc, **kw)), iters=10)
    stats = 0.0
    if xs is not None:  # the stats pass over (x, dx) of the layer below that the fused kernel absorbs
        dxt = torch.randn_like(x)
        stats = bench.time_op(lambda: K.bn_relu_backward_stats(x, dxt, xg, xsc, xsh, xm, xi, True), iters=10)
    res[name] = {"M": m, "K": k, "columns": cols, "fused_us": round(fused, 1), "two_gemms_us": round(two, 1),
                 "stats_pass_below_us": round(stats, 1),
                 "hbm_frac": round(byts / (fused * 1e-6) / 8e12, 3),
                 "mfma_frac": round(flops / (fused * 1e-6) / 157e12, 3)}
    print(name, res[name], flush=True)
if "--json" in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
