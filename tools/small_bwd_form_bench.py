"""Backward of a small layer behind BatchNorm + ReLU, two forms (us per layer, 16 per graph replay):
  fly : mlp_bn_relu_backward_stats, then ONE pair launch that forms dy inside its operand loads
        (every 64-row tile of the data gradient and every tile of the weight gradient re-forms it)
  dy  : mlp_bn_relu_backward (sums + the dy tensor written once), then the pair launch on dy
    python tools/small_bwd_form_bench.py [out.json]
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
SHAPES = [("fp2_l1", 256, 512, 1024), ("fp2_l2", 256, 256, 1024), ("fp1_l1", 256, 512, 512), ("fp1_l2", 256, 256, 512),
          ("vote_l2", 256, 256, 1024), ("prop_l2", 128, 128, 256), ("pre_sa2", 128, 131, 3072),
          ("pre_sa3", 128, 259, 1536), ("pre_sa4", 128, 259, 768), ("pre_vote", 128, 131, 1280)]
REP = 16
res = {}
for name, m, k, r in SHAPES:
    w = torch.randn(m, k, device=dev) / k ** 0.5
    x = torch.randn(B, k, r, device=dev)
    y = torch.randn(B, m, r, device=dev)
    dz = torch.randn(B, m, r, device=dev)
    vec = lambda n: torch.rand(n, device=dev) + 0.5  # noqa: E731
    ck = (vec(k), vec(k))
    gamma, scale, shift, mean, invstd = vec(m), vec(m), vec(m), vec(m), vec(m)

    def many(fn):
        def run():
            for _ in range(REP):
                fn()
        return bench.time_op(run, iters=5, warm=2) / REP

    def form_fly(xc):
        _, _, coef = K.bn_relu_backward_stats(y, dz, gamma, scale, shift, mean, invstd, True)
        K.gemm_backward_small(w, x, xc, fly=(y, dz, scale, shift, mean, invstd, coef))

    def form_dy(xc):
        dy, _, _ = K.bn_relu_backward(y, dz, gamma, scale, shift, mean, invstd, True)
        K.gemm_backward_small(w, x, xc, dy=dy)

    with K.deferred_weight_reductions():
        row = {"fly_bnrelu_input": many(lambda: form_fly(ck)), "dy_bnrelu_input": many(lambda: form_dy(ck)),
               "fly_direct_input": many(lambda: form_fly(None)), "dy_direct_input": many(lambda: form_dy(None))}
    res[name] = {"m": m, "k": k, "cols_per_cloud": r, **{a: round(v, 2) for a, v in row.items()}}
    print("%-8s M=%3d K=%3d R=%5d | input relu(bn(.)): fly %5.1f  dy %5.1f | input direct: fly %5.1f  dy %5.1f us"
          % (name, m, k, r, row["fly_bnrelu_input"], row["dy_bnrelu_input"], row["fly_direct_input"], row["dy_direct_input"]))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
