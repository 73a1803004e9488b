"""The train step's SMALL shared-MLP GEMMs (FP modules, vote / proposal / IoU heads, the
pre-gather first layers): us per launch, forward (direct and BN/ReLU operand), data gradient and
weight gradient with the on-the-fly dY operand.  Timed as 16 back-to-back launches inside one
HIP-graph replay (what the step does), so the figure includes the launch-to-launch gap.

    python tools/small_gemm_bench.py [out.json]
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
# (name, m, k, columns per cloud)
SHAPES = [("fp2_l1", 256, 512, 1024), ("fp2_l2", 256, 256, 1024), ("fp1_l1", 256, 512, 512),
          ("vote_l1", 256, 256, 1024), ("vote_l3", 259, 256, 1024), ("prop_l1", 128, 128, 256),
          ("prop_l3", 79, 128, 256), ("pre_sa2", 128, 131, 3072), ("pre_sa3", 128, 259, 1536),
          ("pre_sa4", 128, 259, 768), ("pre_vote", 128, 131, 1280)]
REP = 16
res = {}
for name, m, k, r in SHAPES:
    w = torch.randn(m, k, device=dev) / k ** 0.5
    x = torch.randn(B, k, r, device=dev)
    y = torch.randn(B, m, r, device=dev)
    dz = torch.randn(B, m, r, device=dev)
    vec = lambda n: torch.rand(n, device=dev) + 0.5  # noqa: E731
    ck = (vec(k), vec(k))
    fly = (y, dz, vec(m), vec(m), vec(m), vec(m), torch.rand(m, 3, device=dev))

    def many(fn):
        def run():
            for _ in range(REP):
                fn()
        return bench.time_op(run, iters=5, warm=2) / REP

    row = {"fwd_direct": many(lambda: K.gemm_forward(w, x)),
           "fwd_bnrelu": many(lambda: K.gemm_forward(w, x, ck)),
           "dgrad_fly": many(lambda: K.gemm_dgrad(w, fly=fly)),
           "dgrad_direct": many(lambda: K.gemm_dgrad(w, dy=dz)),
           "wgrad_fly": many(lambda: K.gemm_wgrad(m, k, x, ck, fly=fly)),
           "wgrad_direct": many(lambda: K.gemm_wgrad(m, k, x, None, dy=dz))}
    res[name] = {"m": m, "k": k, "cols_per_cloud": r, **{a: round(v, 2) for a, v in row.items()}}
    gf = 2.0 * B * m * k * r * 1e-9
    print("%-9s M=%3d K=%3d R=%5d %5.2f GF | fwd %5.1f / %5.1f  dgrad fly %5.1f direct %5.1f  wgrad fly %5.1f direct %5.1f us"
          % (name, m, k, r, gf, row["fwd_direct"], row["fwd_bnrelu"], row["dgrad_fly"], row["dgrad_direct"],
             row["wgrad_fly"], row["wgrad_direct"]))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
