"""Per-shape timing of the shared-MLP GEMM primitives (forward / dgrad / wgrad, operand modes as
used by the train step) at the BASELINE config-2 layer shapes; prints us and TFLOP/s."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
importlib.import_module("3dioumatch_amd")
K = importlib.import_module("pointnet2._mlp_ext")
import bench  # noqa: E402  (time_op with graph replay)

dev = torch.device("cuda:0")
B = 8
LAYERS = [("sa1", 131072, [(64, 4), (64, 64), (128, 64)]),
          ("sa2", 32768, [(128, 131), (128, 128), (256, 128)]),
          ("sa3", 8192, [(128, 259), (128, 128), (256, 128)]),
          ("grid", 32768, [(128, 259), (128, 128), (128, 128)]),
          ("fp2", 1024, [(256, 512), (256, 256)])]
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for name, r, mk in LAYERS:
    for li, (m, k) in enumerate(mk):
        w = torch.randn(m, k, device=dev) / k ** 0.5
        x = torch.randn(B, k, r, device=dev)
        y = torch.randn(B, m, r, device=dev)
        dz = torch.randn(B, m, r, device=dev)
        vec = lambda n: torch.rand(n, device=dev) + 0.5  # noqa: E731
        coeff_k = (vec(k), vec(k))
        fly = (y, dz, vec(m), vec(m), vec(m), vec(m), torch.rand(m, 3, device=dev))
        flops = 2.0 * B * m * k * r
        t_f = bench.time_op(lambda: K.gemm_forward(w, x, None if li == 0 else coeff_k), iters=5, warm=2)
        t_d = bench.time_op(lambda: K.gemm_dgrad(w, fly=fly), iters=5, warm=2)
        t_w = bench.time_op(lambda: K.gemm_wgrad(m, k, x, None if li == 0 else coeff_k, fly=fly), iters=5, warm=2)
        if "--direct" in sys.argv:  # the same GEMMs on a materialised dy / plain x (no transforms)
            t_dd = bench.time_op(lambda: K.gemm_dgrad(w, dy=dz), iters=5, warm=2)
            t_wd = bench.time_op(lambda: K.gemm_wgrad(m, k, x, None, dy=dz), iters=5, warm=2)
            print("      direct operands: dgrad %7.1f us (fly %7.1f)   wgrad %7.1f us (fly %7.1f)" % (t_dd, t_d, t_wd, t_w))
        tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
        print("%-5s L%d M=%3d K=%3d R=%6d | fwd %7.1f us %5.1f TF | dgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF"
              % (name, li, m, k, r, t_f, flops / t_f * 1e-6, t_d, flops / t_d * 1e-6, t_w, flops / t_w * 1e-6))
print("totals (us):", {k: round(v, 1) for k, v in tot.items()})
