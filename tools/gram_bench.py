"""The backward of SA1's pooled last layer (128 x 64, B = 8, m = 2048, ns = 64) two ways: from the
Gram matrix of its input (csrc/mlp_pool_gram.hip: y3 neither read nor stored) against the one-pass
kernel that rebuilds dy3 from the stored y3 (mlp_gemm_backward_fused).  us per call (HIP-graph
replays).    python tools/gram_bench.py [out.json]"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
b, m, ns = 8, 2048, 64
g = torch.Generator().manual_seed(1)
y2 = (torch.randn(b, 64, m, ns, generator=g) * 1.3 + 0.2).to(dev)
w3 = (torch.randn(128, 64, generator=g) / 8).to(dev)
g2, be2 = torch.rand(64, generator=g).to(dev) + 0.5, torch.randn(64, generator=g).to(dev) * 0.3
g3, be3 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
c2 = K.bn_coefficients(y2, g2, be2, *z(64), 0.1, 1e-5, True)
y3, mean3, invstd3, sc3, sh3, ext = K.gemm_forward_bn(w3, y2, (c2[2], c2[3]), g3, be3, *z(128), 0.1, 1e-5, pool=True)
pooled, argmax, ymax = K.pool_from_extrema(ext, sc3, sh3)
dpooled = torch.randn(b, 128, m, generator=g).to(dev)
_, _, coef3 = K.bn_relu_pool_backward_stats(y3, dpooled, argmax, ymax, g3, sc3, sh3, mean3, invstd3, True)
out = {}
for rep in range(2):
    out["stored_y3_us_%d" % rep] = round(bench.time_op(lambda: K.gemm_backward_fused(
        w3, y2, (c2[2], c2[3]), pooled=(y3, dpooled, argmax, sc3, sh3, mean3, invstd3, coef3),
        xstats=(c2[0], c2[1], g2, True)), iters=5, warm=2), 1)
    out["gram_us_%d" % rep] = round(bench.time_op(lambda: K.pool_gram_backward(
        w3, y2, c2, g2, coef3, (mean3, invstd3, sc3, sh3), dpooled, argmax, ymax, ns, True), iters=5, warm=2), 1)
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
