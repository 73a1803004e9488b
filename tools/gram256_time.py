"""us of pool_gram_backward (csrc/mlp_pool_gram256.hip, all its launches) at one layer's shape, for A/B
runs of library variants (PN2_LIB_SUFFIX=_x).    python tools/gram256_time.py [sa2|sa3|sa4]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "sa2"
b, m, ns = {"sa2": (8, 1024, 32), "sa3": (8, 512, 16), "sa4": (8, 256, 16)}[which]
g = torch.Generator().manual_seed(1)
y2 = (torch.randn(b, 128, m, ns, generator=g) * 1.3 + 0.2).to(dev)
w3 = (torch.randn(256, 128, generator=g) / 11).to(dev)
g2, be2 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
g3, be3 = torch.rand(256, generator=g).to(dev) + 0.5, torch.randn(256, generator=g).to(dev) * 0.3
z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
c2 = K.bn_coefficients(y2, g2, be2, *z(128), 0.1, 1e-5, True)
y3, mean3, invstd3, sc3, sh3, ext = K.gemm_forward_bn(w3, y2, (c2[2], c2[3]), g3, be3, *z(256), 0.1, 1e-5, pool=True)
pooled, argmax, ymax = K.pool_from_extrema(ext, sc3, sh3)
dpooled = torch.randn(b, 256, m, generator=g).to(dev)
_, _, coef3 = K.bn_relu_pool_backward_stats(y3, dpooled, argmax, ymax, g3, sc3, sh3, mean3, invstd3, True)
t = [bench.time_op(lambda: K.pool_gram_backward(w3, y2, c2, g2, coef3, (mean3, invstd3, sc3, sh3), dpooled,
                                                argmax, ymax, ns, True), iters=5, warm=2) for _ in range(3)]
print("lib%s %s us=%s" % (os.environ.get("PN2_LIB_SUFFIX", ""), which, " ".join("%.1f" % v for v in t)))
