"""us per call of the no-store pooled forward (GEMM + finalize), for A/B of library variants
(PN2_LIB_SUFFIX).   python tools/pool_fwd256_time.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
out = []
for b, m, ns in ((8, 1024, 32), (12, 1024, 32), (8, 512, 16)):
    g = torch.Generator().manual_seed(b + m + ns)
    y2 = (torch.randn(b, 128, m, ns, generator=g) * 1.3 + 0.2).to(dev)
    w3 = (torch.randn(256, 128, generator=g) / 11).to(dev)
    g2, be2 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
    g3, be3 = torch.rand(256, generator=g).to(dev) + 0.5, torch.randn(256, generator=g).to(dev) * 0.3
    z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    c2 = K.bn_coefficients(y2, g2, be2, *z(128), 0.1, 1e-5, True)
    fwd = lambda: K.gemm_forward_bn(w3, y2, (c2[2], c2[3]), g3, be3, *z(256), 0.1, 1e-5, pool=True, store=False)
    big = torch.empty(160 << 20, device=dev)  # 640 MB: written between calls, the input then comes from HBM
    t_fill = bench.time_op(lambda: big.fill_(1.0), iters=5, warm=2)
    t_cold = bench.time_op(lambda: (big.fill_(1.0), fwd()), iters=5, warm=2) - t_fill
    out.append("%d/%d/%d: %.1f %.1f (input from HBM: %.1f)" % (b, m, ns, bench.time_op(fwd, iters=5, warm=2),
                                                               bench.time_op(fwd, iters=5, warm=2), t_cold))
    del big
print("lib[%s] no-store forward us: %s" % (os.environ.get("PN2_LIB_SUFFIX", ""), " | ".join(out)))
