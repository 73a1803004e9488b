"""us per call of a (128, 128) forward with statistics, input hot (replayed back to back) and from HBM (640 MB
written between calls), both kernels.   python tools/fwd128_time.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
for b, m, ns, pool in ((8, 1024, 32, False), (8, 512, 64, True)):
    g = torch.Generator().manual_seed(b + m + ns)
    x = (torch.randn(b, 128, m, ns, generator=g) * 1.3 + 0.2).to(dev)
    w = (torch.randn(128, 128, generator=g) / 11).to(dev)
    g2, be2 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
    g3, be3 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
    z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    c2 = K.bn_coefficients(x, g2, be2, *z(128), 0.1, 1e-5, True)
    fwd = lambda: K.gemm_forward_bn(w, x, (c2[2], c2[3]), g3, be3, *z(128), 0.1, 1e-5, pool=pool)
    big = torch.empty(160 << 20, device=dev)
    t_fill = bench.time_op(lambda: big.fill_(1.0), iters=5, warm=2)
    for env in ("0", "1"):
        os.environ["MLP_FWD128"] = env
        hot = bench.time_op(fwd, iters=5, warm=2)
        cold = bench.time_op(lambda: (big.fill_(1.0), fwd()), iters=5, warm=2) - t_fill
        print("b %d m %d ns %d pool %d MLP_FWD128=%s: hot %.1f us, input from HBM %.1f us" % (b, m, ns, pool, env, hot, cold), flush=True)
    del big, x
    torch.cuda.empty_cache()
