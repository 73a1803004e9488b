"""The pooled last layer (256 x 128) of SA2 / SA3 / SA4 two ways, forward + backward:
  stored : forward GEMM that stores y3 (+ statistics + extrema), one-pass backward that rebuilds
           dy3 from the stored y3, the statistics pass of the layer below over (y2, da2)
  gram   : forward GEMM that stores nothing but statistics + extrema, backward from the Gram matrix
           of the layer's input (csrc/mlp_pool_gram256.hip: two passes over y2, sums included)
us per call (HIP-graph replays).    python tools/gram256_bench.py [out.json]"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
out = {}
for name, b, m, ns in (("sa2", 8, 1024, 32), ("semi_sa2", 12, 1024, 32), ("sun_sa2", 16, 1024, 32)):
    g = torch.Generator().manual_seed(1)
    y2 = (torch.randn(b, 128, m, ns, generator=g) * 1.3 + 0.2).to(dev)
    w3 = (torch.randn(256, 128, generator=g) / 11).to(dev)
    g2, be2 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
    g3, be3 = torch.rand(256, generator=g).to(dev) + 0.5, torch.randn(256, generator=g).to(dev) * 0.3
    z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    c2 = K.bn_coefficients(y2, g2, be2, *z(128), 0.1, 1e-5, True)
    fwd = lambda store: K.gemm_forward_bn(w3, y2, (c2[2], c2[3]), g3, be3, *z(256), 0.1, 1e-5, pool=True,
                                          store=store)
    y3, mean3, invstd3, sc3, sh3, ext = fwd(True)
    pooled, argmax, ymax = K.pool_from_extrema(ext, sc3, sh3)
    dpooled = torch.randn(b, 256, m, generator=g).to(dev)
    _, _, coef3 = K.bn_relu_pool_backward_stats(y3, dpooled, argmax, ymax, g3, sc3, sh3, mean3, invstd3, True)
    res = {}

    def stored_bwd():
        dx, dw, below = K.gemm_backward_fused(w3, y2, (c2[2], c2[3]),
                                              pooled=(y3, dpooled, argmax, sc3, sh3, mean3, invstd3, coef3),
                                              xstats=(c2[0], c2[1], g2, True))
        if below is None:
            below = K.bn_relu_backward_stats(y2, dx.view_as(y2), g2, c2[2], c2[3], c2[0], c2[1], True)
        return dx, dw, below

    for rep in range(2):
        res["fwd_stored_us_%d" % rep] = round(bench.time_op(lambda: fwd(True), iters=5, warm=2), 1)
        res["fwd_nostore_us_%d" % rep] = round(bench.time_op(lambda: fwd(False), iters=5, warm=2), 1)
        res["bwd_stored_us_%d" % rep] = round(bench.time_op(stored_bwd, iters=5, warm=2), 1)
        res["bwd_gram_us_%d" % rep] = round(bench.time_op(lambda: K.pool_gram_backward(
            w3, y2, c2, g2, coef3, (mean3, invstd3, sc3, sh3), dpooled, argmax, ymax, ns, True), iters=5, warm=2), 1)
    out[name] = res
    print(name, json.dumps(res), flush=True)
    del y2, y3
    torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
