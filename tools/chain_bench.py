"""SA1's shared MLP forward (4 -> 64 -> 64 -> 128, B = 8, m = 2048, ns = 64) two ways: the
register-chained kernels (csrc/mlp_chain.hip: statistics pass + full pass) against the layer-by-layer
kernels of rounds 1-4 (lin4 GEMM with statistics, pooled GEMM with statistics / extrema epilogue).
Device time per forward from HIP-graph replays.      python tools/chain_bench.py [out.json]"""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
dev = torch.device("cuda:0")
b, m, ns = 8, 2048, 64
g = torch.Generator().manual_seed(1)
x = (torch.randn(b, 4, m, ns, generator=g)).to(dev)
w0 = (torch.randn(64, 4, generator=g) * 0.7).to(dev)
w1 = (torch.randn(64, 64, generator=g) / 8).to(dev)
w2 = (torch.randn(128, 64, generator=g) / 8).to(dev)
bn = lambda c: [torch.rand(c, generator=g).to(dev) + 0.5, torch.randn(c, generator=g).to(dev) * 0.3,
                torch.zeros(c, device=dev), torch.ones(c, device=dev)]
p0, p1, p2 = bn(64), bn(64), bn(128)
mom = K.first4_moments(x)
c0 = K.first4_bn(mom, x.numel() // 4, w0, p0[0], p0[1], p0[2], p0[3], 0.1, 1e-5)

def chained(store=True):
    return K.chain_lin4_forward(x, w0, (c0[2], c0[3]), (w1, *p1, 0.1, 1e-5), (w2, *p2, 0.1, 1e-5), store=store)

def layerwise():
    y1, *c1 = K.gemm_forward_bn_lin4(w1, x, w0, (c0[2], c0[3]), p1[0], p1[1], p1[2], p1[3], 0.1, 1e-5)
    return K.gemm_forward_bn(w2, y1, (c1[2], c1[3]), p2[0], p2[1], p2[2], p2[3], 0.1, 1e-5, pool=True)

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters): fn()
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters

out = {"shape": "B=8 m=2048 ns=64: 4 -> 64 -> 64 -> 128, 1 048 576 columns"}
for rep in range(2):
    out["layerwise_us_%d" % rep] = round(timeit(layerwise), 1)
    out["chained_us_%d" % rep] = round(timeit(chained), 1)
    out["chained_no_store_us_%d" % rep] = round(timeit(lambda: chained(False)), 1)
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
