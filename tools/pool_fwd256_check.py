"""The pooled 128 -> 256 forward that stores nothing (csrc/mlp_pool_fwd256.hip) against the tiled kernel
that stores y3 and against float64 torch; us per call of both.   python tools/pool_fwd256_check.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
for b, m, ns in ((8, 1024, 32), (2, 512, 16), (3, 200, 32), (12, 1024, 32), (8, 512, 16), (8, 256, 16)):
    g = torch.Generator().manual_seed(b + m + ns)
    y2 = (torch.randn(b, 128, m, ns, generator=g) * 1.3 + 0.2).to(dev)
    y2[:, :, :, 3] = y2[:, :, :, 1]
    w3 = (torch.randn(256, 128, generator=g) / 11).to(dev)
    g2, be2 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
    g3 = torch.rand(256, generator=g) + 0.5
    g3[::5] *= -1
    g3 = g3.to(dev)
    be3 = torch.randn(256, generator=g).to(dev) * 0.3
    z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    c2 = K.bn_coefficients(y2, g2, be2, *z(128), 0.1, 1e-5, True)
    fwd = lambda store: K.gemm_forward_bn(w3, y2, (c2[2], c2[3]), g3, be3, *z(256), 0.1, 1e-5, pool=True, store=store)
    os.environ["MLP_POOL_FWD256"] = "0"   # the tiled kernel
    y3, mean, invstd, sc, sh, ext = fwd(True)
    os.environ.pop("MLP_POOL_FWD256")
    if ext is None:
        print("b %d m %d ns %d: no pooled epilogue at this size" % (b, m, ns))
        continue
    _, mean_n, invstd_n, sc_n, sh_n, ext_n = fwd(False)
    torch.cuda.synchronize()
    a2 = torch.relu(y2.double() * c2[2].double().view(1, -1, 1, 1) + c2[3].double().view(1, -1, 1, 1))
    y64 = torch.einsum("ck,bkmn->bcmn", w3.double(), a2)
    rng = float(y64.abs().max())
    mean64 = y64.mean(dim=(0, 2, 3)); var64 = y64.var(dim=(0, 2, 3), unbiased=False)
    sel = torch.where(g3.view(1, -1, 1, 1) < 0, -y64, y64)
    best64 = sel.max(dim=3).values * torch.where(g3 < 0, -1.0, 1.0).view(1, -1, 1)
    e = lambda a_, b_: float((a_.double() - b_.double()).abs().max())
    idx_n = ext_n[1].view(torch.int32).long()
    picked = torch.gather(y64, 3, idx_n.unsqueeze(-1)).squeeze(-1)
    first = (torch.gather(y3, 3, idx_n.unsqueeze(-1)).squeeze(-1))
    print("b %d m %d ns %d | vs f64: ext %.2e (tiled %.2e) picked %.2e mean %.2e (tiled %.2e) invstd rel %.2e (tiled %.2e) | "
          "argmax differs from tiled on %.4f%%" % (
              b, m, ns, e(ext_n[0], best64) / rng, e(ext[0], best64) / rng, e(picked, best64) / rng,
              e(mean_n, mean64) / rng, e(mean, mean64) / rng,
              float(((invstd_n.double() - (var64 + 1e-5).rsqrt()) * (var64 + 1e-5).sqrt()).abs().max()),
              float(((invstd.double() - (var64 + 1e-5).rsqrt()) * (var64 + 1e-5).sqrt()).abs().max()),
              100 * float((ext_n[1].view(torch.int32) != ext[1].view(torch.int32)).float().mean())), flush=True)
    # exact ties (columns 1 and 3 are equal): never the later one
    assert not bool((idx_n == 3).any())
    y3n = fwd(True)[0]
    print("   stored y3 vs f64: %.2e (tiled %.2e)" % (e(y3n, y64) / rng, e(y3, y64) / rng))
    os.environ["MLP_POOL_FWD256"] = "0"
    t_old = bench.time_op(lambda: fwd(True), iters=5, warm=2)
    t_old_n = bench.time_op(lambda: fwd(False), iters=5, warm=2)
    os.environ.pop("MLP_POOL_FWD256")
    t_new_s = bench.time_op(lambda: fwd(True), iters=5, warm=2)
    t_new = bench.time_op(lambda: fwd(False), iters=5, warm=2)
    print("   us per call (GEMM + finalize): tiled stored %.1f / no store %.1f; T form stored %.1f / no store %.1f" % (
        t_old, t_old_n, t_new_s, t_new), flush=True)
    del y2, y3, a2, y64, sel
    torch.cuda.empty_cache()
