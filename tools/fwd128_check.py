"""The (128, 128) layers' forward with statistics as the persistent T-form kernel (csrc/mlp_pool_fwd256.hip:
fwd128_kernel) against the tiled kernels (MLP_FWD128=0) and float64 torch; us per call of both.
    python tools/fwd128_check.py"""
import importlib, os, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
import bench
dev = torch.device("cuda:0")
e = lambda a_, b_: float((a_.double() - b_.double()).abs().max())
for b, m, ns, pool in ((8, 1024, 32, False), (8, 512, 64, True), (8, 512, 16, False), (8, 256, 16, True),
                       (3, 200, 32, True), (12, 1024, 32, False), (8, 256, 16, False)):
    g = torch.Generator().manual_seed(b + m + ns)
    x = (torch.randn(b, 128, m, ns, generator=g) * 1.3 + 0.2).to(dev)
    x[:, :, :, 3] = x[:, :, :, 1]
    w = (torch.randn(128, 128, generator=g) / 11).to(dev)
    g2, be2 = torch.rand(128, generator=g).to(dev) + 0.5, torch.randn(128, generator=g).to(dev) * 0.3
    g3 = torch.rand(128, generator=g) + 0.5
    g3[::5] *= -1
    g3 = g3.to(dev)
    be3 = torch.randn(128, generator=g).to(dev) * 0.3
    z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    c2 = K.bn_coefficients(x, g2, be2, *z(128), 0.1, 1e-5, True)
    fwd = lambda: K.gemm_forward_bn(w, x, (c2[2], c2[3]), g3, be3, *z(128), 0.1, 1e-5, pool=pool)
    os.environ.pop("MLP_FWD128", None)
    old = fwd()
    t_old = bench.time_op(fwd, iters=5, warm=2)
    os.environ["MLP_FWD128"] = "1"
    new = fwd()
    t_new = bench.time_op(fwd, iters=5, warm=2)
    os.environ.pop("MLP_FWD128")
    torch.cuda.synchronize()
    a2 = torch.relu(x.double() * c2[2].double().view(1, -1, 1, 1) + c2[3].double().view(1, -1, 1, 1))
    y64 = torch.einsum("ck,bkmn->bcmn", w.double(), a2)
    rng = float(y64.abs().max())
    mean64, var64 = y64.mean(dim=(0, 2, 3)), y64.var(dim=(0, 2, 3), unbiased=False)
    msg = "b %d m %d ns %d pool %d | y vs f64 %.2e (tiled %.2e) mean %.2e (%.2e) invstd rel %.2e (%.2e)" % (
        b, m, ns, pool, e(new[0], y64) / rng, e(old[0], y64) / rng, e(new[1], mean64) / rng, e(old[1], mean64) / rng,
        float((new[2].double() * (var64 + 1e-5).sqrt() - 1).abs().max()),
        float((old[2].double() * (var64 + 1e-5).sqrt() - 1).abs().max()))
    if pool and new[5] is not None:
        sign = torch.where(g3 < 0, -1.0, 1.0).double().view(1, -1, 1)
        best64 = (y64 * sign.unsqueeze(-1)).max(dim=3).values * sign
        idx = new[5][1].view(torch.int32).long()
        picked = torch.gather(y64, 3, idx.clamp(0, ns - 1).unsqueeze(-1)).squeeze(-1)
        msg += " | ext %.2e (%.2e) picked %.2e idx range %d..%d differs from tiled %.4f%% idx==3: %d" % (
            e(new[5][0], best64) / rng, e(old[5][0], best64) / rng, e(picked, best64) / rng, int(idx.min()), int(idx.max()),
            100 * float((new[5][1].view(torch.int32) != old[5][1].view(torch.int32)).float().mean()), int((idx == 3).sum()))
    print(msg, flush=True)
    print("   us per call (GEMM + finalize): tiled %.1f  T form %.1f" % (t_old, t_new), flush=True)
    del x, a2, y64, old, new
    torch.cuda.empty_cache()
