"""Soak of csrc/mlp_pool_gram256.hip (and SA1's csrc/mlp_pool_gram.hip): random shapes / seeds, the Gram
backward against the stored-y3 one-pass backward; prints the worst relative errors and any failure.
    python tools/gram256_soak.py [rounds]"""
import importlib, os, sys
import torch
os.environ["MLP_POOL_GRAM256_MIN_CHUNKS"] = "64"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
worst = {"dx": 0.0, "dw": 0.0, "sums": 0.0}
bad = 0
g = torch.Generator().manual_seed(12345)
for it in range(rounds):
    kin, mout = ((128, 256) if it % 3 else (64, 128))
    ns = [16, 32][int(torch.randint(0, 2, (1,), generator=g))] if mout == 256 else [16, 32, 64][int(torch.randint(0, 3, (1,), generator=g))]
    b = int(torch.randint(1, 13, (1,), generator=g))
    m = int(torch.randint(1, 17, (1,), generator=g)) * (64 if ns == 16 else 32) * 2
    if b * m * ns // 32 < 64:
        continue
    y2 = (torch.randn(b, kin, m, ns, generator=g) * (0.5 + 2 * torch.rand(1, generator=g)) + torch.randn(1, generator=g)).to(dev)
    y2[:, :, :, ns // 2] = y2[:, :, :, 0]
    w3 = (torch.randn(mout, kin, generator=g) / kin ** 0.5).to(dev)
    def bn(c):
        gamma = torch.rand(c, generator=g) + 0.5
        gamma[::5] *= -1
        return gamma.to(dev), (torch.randn(c, generator=g) * 0.3).to(dev)
    g2, be2 = bn(kin); g3, be3 = bn(mout)
    z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    c2 = K.bn_coefficients(y2, g2, be2, *z(kin), 0.1, 1e-5, True)
    if not K.pool_gram_supported(w3, y2, ns):
        continue
    y3 = K.gemm_forward(w3, y2, (c2[2], c2[3]))
    mean3, invstd3, sc3, sh3 = K.bn_coefficients(y3, g3, be3, *z(mout), 0.1, 1e-5, True)
    pooled, argmax, ymax = K.bn_relu_pool(y3, sc3, sh3)
    dpooled = torch.randn(b, mout, m, generator=g).to(dev)
    _, _, coef3 = K.bn_relu_pool_backward_stats(y3, dpooled, argmax, ymax, g3, sc3, sh3, mean3, invstd3, True)
    want_dx = K.gemm_dgrad(w3, pooled=(y3, dpooled, argmax, sc3, sh3, mean3, invstd3, coef3))
    want_dw = K.gemm_wgrad(mout, kin, y2, (c2[2], c2[3]), pooled=(y3, dpooled, argmax, sc3, sh3, mean3, invstd3, coef3))
    dx, dw, below = K.pool_gram_backward(w3, y2, c2, g2, coef3, (mean3, invstd3, sc3, sh3), dpooled, argmax, ymax, ns, True)
    xh = (y2.double() - c2[0].double().view(1, -1, 1, 1)) * c2[1].double().view(1, -1, 1, 1)
    gate = (y2.double() * c2[2].double().view(1, -1, 1, 1) + c2[3].double().view(1, -1, 1, 1)) > 0
    gd = torch.where(gate, dx.double(), torch.zeros((), dtype=torch.float64, device=dev))
    e = {"dx": rel(dx, want_dx.view_as(dx)), "dw": rel(dw, want_dw),
         "sums": max(rel(below[0], (gd * xh).sum(dim=(0, 2, 3))), rel(below[1], gd.sum(dim=(0, 2, 3))))}
    ok = all(torch.isfinite(t).all() for t in (dx, dw, below[0], below[1])) and e["dx"] < 2e-5 and e["dw"] < 2e-4 and e["sums"] < 1e-5
    if not ok:
        bad += 1
        print("FAIL", (b, kin, mout, m, ns), e, flush=True)
    for k_ in worst:
        worst[k_] = max(worst[k_], e[k_] if e[k_] == e[k_] else 1e9)
print("rounds %d, failures %d, worst relative errors %s" % (rounds, bad, worst))
sys.exit(1 if bad else 0)
