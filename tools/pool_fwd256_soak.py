"""Random shapes of the pooled 128 -> 256 forward (csrc/mlp_pool_fwd256.hip, with and without its store)
against float64 torch: N rounds, prints the worst errors.    python tools/pool_fwd256_soak.py [rounds]"""
import importlib, os, random, sys
import torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(7)
worst = {"y": 0.0, "ext": 0.0, "mean": 0.0, "invstd": 0.0, "picked": 0.0}
fails = 0
for it in range(rounds):
    ns = rnd.choice((16, 32))
    b = rnd.randint(1, 9)
    m = rnd.choice((8, 16, 24, 40, 64, 100, 128, 200, 256, 512, 1000, 1024)) * (32 // ns if ns == 16 else 1)
    if (m * ns) % 256 != 0 or b * m * ns <= 16384:
        continue
    g = torch.Generator().manual_seed(1000 + it)
    y2 = (torch.randn(b, 128, m, ns, generator=g) * rnd.uniform(0.3, 3.0) + rnd.uniform(-1, 1)).to(dev)
    y2[:, :, :, 2] = y2[:, :, :, 0]
    w3 = (torch.randn(256, 128, generator=g) / 11).to(dev)
    g2, be2 = (torch.rand(128, generator=g) + 0.5).to(dev), (torch.randn(128, generator=g) * 0.3).to(dev)
    g3 = torch.rand(256, generator=g) + 0.5
    g3[::3] *= -1
    g3 = g3.to(dev)
    be3 = (torch.randn(256, generator=g) * 0.3).to(dev)
    z = lambda c: (torch.zeros(c, device=dev), torch.ones(c, device=dev))
    c2 = K.bn_coefficients(y2, g2, be2, *z(128), 0.1, 1e-5, True)
    if not K.forward_pool_supported(w3, y2, (c2[2], c2[3])):
        continue
    a2 = torch.relu(y2.double() * c2[2].double().view(1, -1, 1, 1) + c2[3].double().view(1, -1, 1, 1))
    y64 = torch.einsum("ck,bkmn->bcmn", w3.double(), a2)
    rng = float(y64.abs().max())
    mean64, var64 = y64.mean(dim=(0, 2, 3)), y64.var(dim=(0, 2, 3), unbiased=False)
    sign = torch.where(g3 < 0, -1.0, 1.0).double().view(1, -1, 1)
    best64 = (y64 * sign.unsqueeze(-1)).max(dim=3).values * sign
    for store in (False, True):
        y3, mean, invstd, sc, sh, ext = K.gemm_forward_bn(w3, y2, (c2[2], c2[3]), g3, be3, *z(256), 0.1, 1e-5,
                                                          pool=True, store=store)
        idx = ext[1].view(torch.int32).long()
        ok = int(idx.min()) >= 0 and int(idx.max()) < ns and not bool((idx == 2).any())
        picked = torch.gather(y64, 3, idx.clamp(0, ns - 1).unsqueeze(-1)).squeeze(-1)
        errs = {"ext": float((ext[0].double() - best64).abs().max()) / rng,
                "picked": float((picked - best64).abs().max()) / rng,
                "mean": float((mean.double() - mean64).abs().max()) / rng,
                "invstd": float((invstd.double() * (var64 + 1e-5).sqrt() - 1).abs().max())}
        if store:
            errs["y"] = float((y3.double() - y64).abs().max()) / rng
        for k, v in errs.items():
            worst[k] = max(worst[k], v)
        if not ok or errs["ext"] > 2e-6 or errs["picked"] > 2e-6 or errs["mean"] > 1e-6 or errs["invstd"] > 1e-5 or \
                errs.get("y", 0.0) > 2e-6 or not all(v == v for v in errs.values()):
            fails += 1
            print("FAIL b %d m %d ns %d store %d: %s ok=%s" % (b, m, ns, store, errs, ok), flush=True)
    del y2, a2, y64
print("rounds %d, failures %d, worst relative errors %s" % (rounds, fails, worst))
sys.exit(1 if fails else 0)
