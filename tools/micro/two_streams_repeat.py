import importlib, sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_pkg
load_pkg()
K = importlib.import_module("pointnet2._mlp_ext")
DEV = "cuda:0"
g = torch.Generator().manual_seed(5)
cases = []
for c, r in ((64, 40000), (128, 9000)):
    y = (torch.randn(4, c, r, generator=g) * 1.5 + 0.3).to(DEV)
    dz = torch.randn(4, c, r, generator=g).to(DEV)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(c, generator=g) * 0.3).to(DEV)
    cases.append((y, dz, gamma, beta))
def work(case):
    y, dz, gamma, beta = case
    rm, rv = torch.zeros_like(gamma), torch.ones_like(gamma)
    mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, rm, rv, 0.1, 1e-5, True)
    dgamma, dbeta, coef = K.bn_relu_backward_stats(y, dz, gamma, scale, shift, mean, invstd, True)
    return [mean, invstd, dgamma, dbeta, coef]
names = ["mean", "invstd", "dgamma", "dbeta", "coef"]
alone = [[t.clone() for t in work(cs)] for cs in cases]
torch.cuda.synchronize()
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    for _ in range(40):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                got[si].append(work(cases[si]))
    torch.cuda.synchronize()
    for si in range(2):
        for it, res in enumerate(got[si]):
            for nm, a, b in zip(names, res, alone[si]):
                if not torch.equal(a, b):
                    d = (a - b).abs()
                    bad += 1
                    if bad <= 12:
                        print("rep %d stream %d iter %d %s: %d of %d differ, max %.3e (ref max %.3e), first idx %d" % (
                            rep, si, it, nm, int((d > 0).sum()), d.numel(), float(d.max()), float(b.abs().max()), int((d.reshape(-1) > 0).nonzero()[0])))
print("mismatching tensors:", bad)
