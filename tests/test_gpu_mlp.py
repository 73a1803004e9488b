"""GPU tests of the fused BatchNorm(+ReLU)(+max-pool) kernels behind SharedMLP against plain
torch ops on the same device (fp32 reference of the same op): forward values, running
statistics, and gradients w.r.t. input, gamma, beta -- training and eval mode, vector and
ragged shapes.  Tolerance 1e-4 (relative to the tensor scale)."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_pkg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _mods():
    load_pkg()
    return importlib.import_module("pointnet2.pytorch_utils")


def close(a, b, tol=1e-4):
    a, b = a.detach().float().cpu().numpy(), b.detach().float().cpu().numpy()
    scale = max(1.0, float(np.abs(b).max()))
    assert np.abs(a - b).max() <= tol * scale, float(np.abs(a - b).max())


@pytest.mark.parametrize("shape", [(2, 7, 33, 5), (2, 16, 40, 16), (3, 64, 100, 64),
                                   (2, 8, 50, 32), (2, 5, 1000, 1), (1, 3, 7, 8), (4, 128, 64, 4)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("pool", [False, True])
def test_bn_relu_pool_vs_torch(shape, training, pool):
    P = _mods()
    g = torch.Generator().manual_seed(sum(shape))
    b, c, m, ns = shape
    y0 = (torch.randn(shape, generator=g) * 2 + 0.5).to(DEV)
    gamma0 = (torch.rand(c, generator=g) + 0.5).to(DEV)
    gamma0[0] = -0.7  # negative scale: max and relu must not be commuted blindly
    beta0 = (torch.randn(c, generator=g) * 0.3).to(DEV)
    rm0 = (torch.randn(c, generator=g) * 0.1).to(DEV)
    rv0 = (torch.rand(c, generator=g) + 0.5).to(DEV)
    w = torch.randn((b, c, m) if pool else shape, generator=g).to(DEV)

    def run(fused):
        y = y0.clone().requires_grad_(True)
        gamma = gamma0.clone().requires_grad_(True)
        beta = beta0.clone().requires_grad_(True)
        rm, rv = rm0.clone(), rv0.clone()
        if fused:
            op = P._BNReLUMaxPool if pool else P._BNReLU
            out = op.apply(y, gamma, beta, rm, rv, 0.1, 1e-5, training)
        else:
            z = F.relu(F.batch_norm(y, rm, rv, gamma, beta, training, 0.1, 1e-5))
            out = torch.max(z, dim=3)[0] if pool else z
        (out * w).sum().backward()
        return out, y.grad, gamma.grad, beta.grad, rm, rv

    got, want = run(True), run(False)
    names = ["out", "dy", "dgamma", "dbeta", "running_mean", "running_var"]
    for n, a, bb in zip(names, got, want):
        tol = 1e-4 if n in ("out", "running_mean", "running_var") else 3e-4
        close(a, bb, tol)


def test_shared_mlp_fused_matches_sequential():
    """SharedMLP on the GPU (fused path) == the same module evaluated layer by layer with torch
    ops (its nn.Sequential definition), forward + backward, incl. forward_pooled."""
    P = _mods()
    torch.manual_seed(0)
    mlp = P.SharedMLP([6, 16, 16, 32], bn=True).to(DEV).train()
    ref = P.SharedMLP([6, 16, 16, 32], bn=True).to(DEV).train()
    ref.load_state_dict(mlp.state_dict())
    x1 = torch.randn(3, 6, 50, 16, device=DEV, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    out = mlp.forward_pooled(x1)
    z = torch.nn.Sequential.forward(ref, x2)
    want = torch.max(z, dim=3)[0]
    close(out, want)
    wgt = torch.randn_like(want)
    (out * wgt).sum().backward()
    (want * wgt).sum().backward()
    close(x1.grad, x2.grad, 3e-4)
    for (n1, p1), (n2, p2) in zip(mlp.named_parameters(), ref.named_parameters()):
        close(p1.grad, p2.grad, 3e-4)
    for (n1, b1), (n2, b2) in zip(mlp.named_buffers(), ref.named_buffers()):
        close(b1.float(), b2.float())
    mlp.eval(); ref.eval()
    with torch.no_grad():
        close(mlp(x1), torch.nn.Sequential.forward(ref, x2))
