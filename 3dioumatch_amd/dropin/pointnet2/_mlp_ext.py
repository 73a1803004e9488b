"""pointnet2._mlp_ext -- ctypes binding of the fused BatchNorm(+ReLU)(+max-pool) kernels
(include/mlp_hip.h) used by pointnet2.pytorch_utils.SharedMLP on the GPU.

No reference counterpart: the reference runs nn.BatchNorm2d / nn.ReLU / F.max_pool2d here
(pointnet2/pytorch_utils.py:14-124, pointnet2/pointnet2_modules.py:256-262).
"""
import torch

from pointnet2._ext import _L, _lib, _stream


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise RuntimeError("%s must be a contiguous float32 GPU tensor" % name)


def _ws(y, b, c, r):
    n = int(_lib.mlp_bn_workspace_floats(b, c, r))
    return torch.empty(max(n, 1), dtype=torch.float32, device=y.device)


def bn_coefficients(y, gamma, beta, running_mean, running_var, momentum, eps, training):
    """y (B,C,...) -> per-channel (mean, invstd, scale, shift); training=True uses (and folds
    into running_*) the batch statistics, otherwise the running statistics."""
    _f32c(y, "y")
    b, c = y.shape[0], y.shape[1]
    r = y.numel() // (b * c)
    out = torch.empty((4, c), dtype=torch.float32, device=y.device)
    mean, invstd, scale, shift = out[0], out[1], out[2], out[3]
    with torch.cuda.device(y.device):
        if training:
            ws = _ws(y, b, c, r)
            rm = running_mean.data_ptr() if running_mean is not None else None
            rv = running_var.data_ptr() if running_var is not None else None
            _L.check(_lib.mlp_bn_train_stats(b, c, r, y.data_ptr(), gamma.data_ptr(),
                                             beta.data_ptr(), float(eps), float(momentum), rm, rv,
                                             mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                             shift.data_ptr(), ws.data_ptr(), _stream(y)),
                     "mlp_bn_train_stats")
        else:
            _L.check(_lib.mlp_bn_eval_coeff(c, gamma.data_ptr(), beta.data_ptr(), float(eps),
                                            running_mean.data_ptr(), running_var.data_ptr(),
                                            mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                                            shift.data_ptr(), _stream(y)), "mlp_bn_eval_coeff")
    return mean, invstd, scale, shift


def bn_relu_apply(y, scale, shift):
    b, c = y.shape[0], y.shape[1]
    r = y.numel() // (b * c)
    z = torch.empty_like(y)
    with torch.cuda.device(y.device):
        _L.check(_lib.mlp_bn_relu_apply(b, c, r, y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                        z.data_ptr(), _stream(y)), "mlp_bn_relu_apply")
    return z


def bn_relu_pool(y, scale, shift):
    """y (B,C,m,ns) -> pooled (B,C,m), argmax int32 (B,C,m), ymax (B,C,m)."""
    b, c, m, ns = y.shape
    pooled = torch.empty((b, c, m), dtype=torch.float32, device=y.device)
    ymax = torch.empty_like(pooled)
    argmax = torch.empty((b, c, m), dtype=torch.int32, device=y.device)
    with torch.cuda.device(y.device):
        _L.check(_lib.mlp_bn_relu_pool(b, c, m, ns, y.data_ptr(), scale.data_ptr(),
                                       shift.data_ptr(), pooled.data_ptr(), argmax.data_ptr(),
                                       ymax.data_ptr(), _stream(y)), "mlp_bn_relu_pool")
    return pooled, argmax, ymax


def bn_relu_backward(y, dz, gamma, scale, shift, mean, invstd, training):
    _f32c(dz, "dz")
    b, c = y.shape[0], y.shape[1]
    r = y.numel() // (b * c)
    dy = torch.empty_like(y)
    small = torch.empty((5, c), dtype=torch.float32, device=y.device)  # dgamma, dbeta, coef[3]
    with torch.cuda.device(y.device):
        ws = _ws(y, b, c, r)
        _L.check(_lib.mlp_bn_relu_backward(b, c, r, 1 if training else 0, y.data_ptr(),
                                           dz.data_ptr(), gamma.data_ptr(), scale.data_ptr(),
                                           shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                           dy.data_ptr(), small[0].data_ptr(), small[1].data_ptr(),
                                           small[2:].data_ptr(), ws.data_ptr(), _stream(y)),
                 "mlp_bn_relu_backward")
    return dy, small[0], small[1]


def bn_relu_pool_backward(y, dpooled, argmax, ymax, gamma, scale, shift, mean, invstd, training):
    _f32c(dpooled, "dpooled")
    b, c, m, ns = y.shape
    dy = torch.empty_like(y)
    small = torch.empty((5, c), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        ws = _ws(y, b, c, m)
        _L.check(_lib.mlp_bn_relu_pool_backward(b, c, m, ns, 1 if training else 0, y.data_ptr(),
                                                dpooled.data_ptr(), argmax.data_ptr(),
                                                ymax.data_ptr(), gamma.data_ptr(),
                                                scale.data_ptr(), shift.data_ptr(),
                                                mean.data_ptr(), invstd.data_ptr(), dy.data_ptr(),
                                                small[0].data_ptr(), small[1].data_ptr(),
                                                small[2:].data_ptr(), ws.data_ptr(), _stream(y)),
                 "mlp_bn_relu_pool_backward")
    return dy, small[0], small[1]
