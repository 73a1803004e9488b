"""conv1d(1) -> BN -> ReLU -> conv1d(1) -> BN -> ReLU -> conv1d(1): the three identical heads of
VoteNet-IoU (voting_module.py:34-60, proposal_module.py:78-125, grid_conv_module.py:60-116) as
ONE autograd node on the gfx950 MFMA / BatchNorm kernels of pointnet2._mlp_ext.

Same parameters (nn.Conv1d with bias, nn.BatchNorm1d) and the same results as
F.relu(bn(conv(x))) x2 + conv; what changes is how they are computed:
  * the 1x1 convolutions are the R-contiguous GEMMs of the shared MLP (no NCHW<->NHWC transposes,
    no MIOpen in the train step);
  * BatchNorm+ReLU of layer i are applied inside the operand loads of layer i+1, their backward
    inside the operand loads of dgrad / wgrad;
  * a bias in front of a training-mode BatchNorm cancels in the output; it only shifts the
    running mean (accounted for) and its gradient is sum(dy) == 0 (returned as exact zeros; the
    reference accumulates rounding noise there).  In eval mode the bias is folded into the shift.
"""
import os

import torch
from torch.autograd import Function


def _enabled():
    """On by default (VOTENET_FUSED_HEADS=0 restores the torch modules, i.e. MIOpen / rocBLAS):
    with the 64x64 small-problem GEMM kernel the step time is the same either way (12.6 ms), and
    this way no library kernel -- hence no solver search on a fresh machine -- is left in the step."""
    return os.environ.get("VOTENET_FUSED_HEADS", "1") != "0"


class _HeadChain(Function):
    """apply(x, training, mom1, eps1, mom2, eps2, tickets,
             w1, b1, g1, be1, rm1, rv1,  w2, b2, g2, be2, rm2, rv2,  w3, b3)
    tickets: the chain's counters for its one-launch reductions (_mlp_ext.tickets_of)"""

    @staticmethod
    def forward(ctx, x, training, mom1, eps1, mom2, eps2, tickets, w1, b1, g1, be1, rm1, rv1, w2, b2, g2,
                be2, rm2, rv2, w3, b3):
        from pointnet2 import _mlp_ext as K
        x = x.contiguous()
        w1m, w2m, w3m = (w.reshape(w.shape[0], -1) for w in (w1, w2, w3))

        def bn(y, bias, gamma, beta, rm, rv, momentum, eps):
            mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, rm, rv, momentum, eps,
                                                           training, tickets)
            if bias is not None:
                if training:
                    from pointnet2.pytorch_utils import deferred_axpy
                    deferred_axpy(rm, bias, momentum)  # running mean of (W x + b)
                else:
                    shift = shift + scale * bias.detach()    # scale*(y + b - rm) + beta
                    mean = mean - bias.detach()              # x-hat of the biased pre-activation
            return mean, invstd, scale, shift

        y1 = K.gemm_forward(w1m, x)
        c1 = bn(y1, b1, g1, be1, rm1, rv1, mom1, eps1)
        y2 = K.gemm_forward(w2m, y1, (c1[2], c1[3]))
        c2 = bn(y2, b2, g2, be2, rm2, rv2, mom2, eps2)
        y3 = K.gemm_forward(w3m, y2, (c2[2], c2[3]))
        if b3 is not None:
            y3 += b3.detach().view(1, -1, 1)
        ctx.save_for_backward(x, y1, y2, *c1, *c2, w1, w2, w3, g1, g2)
        ctx.training, ctx.tickets = training, tickets
        ctx.has_bias = (b1 is not None, b2 is not None, b3 is not None)
        return y3

    @staticmethod
    def backward(ctx, dy3):
        from pointnet2 import _mlp_ext as K
        (x, y1, y2, mean1, invstd1, scale1, shift1, mean2, invstd2, scale2, shift2, w1, w2, w3, g1,
         g2) = ctx.saved_tensors
        training = ctx.training
        dy3 = dy3.contiguous()
        w1m, w2m, w3m = (w.reshape(w.shape[0], -1) for w in (w1, w2, w3))
        db3 = dy3.sum(dim=(0, 2)) if ctx.has_bias[2] else None

        def both(wm, src, src_coeff, need_dx=True, **grad):
            """(dw, d src) of one layer: one launch for the two GEMMs where the layer is small"""
            pair = K.gemm_backward_small(wm, src, src_coeff, need_dx=need_dx, **grad)
            if pair is not None:
                return pair[1], pair[0]
            dwm = K.gemm_wgrad(wm.shape[0], wm.shape[1], src, src_coeff, **grad)
            return dwm, (K.gemm_dgrad(wm, **grad) if need_dx else None)

        def through_bn(wm, y, dz, gamma, scale, shift, mean, invstd):
            """BatchNorm + ReLU backward of (y, dz) as the gradient operand of the layer's GEMMs:
            (dgamma, dbeta, coef, operand keywords) -- dy written once where the pair launch would
            otherwise re-form it in every tile (_mlp_ext.small_backward_prefers_dy), else on the fly"""
            if K.small_backward_prefers_dy(wm, y):
                dy, dg, dbe = K.bn_relu_backward(y, dz, gamma, scale, shift, mean, invstd, training,
                                                 ctx.tickets)
                return dg, dbe, None, dict(dy=dy)
            dg, dbe, coef = K.bn_relu_backward_stats(y, dz, gamma, scale, shift, mean, invstd, training,
                                                     ctx.tickets)
            return dg, dbe, coef, dict(fly=(y, dz, scale, shift, mean, invstd, coef))

        dw3, dz2 = both(w3m, y2, (scale2, shift2), dy=dy3)
        dw3 = dw3.view_as(w3)
        dg2, dbe2, coef2, grad2 = through_bn(w2m, y2, dz2, g2, scale2, shift2, mean2, invstd2)
        dw2, dz1 = both(w2m, y1, (scale1, shift1), **grad2)
        dw2 = dw2.view_as(w2)
        dg1, dbe1, coef1, grad1 = through_bn(w1m, y1, dz1, g1, scale1, shift1, mean1, invstd1)
        dw1, dx = both(w1m, x, None, need_dx=ctx.needs_input_grad[0], **grad1)
        dw1 = dw1.view_as(w1)
        dx = dx.view_as(x) if dx is not None else None

        def dbias(present, coef, dbeta, gamma, invstd):
            if not present:
                return None
            # training: sum(dy) vanishes identically; eval: dy = gamma*invstd*dz_masked
            if training:
                from pointnet2 import pytorch_utils
                return None if pytorch_utils.zero_grads_as_none else torch.zeros_like(dbeta)
            if coef is None:  # (dy was materialised: no coefficient table)
                return gamma.detach() * invstd * dbeta
            return coef.reshape(-1)[0::3] * dbeta  # coef is [C][3]

        db2 = dbias(ctx.has_bias[1], coef2, dbe2, g2, invstd2)
        db1 = dbias(ctx.has_bias[0], coef1, dbe1, g1, invstd1)
        return (dx, None, None, None, None, None, None, dw1, db1, dg1, dbe1, None, None, dw2, db2, dg2,
                dbe2, None, None, dw3, db3)


def _plain_conv(conv):
    return (isinstance(conv, torch.nn.Conv1d) and conv.kernel_size == (1,) and conv.stride == (1,)
            and conv.padding == (0,) and conv.dilation == (1,) and conv.groups == 1)


def _plain_bn(bn):
    return (isinstance(bn, torch.nn.BatchNorm1d) and bn.affine and bn.track_running_stats
            and bn.momentum is not None)


def head_chain(x, conv1, bn1, conv2, bn2, conv3):
    """relu(bn1(conv1(x))) -> relu(bn2(conv2(.))) -> conv3(.) for x of shape (B, C, R)."""
    fused = (_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3
             and all(_plain_conv(c) for c in (conv1, conv2, conv3)) and _plain_bn(bn1)
             and _plain_bn(bn2) and bn1.training == bn2.training)
    if not fused:
        net = torch.nn.functional.relu(bn1(conv1(x)))
        net = torch.nn.functional.relu(bn2(conv2(net)))
        return conv3(net)
    training = bn1.training
    if training:
        from pointnet2.pytorch_utils import bump_batches_tracked
        bump_batches_tracked(bn1.num_batches_tracked)
        bump_batches_tracked(bn2.num_batches_tracked)
    from pointnet2 import _mlp_ext as K
    tickets = K.tickets_of(bn1, max(conv1.out_channels, conv2.out_channels), x.device)
    return _HeadChain.apply(x, training, bn1.momentum, bn1.eps, bn2.momentum, bn2.eps, tickets,
                            conv1.weight, conv1.bias, bn1.weight, bn1.bias, bn1.running_mean,
                            bn1.running_var, conv2.weight, conv2.bias, bn2.weight, bn2.bias,
                            bn2.running_mean, bn2.running_var, conv3.weight, conv3.bias)
