"""pointnet2.pytorch_utils -- shared-MLP building blocks (host-side mirror).

Public names, constructor arguments and the MODULE TREE (hence state_dict keys such as
`layer0.conv.weight`, `layer0.bn.bn.running_mean`) follow the reference
pointnet2/pytorch_utils.py:14-39 (SharedMLP), :42-67 (BatchNorm wrappers), :70-124
(_ConvBase), :127-236 (Conv1d/2d/3d), :239-270 (FC), :272-299 (BN momentum scheduler), so
checkpoints are interchangeable.  The implementation is written fresh around one builder.
"""
import contextlib
import os

import torch
import torch.nn as nn
from torch.autograd import Function


class _BNBase(nn.Sequential):
    """A batch-norm layer wrapped in a Sequential under the child name `<name>bn`
    (weight 1, bias 0), as the reference does at pytorch_utils.py:42-50."""

    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        layer = batch_norm(in_size)
        nn.init.constant_(layer.weight, 1.0)
        nn.init.constant_(layer.bias, 0)
        self.add_module(name + "bn", layer)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm1d, name=name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm2d, name=name)


class BatchNorm3d(_BNBase):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm3d, name=name)


def _assemble(seq, name, core_name, core, bn_unit, activation, preact):
    """Order the (bn, activation, core) children: pre-activation puts bn/act first."""
    tail = []
    if bn_unit is not None:
        tail.append((name + "bn", bn_unit))
    if activation is not None:
        tail.append((name + "activation", activation))
    parts = tail + [(name + core_name, core)] if preact else [(name + core_name, core)] + tail
    for child_name, child in parts:
        seq.add_module(child_name, child)


class _ConvBase(nn.Sequential):
    """conv (bias only without bn) [+ bn] [+ activation]; kaiming-normal weights by default."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                 conv=None, batch_norm=None, bias=True, preact=False, name=""):
        super().__init__()
        use_bias = bias and (not bn)
        conv_unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride,
                         padding=padding, bias=use_bias)
        init(conv_unit.weight)
        if use_bias:
            nn.init.constant_(conv_unit.bias, 0)
        bn_unit = batch_norm(in_size if preact else out_size) if bn else None
        _assemble(self, name, "conv", conv_unit, bn_unit, activation, preact)


def _conv_class(conv, batch_norm, ones):
    class _Conv(_ConvBase):
        def __init__(self, in_size, out_size, *, kernel_size=ones, stride=ones,
                     padding=tuple(0 for _ in ones) if isinstance(ones, tuple) else 0,
                     activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_,
                     bias=True, preact=False, name=""):
            super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn,
                             init, conv=conv, batch_norm=batch_norm, bias=bias, preact=preact,
                             name=name)
    return _Conv


Conv1d = _conv_class(nn.Conv1d, BatchNorm1d, 1)
Conv1d.__name__ = Conv1d.__qualname__ = "Conv1d"
Conv2d = _conv_class(nn.Conv2d, BatchNorm2d, (1, 1))
Conv2d.__name__ = Conv2d.__qualname__ = "Conv2d"
Conv3d = _conv_class(nn.Conv3d, BatchNorm3d, (1, 1, 1))
Conv3d.__name__ = Conv3d.__qualname__ = "Conv3d"


class _BNReLU(Function):
    """z = relu(batch_norm(y)) with the fused gfx950 kernels (pointnet2._mlp_ext).  Saves y and
    four per-channel vectors; the ReLU mask and x-hat are recomputed in the backward."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, momentum, eps, training, tickets=None):
        from pointnet2 import _mlp_ext as K
        y = y.contiguous()
        mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, running_mean, running_var,
                                                       momentum, eps, training, tickets)
        ctx.save_for_backward(y, gamma, scale, shift, mean, invstd)
        ctx.training, ctx.tickets = training, tickets
        return K.bn_relu_apply(y, scale, shift)

    @staticmethod
    def backward(ctx, dz):
        from pointnet2 import _mlp_ext as K
        y, gamma, scale, shift, mean, invstd = ctx.saved_tensors
        dy, dgamma, dbeta = K.bn_relu_backward(y, dz.contiguous(), gamma, scale, shift, mean,
                                               invstd, ctx.training, ctx.tickets)
        return dy, dgamma, dbeta, None, None, None, None, None, None


class _BNReLUMaxPool(Function):
    """(B,C,m,ns) -> (B,C,m): max over nsample of relu(batch_norm(y)) in one pass."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, momentum, eps, training, tickets=None):
        from pointnet2 import _mlp_ext as K
        y = y.contiguous()
        mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, running_mean, running_var,
                                                       momentum, eps, training, tickets)
        pooled, argmax, ymax = K.bn_relu_pool(y, scale, shift)
        ctx.save_for_backward(y, gamma, scale, shift, mean, invstd, argmax, ymax)
        ctx.training, ctx.tickets = training, tickets
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        from pointnet2 import _mlp_ext as K
        y, gamma, scale, shift, mean, invstd, argmax, ymax = ctx.saved_tensors
        dy, dgamma, dbeta = K.bn_relu_pool_backward(y, dpooled.contiguous(), argmax, ymax, gamma,
                                                    scale, shift, mean, invstd, ctx.training, ctx.tickets)
        return dy, dgamma, dbeta, None, None, None, None, None, None


class _FusedMLPChain(Function):
    """The whole conv(1x1)+BN+ReLU stack (and optionally the final max over nsample) as ONE
    autograd node on the gfx950 kernels: MFMA GEMMs whose operand loads apply the previous
    layer's BatchNorm+ReLU (forward) or form the BatchNorm/ReLU backward of the incoming
    gradient (backward) on the fly.  Per layer only the raw GEMM output y_i is kept; no
    normalised / rectified activation and no mask is ever written to memory.

    apply(x, pool, training, momenta, epss, pre, tickets, w_0, g_0, b_0, rm_0, rv_0, w_1, ...)

    tickets = the module's counters for the one-launch reductions (_mlp_ext.tickets_of; None: a
    fresh zeroed array per reduction).

    pre = None, or (idx (B,m,ns) int32, inverse (B,entries) int32, n): the first layer is applied
    BEFORE the gather (csrc/mlp_pregather.hip) -- x is then the packed point-major operand
    src_ext (B, 3+C, n+m) of _mlp_ext.pregather_pack instead of the grouped tensor (B, 3+C, m, ns),
    which is never formed: y_0 = (W_0 . src_ext)[.., idx] - (W_0 . src_ext)[.., n + j]."""

    @staticmethod
    def forward(ctx, x, pool, training, momenta, epss, pre, tickets, *params):
        from pointnet2 import _mlp_ext as K
        n_layers = len(params) // 5
        ctx.tickets = tickets
        x = x.contiguous()
        ys, coefs = [], []
        cur, cur_coeff = x, None
        # A 4 -> 64 first layer followed by a 64 -> 64 layer (SA1) stays VIRTUAL: its output is a
        # rank-4 function of x, so its BatchNorm statistics follow from the second moments of x
        # and every kernel that needs a row of it recomputes that row (four FMAs per element)
        # instead of a 268 MB tensor being written once and read three times.
        moments = None
        if pre is not None and (n_layers < 2 or x.dim() != 3):
            raise RuntimeError("the pre-gather form needs src_ext (B, 3+C, n+m) and two layers or more")
        virtual0 = (training and n_layers >= 3 and x.dim() == 4 and not ctx.needs_input_grad[0]
                    and K.lin4_supported(params[0].reshape(params[0].shape[0], -1),
                                         params[5].reshape(params[5].shape[0], -1), x))
        # ... and layers 2 + 3 of that module (64 -> 64 -> 128, max-pooled) run as ONE register-chained
        # kernel (csrc/mlp_chain.hip): layer 2's activation never leaves the registers, statistics
        # and pooled extrema are in-lane reductions of the second GEMM's accumulators
        chained = (virtual0 and pool and n_layers == 3 and
                   K.chain_lin4_supported(params[0].reshape(params[0].shape[0], -1),
                                          params[5].reshape(params[5].shape[0], -1),
                                          params[10].reshape(params[10].shape[0], -1), x, x.shape[3]))
        ext = None
        gram_last, gram_ns = False, 0
        for i in range(n_layers):
            w, gamma, beta, rm, rv = params[5 * i:5 * i + 5]
            w2 = w.reshape(w.shape[0], -1)
            if chained and i >= 1:
                if i == 2:
                    continue
                w0 = params[0].reshape(params[0].shape[0], -1)
                lay = [(params[5 * q].reshape(params[5 * q].shape[0], -1),) + tuple(params[5 * q + 1:5 * q + 5])
                       + (momenta[q], epss[q]) for q in (1, 2)]
                # the LAST layer's raw output is not even stored when its backward can run from the
                # Gram matrix of its input (csrc/mlp_pool_gram.hip): 537 MB at SA1
                gram = K.pool_gram_supported(lay[1][0], x, x.shape[3])
                # (a pass that no backward follows -- the EMA teacher -- stores neither raw output:
                # layer 2's activation goes from one GEMM to the next in registers anyway)
                y1, c1, y2, c2, ext = K.chain_lin4_forward(x, w0, cur_coeff, lay[0], lay[1],
                                                           store=any(ctx.needs_input_grad), store_last=not gram)
                if y1 is None:
                    y1 = x.new_empty(0)  # never materialised
                if y2 is None:
                    y2 = x.new_empty(0)  # never materialised
                ys += [y1, y2]
                coefs += [c1, c2]
                cur, cur_coeff = y2, (c2[2], c2[3])
                gram_last = gram
                continue
            if pre is not None and isinstance(pre[0], str) and i == 0:
                # ("interp", idx, weight, rel, shape): the layer's input is cat([rel (3 rows),
                # three_interpolate(x)]) over n >> m queries and the convolution commutes with the
                # interpolation: GEMM over the m source points, then ONE kernel interpolates its output
                # and adds the coordinate rows' part (no-grad passes only: the weight gradient would
                # need the interpolated input after all)
                from pointnet2 import _ext
                _, q_idx, q_weight, rel, shape = pre
                z = K.gemm_forward(w2[:, 3:].contiguous(), x)
                y = _ext.three_interpolate_affine(z, q_idx, q_weight, w2[:, :3].contiguous(), rel).view(shape)
                mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, rm, rv, momenta[0], epss[0],
                                                               training, tickets)
                ys.append(y)
                coefs.append((mean, invstd, scale, shift))
                cur, cur_coeff = y, (scale, shift)
                continue
            if pre is not None and i == 0:
                idx, _, npts = pre
                z = K.gemm_forward(w2, x)  # over the n + m points, not the m * ns gathered columns
                if training:  # the gather kernel leaves the rows' moments behind
                    y, mean, invstd, scale, shift = K.pregather_forward(
                        z, idx, npts, (gamma, beta, rm, rv, momenta[0], epss[0]))
                else:
                    y = K.pregather_forward(z, idx, npts)
                    mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, rm, rv, momenta[0],
                                                                   epss[0], False, tickets)
                ys.append(y)
                coefs.append((mean, invstd, scale, shift))
                cur, cur_coeff = y, (scale, shift)
                continue
            if virtual0 and i == 0:
                moments = K.first4_moments(x)
                mean, invstd, scale, shift = K.first4_bn(moments, x.numel() // 4, w2, gamma, beta, rm, rv,
                                                         momenta[0], epss[0])
                ys.append(x.new_empty(0))  # never materialised
                coefs.append((mean, invstd, scale, shift))
                cur, cur_coeff = None, (scale, shift)
                continue
            if virtual0 and i == 1:
                w0 = params[0].reshape(params[0].shape[0], -1)
                y, mean, invstd, scale, shift = K.gemm_forward_bn_lin4(w2, x, w0, cur_coeff, gamma, beta,
                                                                       rm, rv, momenta[1], epss[1])
                ys.append(y)
                coefs.append((mean, invstd, scale, shift))
                cur, cur_coeff = y, (scale, shift)
                continue
            if training and pool and i == n_layers - 1:
                # ... and so do the per-group extrema the max over nsample needs.  Where the layer's
                # backward can run from the Gram matrix of its input (SA2 - SA4: 128 -> 256,
                # csrc/mlp_pool_gram256.hip) the raw output is not stored at all: 268 MB at SA2
                epilogue = (i >= 1 and cur is not None and cur.dim() == 4 and cur.numel() > 0
                            and K.forward_pool_supported(w2, cur, cur_coeff))
                gram = epilogue and K.pool_gram_supported(w2, cur, cur.shape[3])
                # ... and in a pass that no backward follows (the EMA teacher, evaluation in training
                # mode) nobody reads the raw output at all once the extrema are known
                no_backward = not any(ctx.needs_input_grad)
                y, mean, invstd, scale, shift, ext = K.gemm_forward_bn(
                    w2, cur, cur_coeff, gamma, beta, rm, rv, momenta[i], epss[i], pool=True, tickets=tickets,
                    store=not (gram or (epilogue and no_backward)))
                if y is None:
                    y = x.new_empty(0)  # never materialised
                if gram:
                    gram_last, gram_ns = True, cur.shape[3]
            elif training:  # batch statistics come out of the GEMM epilogue where the shape allows
                y, mean, invstd, scale, shift = K.gemm_forward_bn(w2, cur, cur_coeff, gamma, beta,
                                                                  rm, rv, momenta[i], epss[i], tickets=tickets)
            else:
                y = K.gemm_forward(w2, cur, cur_coeff)
                mean, invstd, scale, shift = K.bn_coefficients(y, gamma, beta, rm, rv, momenta[i],
                                                               epss[i], training, tickets)
            ys.append(y)
            coefs.append((mean, invstd, scale, shift))
            cur, cur_coeff = y, (scale, shift)
        extra = []
        if pool and ext is not None:
            out, argmax, ymax = K.pool_from_extrema(ext, cur_coeff[0], cur_coeff[1])
            extra = [argmax, ymax]
        elif pool:
            out, argmax, ymax = K.bn_relu_pool(cur, cur_coeff[0], cur_coeff[1])
            extra = [argmax, ymax]
        else:
            out = K.bn_relu_apply(cur, cur_coeff[0], cur_coeff[1])
        flat = [t for c in coefs for t in c]
        ctx.save_for_backward(x, *ys, *flat, *extra, *params)
        ctx.n_layers, ctx.pool, ctx.training = n_layers, pool, training
        ctx.moments = moments  # not None: the first layer is virtual (ys[0] is a placeholder)
        ctx.gram_last = gram_last  # the last layer's raw output was not stored (ys[-1] is a placeholder)
        ctx.ns = gram_ns if gram_ns else (x.shape[3] if x.dim() == 4 else 0)
        ctx.pre = pre
        return out

    @staticmethod
    def backward(ctx, dout):
        from pointnet2 import _mlp_ext as K
        n, pool, training = ctx.n_layers, ctx.pool, ctx.training
        saved = ctx.saved_tensors
        x, ys = saved[0], saved[1:1 + n]
        pre_gather = ctx.pre
        if ctx.pre is not None and isinstance(ctx.pre[0], str):
            # the interpolation-commuted first layer: its weight gradient needs the layer's real
            # input cat([rel, interpolate(x)]) -- formed here, once, instead of in the forward pass
            # (the layer's input carries no gradient: asserted by interp_first_ok)
            from pointnet2 import _ext
            _, q_idx, q_weight, rel, shape = ctx.pre
            feats = torch.empty((x.shape[0], 3 + x.shape[1], rel.shape[2]), dtype=torch.float32, device=x.device)
            feats[:, :3].copy_(rel)
            _ext.three_interpolate_into(x, q_idx, q_weight, feats, 3)
            x = feats.view(x.shape[0], 3 + x.shape[1], shape[2], shape[3])
            pre_gather = None
        flat = saved[1 + n:1 + 5 * n]
        coefs = [flat[4 * i:4 * i + 4] for i in range(n)]  # mean, invstd, scale, shift
        pos = 1 + 5 * n
        extra = saved[pos:pos + (2 if pool else 0)]
        params = saved[pos + (2 if pool else 0):]
        dout = dout.contiguous()
        grads = [None] * (5 * n)
        need_dx = ctx.needs_input_grad[0]
        dy_tensor, fly, pooled, below = None, None, None, None
        dz = dout
        for i in range(n - 1, -1, -1):
            w, gamma = params[5 * i], params[5 * i + 1]
            w2 = w.reshape(w.shape[0], -1)
            mean, invstd, scale, shift = coefs[i]
            if i == n - 1 and pool and ctx.gram_last:
                # the layer's raw output does not exist: both products of its backward from the Gram
                # matrix of its input and one sparse column per (channel, group)
                dgamma, dbeta, coef = K.bn_relu_pool_backward_stats(
                    None, dz, extra[0], extra[1], gamma, scale, shift, mean, invstd, training, ns=ctx.ns,
                    tickets=ctx.tickets)
                grads[5 * i + 1], grads[5 * i + 2] = dgamma, dbeta
                dz, dw_last, below = K.pool_gram_backward(
                    w2, ys[i - 1], coefs[i - 1], params[5 * (i - 1) + 1], coef, coefs[i], dz, extra[0],
                    extra[1], ctx.ns, training)
                grads[5 * i] = dw_last.view_as(w)
                continue
            if i == n - 1 and pool:
                # dz of the pooled layer is one value per (channel, group): the GEMM operand
                # loads rebuild dy from y, dpooled and the arg-max, nothing dense is written
                dgamma, dbeta, coef = K.bn_relu_pool_backward_stats(
                    ys[i], dz, extra[0], extra[1], gamma, scale, shift, mean, invstd, training,
                    tickets=ctx.tickets)
                dy_tensor, fly = None, None
                pooled = (ys[i], dz, extra[0], scale, shift, mean, invstd, coef)
            else:
                pooled, dy_tensor = None, None
                if below is not None:  # left behind by the fused backward GEMM of layer i+1
                    dgamma, dbeta, coef = below
                elif i == 0 and ctx.moments is not None:
                    raise RuntimeError("the virtual first layer's BatchNorm sums must come from the "
                                       "fused backward kernel of the second layer")
                elif K.small_backward_prefers_dy(w2, ys[i]) and not (pre_gather is not None and i == 0):
                    # a small layer: dy written once and read by the pair launch, instead of
                    # re-formed by each of its tiles (_mlp_ext.small_backward_prefers_dy)
                    dy_tensor, dgamma, dbeta = K.bn_relu_backward(ys[i], dz, gamma, scale, shift, mean,
                                                                  invstd, training, ctx.tickets)
                    coef = None
                else:
                    dgamma, dbeta, coef = K.bn_relu_backward_stats(ys[i], dz, gamma, scale, shift,
                                                                   mean, invstd, training, ctx.tickets)
                fly = None if dy_tensor is not None else (ys[i], dz, scale, shift, mean, invstd, coef)
            grads[5 * i + 1], grads[5 * i + 2] = dgamma, dbeta
            m, k = w2.shape
            if pre_gather is not None and i == 0:
                # gradient of z_ext = W_0 . src_ext: the BatchNorm / ReLU backward of (y_0, dz) formed
                # on the fly, scatter-added over idx and summed per group; then two GEMMs over
                # the n + m points
                inverse = pre_gather[1]
                if inverse is None:  # forward ran without it (no gradient expected then): build it now
                    from pointnet2 import _ext
                    inverse = _ext.group_inverse(pre_gather[0], pre_gather[2])
                dzx = K.pregather_backward(fly, inverse, pre_gather[2])
                pair = K.gemm_backward_small(w2, x, None, dy=dzx, need_dx=need_dx)
                if pair is not None:
                    dx, grads[0] = pair[0], pair[1].view_as(w)
                    continue
                grads[0] = K.gemm_wgrad(m, k, x, None, dy=dzx).view_as(w)
                dx = K.gemm_dgrad(w2, dy=dzx).view_as(x) if need_dx else None
                continue
            virtual0 = ctx.moments is not None
            src = x if (i == 0 or (i == 1 and virtual0)) else ys[i - 1]
            src_coeff = None if i == 0 else (coefs[i - 1][2], coefs[i - 1][3])
            lin_w = params[0].reshape(params[0].shape[0], -1) if (i == 1 and virtual0) else None
            # both GEMMs from one pass over (y_i, dz) where the shape allows
            src_stats = None if i == 0 else (coefs[i - 1][0], coefs[i - 1][1],
                                             params[5 * (i - 1) + 1], training)
            both = None
            if dy_tensor is None:
                both = K.gemm_backward_fused(w2, src, src_coeff, fly, pooled, src_stats,
                                             need_dx=i > 0 or need_dx, lin_w=lin_w)
            if lin_w is not None and both is None:
                raise RuntimeError("the virtual first layer needs the fused backward kernel of the second")
            below = None
            if both is None and pooled is None and lin_w is None and (i > 0 or need_dx):
                # the small layers: both GEMMs in one launch (no BatchNorm sums for the layer below)
                pair = K.gemm_backward_small(w2, src, src_coeff, dy=dy_tensor, fly=fly)
                if pair is not None:
                    both = (pair[0], pair[1], None)
            if both is not None:
                below = both[2]  # BatchNorm-backward sums of layer i-1 (None for the first layer)
                grads[5 * i] = both[1].view_as(w)
                if i == 0:
                    dx = both[0].view_as(x) if need_dx else None
                else:
                    dz = both[0]
            elif i == 0:
                if isinstance(dz, K.GatedSums):
                    # the layer above never wrote the gradient w.r.t. this virtual layer's output: its
                    # one-pass backward left the gated sums the weight gradient needs (and `below`)
                    dw0 = K.wgrad_first4_from_gated(w2, dz, mean, invstd, coef, ctx.moments)
                else:
                    dw0 = K.wgrad_first4(w2, x, fly, ctx.moments) if (fly is not None and not need_dx) else None
                if dw0 is None and virtual0:
                    raise RuntimeError("the virtual first layer needs mlp_wgrad_first4")
                if dw0 is None:
                    dw0 = K.gemm_wgrad(m, k, x, None, dy_tensor, fly, pooled)
                grads[0 + 5 * i] = dw0.view_as(w)
                dx = K.gemm_dgrad(w2, dy_tensor, fly, pooled).view_as(x) if need_dx else None
            else:
                grads[5 * i] = K.gemm_wgrad(m, k, src, src_coeff, dy_tensor, fly, pooled).view_as(w)
                dz = K.gemm_dgrad(w2, dy_tensor, fly, pooled)  # gradient w.r.t. relu(bn(y_{i-1}))
        return (dx if need_dx else None, None, None, None, None, None, None, *grads)


_deferred_counters = None
_deferred_axpy = None   # (tensor, other, alpha): tensor += alpha * other, applied at context exit
# inside deferred_bn_counters() a gradient that is identically zero (the bias of a convolution that
# feeds a training-mode BatchNorm) may be returned as None instead of a freshly zeroed tensor:
# the train step's gradient packing substitutes zeros (votenet/step.py:_pack_gradients)
zero_grads_as_none = False


@contextlib.contextmanager
def deferred_bn_counters():
    """Inside this context the fused layers collect their `num_batches_tracked += 1` updates (and
    the running-mean corrections `rm += momentum * bias` of the head chains) and apply them with
    ONE multi-tensor add each at exit (37 + 6 one-element / one-row kernels per forward otherwise)."""
    global _deferred_counters, _deferred_axpy
    previous, _deferred_counters = _deferred_counters, []
    previous_axpy, _deferred_axpy = _deferred_axpy, []
    try:
        yield
    finally:
        pending, _deferred_counters = _deferred_counters, previous
        axpy, _deferred_axpy = _deferred_axpy, previous_axpy
        if pending:
            torch._foreach_add_(pending, 1)
        by_alpha = {}
        for t, other, alpha in axpy:
            by_alpha.setdefault(float(alpha), ([], []))
            by_alpha[float(alpha)][0].append(t)
            by_alpha[float(alpha)][1].append(other)
        for alpha, (ts, others) in by_alpha.items():
            torch._foreach_add_(ts, others, alpha=alpha)


@contextlib.contextmanager
def zero_grads_none():
    """Run a backward pass with `zero_grads_as_none` set (the caller packs the gradients itself
    and treats a missing one as zeros)."""
    global zero_grads_as_none
    previous, zero_grads_as_none = zero_grads_as_none, True
    try:
        yield
    finally:
        zero_grads_as_none = previous


def deferred_axpy(tensor, other, alpha):
    """tensor += alpha * other, now or -- inside deferred_bn_counters() -- batched at its exit
    (nothing reads `tensor`, a BatchNorm running mean, before then)."""
    if _deferred_axpy is not None:
        _deferred_axpy.append((tensor, other.detach(), alpha))
    else:
        tensor.add_(other.detach(), alpha=alpha)


def bump_batches_tracked(counter):
    if _deferred_counters is not None:
        _deferred_counters.append(counter)
    else:
        counter.add_(1)


def _fused_enabled():
    return os.environ.get("PN2_FUSED_MLP", "1") != "0"


def _mfma_enabled():
    return os.environ.get("PN2_MFMA_MLP", "1") != "0"


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d (+BN+ReLU) layers `layer0..layerK` applied to a (B, C, npoint,
    nsample) tensor: the grouped shared MLP of a set-abstraction layer.

    Same module tree as the reference (pytorch_utils.py:14-39).  On the GPU, layers of the
    standard shape conv(1x1, no bias) -> BatchNorm2d -> ReLU run their BatchNorm/ReLU (and, via
    forward_pooled, the max-pool over nsample that follows the last layer in every SA module)
    through the fused kernels of pointnet2._mlp_ext; any other configuration, and CPU tensors,
    use the plain torch modules."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False,
                 first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # first pre-act layer: no bn / activation
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=None if plain else activation, preact=preact))

    @staticmethod
    def _fusable(layer):
        kids = dict(layer.named_children())
        if set(kids) != {"conv", "bn", "activation"} or list(kids)[0] != "conv":
            return False
        conv, bn_wrap, act = kids["conv"], kids["bn"], kids["activation"]
        bns = list(bn_wrap.children())
        return (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.bias is None
                and conv.stride == (1, 1) and conv.padding == (0, 0) and len(bns) == 1
                and isinstance(bns[0], nn.BatchNorm2d) and bns[0].affine
                and bns[0].track_running_stats and bns[0].momentum is not None
                and isinstance(act, nn.ReLU))

    def _use_fused(self, x):
        return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and len(self) > 0
                and _fused_enabled() and all(self._fusable(layer) for layer in self))

    def _run(self, x, pool, pre=None):
        from pointnet2 import _mlp_ext as K
        layers = list(self)
        # the module's own counters for the one-launch reductions of its layers (include/mlp_hip.h
        # `tickets`): nothing in the library is keyed by stream
        tickets = K.tickets_of(self, max(layer.conv.out_channels for layer in layers), x.device)
        if _mfma_enabled():
            bns = [next(layer.bn.children()) for layer in layers]
            training = bns[0].training
            if all(bn.training == training for bn in bns):
                params = []
                for layer, bn in zip(layers, bns):
                    if training:
                        bump_batches_tracked(bn.num_batches_tracked)
                    params += [layer.conv.weight, bn.weight, bn.bias, bn.running_mean,
                               bn.running_var]
                return _FusedMLPChain.apply(x, pool, training, [bn.momentum for bn in bns],
                                            [bn.eps for bn in bns], pre, tickets, *params)
        for i, layer in enumerate(layers):
            bn = next(layer.bn.children())
            y = layer.conv(x)
            training = bn.training
            if training:
                bump_batches_tracked(bn.num_batches_tracked)
            op = _BNReLUMaxPool if (pool and i == len(layers) - 1) else _BNReLU
            x = op.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum,
                         bn.eps, training, tickets)
        return x

    def forward(self, x):
        if self._use_fused(x):
            return self._run(x, pool=False)
        return super().forward(x)

    def pregather_ok(self, xyz, new_xyz, features, m, ns):
        """Can forward_pregathered replace forward_pooled(grouped) for these inputs (m groups of
        ns members)?"""
        if features is None or len(self) < 2 or not _pregather_enabled():
            return False
        if not (features.is_cuda and features.dtype == torch.float32 and _fused_enabled()
                and _mfma_enabled() and all(self._fusable(layer) for layer in self)):
            return False
        bns = [next(layer.bn.children()) for layer in self]
        if any(bn.training != bns[0].training for bn in bns):
            return False
        from pointnet2 import _mlp_ext as K
        conv = self[0].conv
        return (conv.in_channels == features.shape[1] + 3 and
                K.pregather_supported(features.shape[0], conv.out_channels, xyz.shape[1], m, ns))

    def forward_pregathered(self, xyz, new_xyz, features, idx, inverse, scale):
        """forward_pooled of the grouped tensor [(xyz[idx] - new_xyz) * scale ; features[idx]]
        WITHOUT forming it: the first layer runs before the gather (csrc/mlp_pregather.hip).
        xyz (B,N,3), new_xyz (B,m,3), features (B,C,N), idx (B,m,ns) int32, inverse = its
        _ext.group_inverse; -> (B, C', m)."""
        src = _PackPoints.apply(xyz, new_xyz, features, float(scale))
        return self._run(src, pool=True, pre=(idx, inverse, xyz.shape[1]))

    def interp_first_ok(self, features, idx):
        """forward_pooled_interp covers: features without gradient, the MFMA chain, two layers or
        more, shapes of the affine interpolation kernel."""
        from pointnet2 import _ext
        layers = list(self)
        recording = torch.is_grad_enabled()
        if recording and (features.requires_grad or os.environ.get("PN2_INTERP_FIRST_TRAIN", "1") == "0"):
            return False  # (a gradient w.r.t. the features would need the scatter form)
        return (os.environ.get("PN2_INTERP_FIRST", "1") != "0"
                and _mfma_enabled() and len(layers) >= 2 and features.is_cuda and features.dim() == 3
                and features.dtype == torch.float32 and hasattr(_ext, "three_interpolate_affine")
                and layers[0].conv.weight.shape[1] == features.shape[1] + 3
                and _ext.three_interpolate_affine_supported(layers[0].conv.weight.shape[0],
                                                            features.shape[2], idx.shape[1]))

    def forward_pooled_interp(self, features, idx, weight, rel, npoint, nsample):
        """forward_pooled(cat([rel, three_interpolate(features, idx, weight)]).view(B, 3 + C, npoint,
        nsample)) without that tensor: features (B,C,m) of the source points, idx / weight
        (B, npoint*nsample, 3), rel (B, 3, npoint*nsample); none of them with a gradient
        (interp_first_ok).  In a training pass the layer's real input is formed once, in the backward
        pass, for the weight gradient."""
        shape = (features.shape[0], list(self)[0].conv.weight.shape[0], npoint, nsample)
        return self._run(features.contiguous(), pool=True,
                         pre=("interp", idx.contiguous(), weight.contiguous(), rel.contiguous(), shape))

    def forward_pooled(self, x):
        """max over the last axis of forward(x): (B, C, npoint, nsample) -> (B, C', npoint)."""
        if self._use_fused(x):
            return self._run(x, pool=True)
        return torch.max(super().forward(x), dim=3)[0]


class _PackPoints(Function):
    """(xyz (B,N,3), new_xyz (B,m,3), features (B,C,N), s) -> src_ext (B, 3+C, N+m), the operand of
    the pre-gather first layer (_mlp_ext.pregather_pack).  Its gradient splits back into the three
    inputs: rows 0..2 are the coordinates' (columns < N: xyz, the others: new_xyz -- what
    QueryAndGroup's backward scatters / sums, pointnet2_utils.py:348-358), the other rows the
    features'."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, s):
        from pointnet2 import _mlp_ext as K
        ctx.dims = (xyz.shape[1], new_xyz.shape[1], float(s))
        return K.pregather_pack(xyz.contiguous(), new_xyz.contiguous(), features.contiguous(), s)

    @staticmethod
    def backward(ctx, dsrc):
        from pointnet2 import _mlp_ext as K
        n, m, s = ctx.dims
        dsrc = dsrc.contiguous()
        dfeat = K.pregather_unpack_grad(dsrc, n, m) if ctx.needs_input_grad[2] else None
        dxyz = dnew = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dc = dsrc[:, :3].transpose(1, 2)  # (B, N+m, 3)
            if s != 1.0:
                dc = dc * s
            dxyz = dc[:, :n].contiguous() if ctx.needs_input_grad[0] else None
            dnew = dc[:, n:].contiguous() if ctx.needs_input_grad[1] else None
        return dxyz, dnew, dfeat, None


def _pregather_enabled():
    return os.environ.get("PN2_PREGATHER", "1") != "0"


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False,
                 init=None, preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        bn_unit = BatchNorm1d(in_size if preact else out_size) if bn else None
        _assemble(self, name, "fc", fc, bn_unit, activation, preact)


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """Sets momentum = bn_lambda(epoch) on every batch-norm layer of `model` at each step()."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))
