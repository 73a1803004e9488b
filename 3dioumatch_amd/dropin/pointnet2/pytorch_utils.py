"""pointnet2.pytorch_utils -- shared-MLP building blocks (host-side mirror).

Public names, constructor arguments and the MODULE TREE (hence state_dict keys such as
`layer0.conv.weight`, `layer0.bn.bn.running_mean`) follow the reference
pointnet2/pytorch_utils.py:14-39 (SharedMLP), :42-67 (BatchNorm wrappers), :70-124
(_ConvBase), :127-236 (Conv1d/2d/3d), :239-270 (FC), :272-299 (BN momentum scheduler), so
checkpoints are interchangeable.  The implementation is written fresh around one builder.
"""
import torch.nn as nn


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


class SharedMLP(nn.Sequential):
    """Stack of 1x1 Conv2d (+BN+ReLU) layers `layer0..layerK` applied to a (B, C, npoint,
    nsample) tensor: the grouped shared MLP of a set-abstraction layer."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False,
                 first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # first pre-act layer: no bn / activation
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=None if plain else activation, preact=preact))


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
