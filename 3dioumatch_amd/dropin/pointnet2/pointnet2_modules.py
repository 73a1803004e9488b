"""pointnet2.pointnet2_modules -- set-abstraction (SA) and feature-propagation (FP) layers
(host-side mirror of the reference pointnet2/pointnet2_modules.py: _PointnetSAModuleBase :30-80,
PointnetSAModuleMSG :83-128, PointnetSAModule :131-166, PointnetSAModuleVotes :169-277,
PointnetSAModuleMSGVotes :280-359, PointnetFPModule :362-422, PointnetLFPModuleMSG :425-503).

Same class names, constructor keywords, attribute names (`mlp_module`, `groupers`, `mlps`,
`mlp`, `post_mlp` -> identical state_dict keys) and forward signatures / return tuples.
VoteNet uses PointnetSAModuleVotes and PointnetFPModule (models/backbone_module.py:35-72,
models/proposal_module.py:72-79).

The custom operators underneath are the gfx950 kernels of this package; the index tensors
(`inds`) stay int32 exactly as they come out of furthest point sampling.
"""
import os
import sys
from typing import List

import torch
import torch.nn as nn

try:  # imported as pointnet2.pointnet2_modules
    from . import pointnet2_utils
    from . import pytorch_utils as pt_utils
except ImportError:  # imported flat, as the reference models do after sys.path.append(.../pointnet2)
    sys.path.append(os.path.dirname(os.path.abspath(__file__)))
    import pointnet2_utils
    import pytorch_utils as pt_utils


def _sample_centroids(xyz, npoint, inds=None, radius=None):
    """FPS (unless `inds` is given) and the centroid coordinates (B, npoint, 3).  With `radius`
    the sampling may also leave the cloud's cell lists for ball queries of that radius behind
    (third return value, None otherwise)."""
    lists = None
    first_tie = None
    if inds is None:
        # a cloud that is the previous module's centroids, in pick order, samples to 0, 1, 2, ...
        # unless that run met an exact tie (pointnet2_utils.sample_chain, include/pn2_hip.h)
        first_tie = pointnet2_utils.head_record(xyz)
        inds, lists, first_tie = pointnet2_utils.sample_chain(xyz, npoint, radius, first_tie,
                                                              head=first_tie is not None)
    flipped = xyz.transpose(1, 2).contiguous()
    new_xyz = pointnet2_utils.gather_operation(flipped, inds).transpose(1, 2).contiguous()
    pointnet2_utils.remember_head(new_xyz, first_tie)
    if lists is not None:  # the centroids are this sampling call's picks: its query plans apply
        lists.mark_centroids(new_xyz, inds)
    return (new_xyz, inds, lists) if radius is not None else (new_xyz, inds)


def _pool_max(x):
    """max over the nsample axis of (B, C, npoint, nsample) -> (B, C, npoint)."""
    return torch.max(x, dim=3)[0]


def _build_scales(owner, npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly):
    owner.groupers = nn.ModuleList()
    owner.mlps = nn.ModuleList()
    for radius, nsample, spec in zip(radii, nsamples, mlps):
        owner.groupers.append(
            pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                          sample_uniformly=sample_uniformly)
            if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
        if use_xyz:
            spec[0] += 3  # in place, like the reference (the caller's list is modified)
        owner.mlps.append(pt_utils.SharedMLP(spec, bn=bn))


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def _multi_scale(self, xyz, new_xyz, features):
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            outs.append(mlp.forward_pooled(grouper(xyz, new_xyz, features)))
        return torch.cat(outs, dim=1)

    def forward(self, xyz, features=None):
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B,sum C_k,npoint)"""
        new_xyz = _sample_centroids(xyz, self.npoint)[0] if self.npoint is not None else None
        return new_xyz, self._multi_scale(xyz, new_xyz, features)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping."""

    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int],
                 mlps: List[List[int]], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        _build_scales(self, npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly)


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class PointnetSAModuleVotes(nn.Module):
    """Single-scale set abstraction that also returns (or accepts) the sampled indices, as
    VoteNet needs them to look up ground-truth votes.

    forward(xyz (B,N,3), features (B,C,N), inds (B,npoint) int32 or None)
      -> new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint), inds [, unique_cnt]
    """

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True,
                 pooling: str = 'max', sigma: float = None, normalize_xyz: bool = False,
                 sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else self.radius / 2  # RBF pooling width
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)
        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = pt_utils.SharedMLP(mlp_spec, bn=bn)

    def _pool(self, feats, grouped_xyz):
        if self.pooling == 'max':
            return _pool_max(feats)  # (the forward below pools inside the shared MLP instead)
        if self.pooling == 'avg':
            return feats.mean(dim=3)
        if self.pooling == 'rbf':
            # Gaussian weights on the (normalised) local offsets, summed and divided by nsample
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1) / (self.sigma ** 2) / 2)
            return torch.sum(feats * rbf.unsqueeze(1), -1) / float(self.nsample)
        raise ValueError("unknown pooling %r" % (self.pooling,))

    def forward(self, xyz, features=None, inds=None, ball_idx=None, new_xyz=None, ball_inv=None,
                grouped=None):
        """ball_idx / new_xyz / ball_inv: optional precomputed ball-query indices
        (B,npoint,nsample) int32, centroid coordinates (B,npoint,3) for the centroids `inds`, and
        the inverse index of ball_idx (_ext.group_inverse) -- all depend on coordinates only; see
        votenet/step.py.  grouped: the layer's whole grouped tensor (B, 3+C, npoint, nsample),
        precomputed by the fused query + gather kernel for a layer whose features are network
        inputs (no gradient flows into them); requires inds and new_xyz."""
        if inds is not None:
            assert inds.shape[1] == self.npoint
        lists = None
        if new_xyz is not None and inds is not None:
            pass  # the index chain was computed ahead of time
        elif self.npoint is not None:
            fused = isinstance(self.grouper, pointnet2_utils.QueryAndGroup) and \
                not self.grouper.sample_uniformly and ball_idx is None
            if fused:  # the sampling kernel may leave the cell lists of this layer's balls behind
                new_xyz, inds, lists = _sample_centroids(xyz, self.npoint, inds, self.radius)
            else:
                new_xyz, inds = _sample_centroids(xyz, self.npoint, inds)
        else:
            new_xyz = None
        if grouped is not None:
            if new_xyz is None or inds is None or self.pooling != 'max' or self.ret_unique_cnt or \
                    (features is not None and features.requires_grad):
                raise RuntimeError("a precomputed grouped tensor needs inds, new_xyz, max pooling "
                                   "and input features without gradient")
            return new_xyz, self.mlp_module.forward_pooled(grouped), inds
        if (self.pooling == 'max' and not self.ret_unique_cnt and self.use_xyz
                and isinstance(self.grouper, pointnet2_utils.QueryAndGroup)
                and not self.grouper.sample_uniformly and new_xyz is not None
                and self.mlp_module.pregather_ok(xyz, new_xyz, features, self.npoint, self.nsample)):
            # the first layer commutes with the gather: run it over the N points and gather its
            # output; the (3+C)-row grouped tensor and its gradient are never formed
            if ball_idx is None:
                if lists is not None and self.nsample <= 256:
                    ball_idx = pointnet2_utils._ext.ball_query_prebuilt(new_xyz, xyz, self.radius,
                                                                        self.nsample, lists)
                else:
                    ball_idx = pointnet2_utils.ball_query(self.radius, self.nsample, xyz, new_xyz)
            # the inverse index serves the backward only: no-grad passes (teacher, inference) skip its
            # sort kernel; `False` = not built (the shape is inside the fast backward's range)
            need_bwd = torch.is_grad_enabled() and (
                features.requires_grad or xyz.requires_grad or new_xyz.requires_grad
                or any(p.requires_grad for p in self.mlp_module.parameters()))
            if ball_inv is None:
                if need_bwd:
                    ball_inv = pointnet2_utils._ext.group_inverse(ball_idx, xyz.shape[1])
                elif pointnet2_utils._ext.group_inverse_supported(xyz.shape[1], self.npoint, self.nsample):
                    ball_inv = False
            if ball_inv is not None:
                scale = 1.0 / self.radius if self.normalize_xyz else 1.0
                return new_xyz, self.mlp_module.forward_pregathered(
                    xyz, new_xyz, features, ball_idx, None if ball_inv is False else ball_inv, scale), inds
        if ball_idx is not None and isinstance(self.grouper, pointnet2_utils.QueryAndGroup):
            grouped = self.grouper(xyz, new_xyz, features, ball_idx, None, ball_inv)
        elif lists is not None:
            grouped = self.grouper(xyz, new_xyz, features, None, lists)
        else:
            grouped = self.grouper(xyz, new_xyz, features)
        unique_cnt = grouped[2] if self.ret_unique_cnt else None
        if self.pooling == 'max':  # BN + ReLU + max over nsample fused into the last layer
            new_features = self.mlp_module.forward_pooled(grouped[0])
        else:
            new_features = self._pool(self.mlp_module(grouped[0]), grouped[1])
        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetSAModuleMSGVotes(nn.Module):
    """Multi-scale set abstraction returning the sampled indices."""

    def __init__(self, *, mlps: List[List[int]], npoint: int, radii: List[float],
                 nsamples: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.npoint = npoint
        _build_scales(self, npoint, radii, nsamples, mlps, bn, use_xyz, sample_uniformly)

    def forward(self, xyz, features=None, inds=None):
        if self.npoint is not None:
            new_xyz, inds = _sample_centroids(xyz, self.npoint, inds)
        else:
            new_xyz = None
        outs = [mlp.forward_pooled(grouper(xyz, new_xyz, features))
                for grouper, mlp in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(outs, dim=1), inds


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance interpolation of `known_feats` from the three
    nearest known points, concatenated with `unknow_feats`, then a shared MLP.

    forward(unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m))
      -> (B, mlp[-1], n)
    """

    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = pt_utils.SharedMLP(mlp, bn=bn)

    @staticmethod
    def interpolation(unknown, known):
        """(idx (B,n,3) int32, weight (B,n,3)): the three nearest known points of every unknown
        point and their normalised inverse distances -- coordinates only."""
        return pointnet2_utils.three_nn_with_weights(unknown, known)

    def forward(self, unknown, known, unknow_feats, known_feats, interpolation=None):
        if known is None:
            interpolated = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
            stacked = interpolated if unknow_feats is None else \
                torch.cat([interpolated, unknow_feats], dim=1)
        else:
            idx, weight = interpolation if interpolation is not None else \
                self.interpolation(unknown, known)
            if unknow_feats is None:
                stacked = pointnet2_utils.three_interpolate(known_feats, idx, weight)
            else:  # interpolation written straight into the concatenated tensor
                stacked = pointnet2_utils.interpolate_concat(known_feats, idx, weight, unknow_feats)
        return self.mlp(stacked.unsqueeze(-1)).squeeze(-1)


class PointnetLFPModuleMSG(nn.Module):
    """Learnable feature propagation: group features of (xyz1, features1) around xyz2."""

    def __init__(self, *, mlps: List[List[int]], radii: List[float], nsamples: List[int],
                 post_mlp: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.post_mlp = pt_utils.SharedMLP(post_mlp, bn=bn)
        _build_scales(self, 1, radii, nsamples, mlps, bn, use_xyz, sample_uniformly)

    def forward(self, xyz2, xyz1, features2, features1):
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            feats = mlp.forward_pooled(grouper(xyz1, xyz2, features1))
            if features2 is not None:
                feats = torch.cat([feats, features2], dim=1)
            outs.append(self.post_mlp(feats.unsqueeze(-1)))
        return torch.cat(outs, dim=1).squeeze(-1)
