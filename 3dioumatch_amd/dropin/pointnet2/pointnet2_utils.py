"""pointnet2.pointnet2_utils -- autograd wrappers of the set-abstraction operators and the
grouping modules (host-side mirror of the reference pointnet2/pointnet2_utils.py: the six
autograd Functions at :52-292 and QueryAndGroup / GroupAll at :295-426).

Same public names, call signatures and autograd contract:
  furthest_point_sample(xyz, npoint) -> (B, npoint) int32, non-differentiable
  gather_operation(features, idx)    -> (B, C, npoint); backward = scatter-add
  three_nn(unknown, known)           -> (dist, idx) with dist = sqrt(squared distance)
  three_interpolate(features, idx, weight)
  grouping_operation(features, idx)  -> (B, C, npoint, nsample); backward = scatter-add
  ball_query(radius, nsample, xyz, new_xyz) -> (B, npoint, nsample) int32

What differs underneath: every op is a hand-written gfx950 kernel behind the C ABI
(include/pn2_hip.h), and QueryAndGroup runs ONE fused front end (ball query + both gathers +
centroid subtraction + 1/radius scaling + channel concatenation written once) instead of the
reference's ball_query, two group_points, a subtract, a divide and a torch.cat.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

import pointnet2._ext as _ext


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        fps_inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(fps_inds)
        return fps_inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n_points = features.size(2)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.gather_points_grad(grad_out.contiguous(), idx, ctx.n_points), None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


def three_nn_with_weights(unknown, known):
    """(idx (B,n,3) int32, weight (B,n,3)): the three nearest known points of every unknown point
    and their normalised inverse distances 1 / (dist + 1e-8) / sum -- coordinates only, nothing
    differentiable (pointnet2_modules.py:393-398, grid_conv_module.py:87-98).  One kernel behind
    the search where the extension has it (the stand-in of the CPU tests has not)."""
    fused = getattr(_ext, "three_nn_weights", None)
    if fused is not None and unknown.is_cuda:
        with torch.no_grad():
            dist2, idx = _ext.three_nn(unknown.detach(), known.detach())
            return idx, fused(dist2)
    dist, idx = three_nn(unknown, known)
    recip = 1.0 / (dist + 1e-8)
    return idx, (recip / torch.sum(recip, dim=2, keepdim=True)).contiguous()


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.m_known = features.size(2)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_features = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight,
                                                    ctx.m_known)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class InterpolateConcat(Function):
    """cat([three_interpolate(known_feats, idx, weight), skip_feats], dim=1) without the
    intermediate: the interpolation is written straight into the first channels of the output,
    and its gradient is read straight out of the incoming gradient (pointnet2_modules.py:404-410:
    the feature-propagation layer's interpolate + torch.cat)."""

    @staticmethod
    def forward(ctx, known_feats, idx, weight, skip_feats):
        b, c2, m = known_feats.shape
        n = idx.shape[1]
        c1 = skip_feats.shape[1]
        out = torch.empty((b, c2 + c1, n), dtype=torch.float32, device=known_feats.device)
        _ext.three_interpolate_into(known_feats.contiguous(), idx, weight, out, 0)
        out[:, c2:].copy_(skip_feats)
        ctx.save_for_backward(idx, weight)
        ctx.dims = (c2, c1, m)
        return out

    @staticmethod
    def backward(ctx, grad):
        idx, weight = ctx.saved_tensors
        c2, c1, m = ctx.dims
        grad = grad.contiguous()
        g_known = _ext.three_interpolate_grad_from(grad, 0, c2, idx, weight, m) \
            if ctx.needs_input_grad[0] else None
        g_skip = grad[:, c2:] if ctx.needs_input_grad[3] else None
        return g_known, None, None, g_skip


def interpolate_concat(known_feats, idx, weight, skip_feats):
    """The fused form on the GPU extension, the reference composition elsewhere."""
    if hasattr(_ext, "three_interpolate_into") and known_feats.is_cuda:
        return InterpolateConcat.apply(known_feats, idx, weight, skip_feats)
    return torch.cat([three_interpolate(known_feats, idx, weight), skip_feats], dim=1)


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n_points = features.size(2)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return _ext.group_points_grad(grad_out.contiguous(), idx, ctx.n_points), None


grouping_operation = GroupingOperation.apply


def sample_with_cell_lists(xyz, npoint, radius):
    """furthest_point_sample(xyz, npoint) that also returns the cell lists of `xyz` for ball
    queries of `radius` when the sampling kernel can leave them behind (large clouds on the GPU:
    SA1), else None.  A set-abstraction layer samples a cloud and then queries balls in the same
    cloud (pointnet2_modules.py:236-250): with the lists the query + gathers are one kernel."""
    make = getattr(_ext, "furthest_point_sampling_with_grid", None)
    if make is None or not xyz.is_cuda or radius is None:
        return furthest_point_sample(xyz, npoint), None
    inds, lists = make(xyz.detach(), npoint, radius)
    return inds, lists


# Centroid tensors that are the picks of a sampling run, in pick order: id -> (weak reference,
# in-place version, first_tie of the run).  A set-abstraction module hands its new_xyz to the next
# module as that module's cloud (backbone_module.py:97-112): the next module's sampling looks its
# cloud up here and, when it finds it, answers from the tie record (sample_chain).  A tensor that
# was sliced, permuted, copied or modified in place is a different object / version: no record.
_HEADS = {}


def remember_head(new_xyz, first_tie):
    import weakref
    if first_tie is None:
        return
    if len(_HEADS) > 64:
        for key in [k for k, (ref, _, _) in _HEADS.items() if ref() is None]:
            del _HEADS[key]
    version = _ext._tensor_version(new_xyz)
    if version is None:  # inference-mode tensors keep no version counter: not remembered
        return
    _HEADS[id(new_xyz)] = (weakref.ref(new_xyz), version, first_tie)


def head_record(xyz):
    """first_tie of the sampling run whose picks `xyz` holds in order, or None."""
    rec = _HEADS.get(id(xyz))
    if rec is None:
        return None
    ref, version, first_tie = rec
    if ref() is not xyz or _ext._tensor_version(xyz) != version or first_tie.device != xyz.device or \
            first_tie.numel() != xyz.shape[0]:
        return None
    return first_tie


def sample_chain(xyz, npoint, radius, first_tie=None, head=False):
    """One link of a set-abstraction stack's sampling chain.  head=False: xyz is a raw cloud --
    sample_with_cell_lists that also records the run's first tie when the kernel can
    (`_ext.furthest_point_sampling_ties`).  head=True: xyz is the previous link's centroids in
    pick order -- `_ext.furthest_point_sampling_prefix`: 0..npoint-1 without a round for every
    cloud whose chain had neither a tie nor a repeated pick among the picks it holds.  Returns (inds, lists, first_tie)."""
    ties = getattr(_ext, "furthest_point_sampling_ties", None)
    if ties is None or not xyz.is_cuda:
        inds, lists = sample_with_cell_lists(xyz, npoint, radius)
        return inds, lists, None
    if head:
        return _ext.furthest_point_sampling_prefix(xyz.detach(), npoint, first_tie), None, first_tie
    return ties(xyz.detach(), npoint, radius)


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        inds = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply


class _FusedQueryAndGroup(Function):
    """(xyz, new_xyz, features) -> (B, 3+C, npoint, nsample) in one pass, plus idx.

    Gradients equal those of the reference composition (pointnet2_utils.py:348-358):
      d features = scatter-add of the feature channels         (group_points_grad)
      d xyz      = scatter-add of the xyz channels * scale     (xyz enters through its gather)
      d new_xyz  = -sum over nsample of the xyz channels * scale
    """

    @staticmethod
    def forward(ctx, xyz, new_xyz, features, radius, nsample, normalize_xyz, idx=None, lists=None,
                inverse=None):
        if lists is not None and idx is None:
            idx, grouped = _ext.query_and_group(new_xyz, xyz, features, radius, nsample,
                                                normalize_xyz, None, lists)
        else:
            idx, grouped = _ext.query_and_group(new_xyz, xyz, features, radius, nsample,
                                                normalize_xyz, idx)
        ctx.save_for_backward(idx)
        ctx.inverse = inverse  # inverse index of idx (group_inverse), when the caller has one
        ctx.n_points = xyz.size(1)
        ctx.scale = (1.0 / radius) if normalize_xyz else 1.0
        ctx.has_features = features is not None
        ctx.mark_non_differentiable(idx)
        return grouped, idx

    @staticmethod
    def backward(ctx, grad_grouped, _grad_idx=None):
        (idx,) = ctx.saved_tensors
        need_xyz, need_new_xyz, need_feat = ctx.needs_input_grad[:3]
        g_xyz = g_new = g_feat = None
        if need_feat and ctx.has_features:
            if ctx.inverse is not None:  # every element of the gradient is added exactly once,
                # and the feature channels are read where they lie
                g_feat = _ext.group_points_grad_sorted(grad_grouped.contiguous(), ctx.inverse,
                                                       ctx.n_points, 3)
            else:
                g_feat = _ext.group_points_grad(grad_grouped[:, 3:].contiguous(), idx, ctx.n_points)
        if need_xyz or need_new_xyz:
            gx = grad_grouped[:, :3]
            if ctx.scale != 1.0:
                gx = gx * ctx.scale
            if need_xyz:
                g_xyz = _ext.group_points_grad(gx.contiguous(), idx, ctx.n_points).transpose(1, 2)
            if need_new_xyz:
                g_new = -gx.sum(dim=3).transpose(1, 2)
        return g_xyz, g_new, g_feat, None, None, None, None, None, None


class QueryAndGroup(nn.Module):
    """Ball query of `radius` around each centroid, then gather of (relative xyz, features).

    forward(xyz (B,N,3), new_xyz (B,npoint,3), features (B,C,N) or None)
      -> (B, 3+C, npoint, nsample) [, grouped_xyz (B,3,npoint,nsample)] [, unique_cnt]
    """

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt
        if self.ret_unique_cnt:
            assert self.sample_uniformly

    def _resample_uniformly(self, idx):
        # replace the first-hit padding by uniform re-draws of the unique members of each
        # ball (host loop, as in the reference, pointnet2_utils.py:337-346; off by default)
        unique_cnt = torch.zeros((idx.shape[0], idx.shape[1]))
        for b in range(idx.shape[0]):
            for j in range(idx.shape[1]):
                members = torch.unique(idx[b, j, :])
                k = members.shape[0]
                unique_cnt[b, j] = k
                draw = torch.randint(0, k, (self.nsample - k,), dtype=torch.long)
                idx[b, j, :] = torch.cat((members, members[draw]))
        return unique_cnt

    def forward(self, xyz, new_xyz, features=None, idx=None, lists=None, inverse=None):
        """idx: optional ball-query result for (xyz, new_xyz) computed earlier; lists: optional
        cell lists of xyz for this radius (sample_with_cell_lists); inverse: optional inverse
        index of idx (_ext.group_inverse) for the backward scatter-add."""
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        unique_cnt = None
        if not self.sample_uniformly:
            grouped, _idx = _FusedQueryAndGroup.apply(xyz, new_xyz, features, self.radius,
                                                      self.nsample, self.normalize_xyz, idx, lists,
                                                      inverse)
            grouped_xyz = grouped[:, :3]
            new_features = grouped if self.use_xyz else grouped[:, 3:]
        else:
            idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
            unique_cnt = self._resample_uniformly(idx)
            grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
            grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
            if self.normalize_xyz:
                grouped_xyz = grouped_xyz / self.radius
            if features is not None:
                grouped_features = grouping_operation(features, idx)
                new_features = torch.cat([grouped_xyz, grouped_features], dim=1) \
                    if self.use_xyz else grouped_features
            else:
                new_features = grouped_xyz
        ret = [new_features]
        if self.ret_grouped_xyz:
            ret.append(grouped_xyz)
        if self.ret_unique_cnt:
            ret.append(unique_cnt)
        return ret[0] if len(ret) == 1 else tuple(ret)


class GroupAll(nn.Module):
    """Single group holding every point: (B, 3+C, 1, N)."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            new_features = grouped_xyz
        else:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) \
                if self.use_xyz else grouped_features
        return (new_features, grouped_xyz) if self.ret_grouped_xyz else new_features
