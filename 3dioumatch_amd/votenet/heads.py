"""Voting, proposal and IoU-estimation heads of VoteNet-IoU.

Host-side mirror of the reference models/voting_module.py:15-65, models/proposal_module.py:24-125
and models/grid_conv_module.py:22-116: same attribute names (-> state_dict keys), same
end_points keys and tensor shapes.  Device-agnostic (the reference hard-codes .cuda()).

GridConv differs in HOW, not in WHAT: the reference materialises the three nearest seeds of
every grid point with torch.gather / per-batch index_select lists (a (B, K*64*3, 256) tensor,
~0.8 GB at B=8, K=512) and re-derives the distances; here the distances come straight from the
three_nn kernel and the weighted sum is the three_interpolate kernel (same three products
summed left to right), since the seed features are detached on this branch.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from pointnet2 import pointnet2_utils
from pointnet2 import pytorch_utils as pt_utils
from pointnet2.pointnet2_modules import PointnetSAModuleVotes

from .fused_head import head_chain


class VotingModule(nn.Module):
    """seed (xyz, features) -> votes: xyz + offset, features + residual (vote_factor per seed)."""

    def __init__(self, vote_factor, seed_feature_dim):
        super().__init__()
        self.vote_factor = vote_factor
        self.in_dim = seed_feature_dim
        self.out_dim = self.in_dim  # residual features: widths must agree
        self.conv1 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv2 = nn.Conv1d(self.in_dim, self.in_dim, 1)
        self.conv3 = nn.Conv1d(self.in_dim, (3 + self.out_dim) * self.vote_factor, 1)
        self.bn1 = nn.BatchNorm1d(self.in_dim)
        self.bn2 = nn.BatchNorm1d(self.in_dim)

    def forward(self, seed_xyz, seed_features):
        b, num_seed = seed_xyz.shape[:2]
        net = head_chain(seed_features, self.conv1, self.bn1, self.conv2, self.bn2, self.conv3)
        net = net.transpose(2, 1).view(b, num_seed, self.vote_factor, 3 + self.out_dim)
        offset, residual = torch.split(net, [3, self.out_dim], dim=-1)
        vote_xyz = (seed_xyz.unsqueeze(2) + offset).reshape(b, num_seed * self.vote_factor, 3)
        vote_features = seed_features.transpose(2, 1).unsqueeze(2) + residual
        vote_features = vote_features.reshape(b, num_seed * self.vote_factor, self.out_dim)
        return vote_xyz, vote_features.transpose(2, 1).contiguous()


class _DecodeScores(torch.autograd.Function):
    """decode_scores as one launch each way (include/loss_hip.h votenet_decode_scores): the nine
    named predictions as contiguous (B,K,.) tensors from the head output (B,C,K)."""

    @staticmethod
    def forward(ctx, net, agg_xyz, mean_size, nh, ns, nc):
        _L = _fused_front_end()
        net, agg = net.contiguous(), agg_xyz.contiguous()
        b, _, k = net.shape
        f32 = dict(dtype=torch.float32, device=net.device)
        outs = [torch.empty((b, k, 2), **f32), torch.empty((b, k, 3), **f32), torch.empty((b, k, nh), **f32),
                torch.empty((b, k, nh), **f32), torch.empty((b, k, nh), **f32), torch.empty((b, k, ns), **f32),
                torch.empty((b, k, ns, 3), **f32), torch.empty((b, k, ns, 3), **f32),
                torch.empty((b, k, nc), **f32)]
        with torch.cuda.device(net.device):
            _L.check(_L.lib.votenet_decode_scores(
                b, k, nh, ns, nc, net.data_ptr(), agg.data_ptr(), mean_size.data_ptr(),
                *[o.data_ptr() for o in outs], torch.cuda.current_stream(net.device).cuda_stream),
                "votenet_decode_scores")
        ctx.save_for_backward(net, mean_size)
        ctx.dims = (nh, ns, nc)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        _L = _fused_front_end()
        net, mean_size = ctx.saved_tensors
        nh, ns, nc = ctx.dims
        b, _, k = net.shape
        grads = [g.contiguous() if g is not None else None for g in grads]
        d_net = torch.empty_like(net)
        with torch.cuda.device(net.device):
            _L.check(_L.lib.votenet_decode_scores_grad(
                b, k, nh, ns, nc, net.data_ptr(), mean_size.data_ptr(),
                *[g.data_ptr() if g is not None else None for g in grads], d_net.data_ptr(),
                torch.cuda.current_stream(net.device).cuda_stream), "votenet_decode_scores_grad")
        # (center = aggregated_vote_xyz + offset: the centre's gradient, where one is wanted)
        return d_net, (grads[1] if ctx.needs_input_grad[1] else None), None, None, None, None


def _decode_fused(net, mean_size, width, agg_xyz, num_size_cluster):
    """The one-launch decode takes raw pointers: float32 tensors on net's device, net (B, width, K),
    aggregated_vote_xyz (B, K, 3), mean_size (num_size_cluster, 3) -- anything else (a float64 or
    CPU mean_size, another layout of the vote centres) is the tensor path's."""
    import os
    _L = _fused_front_end()
    return (os.environ.get("VOTENET_FUSED_DECODE", "1") != "0" and net.is_cuda and net.dtype == torch.float32
            and net.dim() == 3 and _L is not None and hasattr(_L.lib, "votenet_decode_scores")
            and net.shape[1] == width
            and mean_size.device == net.device and mean_size.dtype == torch.float32
            and mean_size.is_contiguous() and tuple(mean_size.shape) == (num_size_cluster, 3)
            and torch.is_tensor(agg_xyz) and agg_xyz.device == net.device and agg_xyz.dtype == torch.float32
            and tuple(agg_xyz.shape) == (net.shape[0], net.shape[2], 3))


def decode_scores(net, end_points, num_class, num_heading_bin, num_size_cluster, mean_size):
    """Split the proposal head output (B, C, K) into the named predictions
    (proposal_module.py:24-54); mean_size is a (num_size_cluster, 3) tensor."""
    nh, ns = num_heading_bin, num_size_cluster
    if _decode_fused(net, mean_size, 5 + nh * 2 + ns * 4 + num_class, end_points.get('aggregated_vote_xyz'), ns):
        (objectness, center, heading_scores, hrn, hr, size_scores, srn, sr, sem) = _DecodeScores.apply(
            net, end_points['aggregated_vote_xyz'], mean_size, nh, ns, num_class)
        end_points['objectness_scores'] = objectness
        end_points['center'] = center
        end_points['heading_scores'] = heading_scores
        end_points['heading_residuals_normalized'] = hrn
        end_points['heading_residuals'] = hr
        end_points['size_scores'] = size_scores
        end_points['size_residuals_normalized'] = srn
        end_points['size_residuals'] = sr
        end_points['sem_cls_scores'] = sem
        return end_points
    t = net.transpose(2, 1)
    b, k = t.shape[:2]
    nh, ns = num_heading_bin, num_size_cluster
    # one split instead of seven slices: its backward is a single concatenation (each slice
    # would cost a zero-fill, a copy and an accumulation)
    objectness, offset, heading_scores, hrn, size_scores, srn, sem = torch.split(
        t, [2, 3, nh, nh, ns, ns * 3, t.shape[2] - (5 + nh * 2 + ns * 4)], dim=2)
    end_points['objectness_scores'] = objectness
    end_points['center'] = end_points['aggregated_vote_xyz'] + offset
    end_points['heading_scores'] = heading_scores
    end_points['heading_residuals_normalized'] = hrn  # in [-1, 1]
    end_points['heading_residuals'] = hrn * (np.pi / nh)
    end_points['size_scores'] = size_scores
    srn = F.softplus(srn.reshape(b, k, ns, 3)) - 1
    end_points['size_residuals_normalized'] = srn
    end_points['size_residuals'] = srn * mean_size.unsqueeze(0).unsqueeze(0)
    end_points['sem_cls_scores'] = sem
    return end_points


class ProposalModule(nn.Module):
    """Vote aggregation (one SA layer, r=0.3, ns=16) + proposal head."""

    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal,
                 sampling, seed_feat_dim=256, query_feats='seed'):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_proposal = num_proposal
        self.sampling = sampling
        self.seed_feat_dim = seed_feat_dim
        self.query_feats = query_feats
        self.vote_aggregation = PointnetSAModuleVotes(
            npoint=self.num_proposal, radius=0.3, nsample=16,
            mlp=[self.seed_feat_dim, 128, 128, 128], use_xyz=True, normalize_xyz=True)
        out_width = 2 + 3 + num_heading_bin * 2 + num_size_cluster * 4 + self.num_class
        self.conv1 = nn.Conv1d(128, 128, 1)
        self.conv2 = nn.Conv1d(128, 128, 1)
        self.conv3 = nn.Conv1d(128, out_width, 1)
        self.bn1 = nn.BatchNorm1d(128)
        self.bn2 = nn.BatchNorm1d(128)
        self.register_buffer("_mean_size", torch.from_numpy(mean_size_arr.astype(np.float32)),
                             persistent=False)

    def forward(self, xyz, features, end_points):
        if self.sampling == 'vote_fps':
            xyz, features, sample_inds = self.vote_aggregation(xyz, features)
        elif self.sampling == 'seed_fps':
            # FPS on the seeds, then aggregate the votes of the chosen seeds
            sample_inds = end_points.pop('precomputed_proposal_inds', None)
            if sample_inds is None:
                sample_inds = pointnet2_utils.furthest_point_sample(end_points['seed_xyz'],
                                                                    self.num_proposal)
            xyz, features, _ = self.vote_aggregation(xyz, features, sample_inds)
        elif self.sampling == 'random':
            b, num_seed = end_points['seed_xyz'].shape[:2]
            sample_inds = torch.randint(0, num_seed, (b, self.num_proposal), dtype=torch.int,
                                        device=xyz.device)
            xyz, features, _ = self.vote_aggregation(xyz, features, sample_inds)
        else:
            raise ValueError('Unknown sampling strategy: %s' % (self.sampling,))
        end_points['aggregated_vote_xyz'] = xyz
        end_points['aggregated_vote_inds'] = sample_inds
        net = head_chain(features, self.conv1, self.bn1, self.conv2, self.bn2, self.conv3)
        return decode_scores(net, end_points, self.num_class, self.num_heading_bin,
                             self.num_size_cluster, self._mean_size)


def rot_z(t):
    """(...,) angles -> (..., 3, 3) rotation about the upright axis, the reference's rot_gpu
    (utils/box_util.py:292-306): [[c, s, 0], [-s, c, 0], [0, 0, 1]]."""
    c, s = torch.cos(t), torch.sin(t)
    zero, one = torch.zeros_like(c), torch.ones_like(c)
    return torch.stack([c, s, zero, -s, c, zero, zero, zero, one], dim=-1).view(*t.shape, 3, 3)


def _fused_front_end():
    """the C-ABI library when it has the GridConv front-end kernel (GPU builds), else None"""
    import importlib
    try:
        _L = importlib.import_module("3dioumatch_amd._lib")
    except Exception:  # noqa: BLE001  (CPU-only test environments without the library)
        return None
    return _L if hasattr(_L.lib, "votenet_gridconv_points") else None


class GridConv(nn.Module):
    """IoU branch: a 4x4x4 grid inside each box, features interpolated from the 3 nearest
    seeds, shared MLP, max-pool, IoU head."""

    GRID = 4

    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal,
                 sampling, seed_feat_dim=256, query_feats='seed', iou_class_depend=True):
        super().__init__()
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.num_proposal = num_proposal
        self.sampling = sampling
        self.seed_feat_dim = seed_feat_dim
        self.query_feats = query_feats
        self.iou_class_depend = iou_class_depend
        self.iou_size = num_class if self.iou_class_depend else 1
        self.mlp_before_iou = pt_utils.SharedMLP([self.seed_feat_dim + 3, 128, 128, 128], bn=True)
        self.conv1_iou = nn.Conv1d(128, 128, 1)
        self.conv2_iou = nn.Conv1d(128, 128, 1)
        self.conv3_iou = nn.Conv1d(128, 3 + num_heading_bin * 2 + num_size_cluster * 3 + self.iou_size, 1)
        self.bn1_iou = nn.BatchNorm1d(128)
        self.bn2_iou = nn.BatchNorm1d(128)

    def _unit_grid(self, device):
        """(64, 3) unit grid, x slowest / z fastest (grid_conv_module.py:64-75), made once per device"""
        key = str(device)
        cache = self.__dict__.setdefault("_unit_cache", {})
        if key not in cache:
            # built outside inference mode (the tensor is reused by autograd passes later) and
            # never cached from inside a graph capture (it would live in the graph's private pool)
            with torch.inference_mode(False), torch.no_grad():
                step = torch.linspace(-1, 1, self.GRID, device=device)
                unit = torch.stack(torch.meshgrid(step, step, step, indexing='ij'),
                                   dim=-1).view(self.GRID ** 3, 3).contiguous()
            if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                return unit
            cache[key] = unit
        return cache[key]

    def _origin(self, end_points):
        if self.query_feats == 'vote':
            return end_points['vote_xyz'], end_points['vote_features']
        if self.query_feats == 'seed':
            return end_points['seed_xyz'], end_points['seed_features']
        if self.query_feats == 'seed+vote':
            return end_points['seed_xyz'], end_points['vote_features']
        raise NotImplementedError()

    def forward(self, center, size, heading, end_points):
        origin_xyz, origin_features = self._origin(end_points)
        origin_xyz = origin_xyz.detach().contiguous()
        origin_features = origin_features.detach().contiguous()
        b, k = size.shape[:2]
        g = self.GRID
        g3 = g * g * g
        no_grad_path = not (torch.is_grad_enabled() and
                            (center.requires_grad or size.requires_grad or heading.requires_grad))
        if no_grad_path and origin_features.is_cuda and _fused_front_end() is not None:
            # training / inference without test-time IoU optimisation: grid points + relative
            # coordinates from ONE kernel (written straight into channels 0..2 of the tensor the
            # shared MLP reads), the interpolation weights from one kernel behind three_nn, the
            # interpolation into channels 3.. -- grid_conv_module.py:64-107 was ~20 tensor kernels
            _L = _fused_front_end()
            c = origin_features.shape[1]
            whole = torch.empty((b, k * g3, 3), dtype=torch.float32, device=size.device)
            ctr, sz, hd = center.detach().contiguous(), size.detach().contiguous(), heading.detach().contiguous()
            # the first layer of the shared MLP commutes with the interpolation
            # (SharedMLP.forward_pooled_interp): only the 3 rows of relative coordinates are formed
            # here; the (3 + C, K*64) input tensor is not (a training pass forms it in its backward,
            # for the layer's weight gradient)
            commute = getattr(self.mlp_before_iou, "interp_first_ok", None) is not None and \
                (k * g3) % 4 == 0 and origin_xyz.shape[1] <= 2048
            rows = 3 if commute else 3 + c
            feats = torch.empty((b, rows, k * g3), dtype=torch.float32, device=size.device)
            with torch.cuda.device(size.device):
                _L.check(_L.lib.votenet_gridconv_points(
                    b, k, rows, self._unit_grid(size.device).data_ptr(), ctr.data_ptr(), sz.data_ptr(),
                    hd.data_ptr(), whole.data_ptr(), feats.data_ptr(),
                    torch.cuda.current_stream(size.device).cuda_stream), "votenet_gridconv_points")
            idx, weight = pointnet2_utils.three_nn_with_weights(whole, origin_xyz)
            if commute and self.mlp_before_iou.interp_first_ok(origin_features, idx):
                iou_features = self.mlp_before_iou.forward_pooled_interp(origin_features, idx, weight, feats,
                                                                         k, g3)
            else:
                if commute:  # (the shared MLP declined: the full input after all)
                    rel, feats = feats, torch.empty((b, 3 + c, k * g3), dtype=torch.float32, device=size.device)
                    feats[:, :3].copy_(rel)
                pointnet2_utils._ext.three_interpolate_into(origin_features, idx, weight, feats, 3)
                iou_features = self.mlp_before_iou.forward_pooled(feats.view(b, -1, k, g3))
            net = head_chain(iou_features, self.conv1_iou, self.bn1_iou, self.conv2_iou, self.bn2_iou,
                             self.conv3_iou)
            end_points['iou_scores'] = net.transpose(2, 1)[:, :, -self.iou_size:]
            return end_points
        unit = self._unit_grid(size.device)
        local = unit.view(1, 1, g3, 3) * size.unsqueeze(2)  # (B, K, 64, 3): half-sizes scale it
        # local @ rot_z(heading)^T, written out (a 3x3 rotation about z is two multiply-adds per
        # coordinate; the reference's torch.bmm, grid_conv_module.py:78-79, launches a BLAS kernel)
        cos, sin = torch.cos(heading).view(b, k, 1), torch.sin(heading).view(b, k, 1)
        lx, ly = local[..., 0], local[..., 1]
        whole = torch.stack([lx * cos + ly * sin, ly * cos - lx * sin, local[..., 2]], dim=-1)
        whole = (whole + center.unsqueeze(2)).view(b, k * g3, 3).contiguous()
        relative = whole - center.unsqueeze(2).expand(-1, -1, g3, -1).reshape(b, k * g3, 3)

        dist, idx = pointnet2_utils.three_nn(whole, origin_xyz)
        if torch.is_grad_enabled() and whole.requires_grad:
            # Test-time IoU optimisation (train.py:431-492; VoteNet.forward(iou_opt=True) and
            # forward_onlyiou_faster): d(iou_scores)/d(center, size, heading) also flows through
            # the interpolation WEIGHTS.  three_nn's distances are non-differentiable and the
            # fused interpolation has no weight gradient, so -- exactly like the reference
            # (grid_conv_module.py:87-104) -- the distances are recomputed from the gathered
            # seeds with tensor ops and the weighted sum is a tensor op too.
            c = origin_features.shape[1]
            flat = idx.view(b, -1, 1).long()
            nbr = torch.gather(origin_xyz, 1, flat.expand(-1, -1, 3))          # (B, K*64*3, 3)
            diff = nbr - whole.unsqueeze(2).expand(-1, -1, 3, -1).reshape(b, -1, 3)
            dist = torch.sqrt(torch.sum(diff * diff, dim=2)).view(b, -1, 3)
            weight = 1 / (dist + 1e-8)
            weight = weight / torch.sum(weight, dim=2, keepdim=True)
            picked = torch.gather(origin_features.transpose(1, 2), 1, flat.expand(-1, -1, c))
            interp = torch.sum(picked.view(b, -1, 3, c) * weight.unsqueeze(-1), dim=2)
            interp = interp.transpose(1, 2).contiguous()                       # (B, C, K*64)
        else:
            weight = 1 / (dist + 1e-8)
            weight = (weight / torch.sum(weight, dim=2, keepdim=True)).contiguous()
            into = getattr(pointnet2_utils._ext, "three_interpolate_into", None)
            if into is not None and origin_features.is_cuda:
                # (B, 3 + C, K*64) written once: the interpolation goes straight into channels
                # 3.., no 136 MB intermediate and no concatenation copy (features are detached
                # here, grid_conv_module.py:60-61, so nothing flows back through this)
                feats = torch.empty((b, 3 + origin_features.shape[1], k * g3),
                                    dtype=torch.float32, device=whole.device)
                into(origin_features, idx, weight, feats, 3)
                feats[:, :3].copy_(relative.transpose(1, 2))
                interp = None
                feats = feats.view(b, -1, k, g3)
            else:
                interp = pointnet2_utils.three_interpolate(origin_features, idx, weight)  # (B, C, K*64)
        if interp is not None:
            feats = torch.cat([relative.transpose(1, 2).reshape(b, 3, k, g3),
                               interp.view(b, -1, k, g3)], dim=1)
        iou_features = self.mlp_before_iou.forward_pooled(feats)
        net = head_chain(iou_features, self.conv1_iou, self.bn1_iou, self.conv2_iou, self.bn2_iou,
                         self.conv3_iou)
        end_points['iou_scores'] = net.transpose(2, 1)[:, :, -self.iou_size:]
        return end_points
