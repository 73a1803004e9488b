"""VoteNet with the IoU-estimation branch (3DIoUMatch's detector).

Host-side mirror of the reference models/votenet_iou_branch.py:24-188: same constructor
arguments, sub-module names (backbone_net, vgen, pnet, grid_conv -> state_dict keys), the
same three forwards (forward, forward_with_pred_jitter, forward_onlyiou_faster) and the same
end_points keys.  `forward(inputs, mode=...)` additionally routes the custom forwards through
nn.Module.__call__, so DistributedDataParallel hooks see them (the reference calls the custom
method directly on the wrapper, which breaks under DataParallel/DDP -- SURVEY section 0).
"""
import numpy as np
import torch
import torch.nn as nn

from .backbone import Pointnet2Backbone
from .heads import GridConv, ProposalModule, VotingModule



class _UnitLength(torch.autograd.Function):
    """features / ||features||_2 over the channel axis (models/votenet_iou_branch.py:103-104) as
    one kernel each way (votenet_channel_normalize[_grad]) instead of ~30 tensor kernels."""

    @staticmethod
    def forward(ctx, x):
        import importlib
        _L = importlib.import_module("3dioumatch_amd._lib")
        x = x.contiguous()
        b, c, n = x.shape
        y = torch.empty_like(x)
        norm = torch.empty((b, n), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _L.check(_L.lib.votenet_channel_normalize(b, c, n, x.data_ptr(), y.data_ptr(), norm.data_ptr(),
                                                      torch.cuda.current_stream(x.device).cuda_stream),
                     "votenet_channel_normalize")
        ctx.save_for_backward(y, norm)
        return y

    @staticmethod
    def backward(ctx, dy):
        import importlib
        _L = importlib.import_module("3dioumatch_amd._lib")
        y, norm = ctx.saved_tensors
        dy = dy.contiguous()
        b, c, n = y.shape
        dx = torch.empty_like(y)
        with torch.cuda.device(y.device):
            _L.check(_L.lib.votenet_channel_normalize_grad(b, c, n, y.data_ptr(), norm.data_ptr(),
                                                           dy.data_ptr(), dx.data_ptr(),
                                                           torch.cuda.current_stream(y.device).cuda_stream),
                     "votenet_channel_normalize_grad")
        return dx


def unit_length_features(features):
    if features.is_cuda and features.dtype == torch.float32 and features.dim() == 3:
        return _UnitLength.apply(features)
    return features.div(torch.norm(features, p=2, dim=1).unsqueeze(1))


class VoteNet(nn.Module):
    def __init__(self, num_class, num_heading_bin, num_size_cluster, mean_size_arr, dataset_config,
                 input_feature_dim=0, num_proposal=128, vote_factor=1, sampling='vote_fps',
                 query_feats='seed'):
        super().__init__()
        assert mean_size_arr.shape[0] == num_size_cluster
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        self.mean_size_arr = mean_size_arr
        self.dataset_config = dataset_config
        self.input_feature_dim = input_feature_dim
        self.num_proposal = num_proposal
        self.vote_factor = vote_factor
        self.sampling = sampling
        self.backbone_net = Pointnet2Backbone(input_feature_dim=self.input_feature_dim)
        self.vgen = VotingModule(self.vote_factor, 256)
        self.pnet = ProposalModule(num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                                   num_proposal, sampling, query_feats=query_feats)
        self.grid_conv = GridConv(num_class, num_heading_bin, num_size_cluster, mean_size_arr,
                                  num_proposal, sampling, query_feats=query_feats)
        self.register_buffer("_mean_size", torch.from_numpy(mean_size_arr.astype(np.float32)),
                             persistent=False)

    def forward_backbone(self, inputs):
        end_points = self.backbone_net(inputs['point_clouds'], {}, inputs.get('geometry'))
        xyz, features = end_points['fp2_xyz'], end_points['fp2_features']
        end_points['seed_inds'] = end_points['fp2_inds']
        end_points['seed_xyz'] = xyz
        end_points['seed_features'] = features
        xyz, features = self.vgen(xyz, features)
        features = unit_length_features(features)
        end_points['vote_xyz'] = xyz
        end_points['vote_features'] = features
        geometry = inputs.get('geometry')
        if geometry is not None and geometry.get('proposal_inds') is not None:
            end_points['precomputed_proposal_inds'] = geometry['proposal_inds']
        return self.pnet(xyz, features, end_points)

    @torch.no_grad()
    def compute_geometry(self, inputs):
        """Coordinate-only index computations of one batch (backbone FPS x4 and, for
        sampling == 'seed_fps', the proposal FPS on the seeds = SA2 centroids)."""
        from pointnet2 import pointnet2_utils
        pc = inputs['point_clouds']
        geometry = self.backbone_net.compute_geometry(pc)
        if self.sampling == 'seed_fps':
            xyz = pc[..., 0:3].contiguous()
            for key in ("sa1_inds", "sa2_inds"):
                xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(),
                                                       geometry[key]).transpose(1, 2).contiguous()
            geometry['proposal_inds'] = pointnet2_utils.furthest_point_sample(xyz,
                                                                              self.num_proposal)
        return geometry

    def calculate_bbox(self, end_points):
        """arg-max size/heading class -> (center, HALF size, heading); votenet_iou_branch.py:111-137"""
        size_scores, size_residuals = end_points['size_scores'], end_points['size_residuals']
        b, k = size_scores.shape[:2]
        size_class = torch.argmax(size_scores, -1)
        size_residual = torch.gather(
            size_residuals, 2, size_class.view(b, k, 1, 1).expand(-1, -1, -1, 3)).squeeze(2)
        size_base = torch.index_select(self._mean_size, 0, size_class.view(-1)).view(b, k, 3)
        size = (size_base + size_residual) / 2
        size = torch.where(size < 0, torch.full_like(size, 1e-6), size)
        heading_class = torch.argmax(end_points['heading_scores'], -1)
        heading_residual = torch.gather(end_points['heading_residuals'], 2,
                                        heading_class.unsqueeze(-1)).squeeze(2)
        heading = self.dataset_config.class2angle_gpu(heading_class, heading_residual)
        end_points['size'] = size
        end_points['heading'] = heading
        return end_points['center'], size, heading

    def forward(self, inputs, iou_opt=False, mode='plain'):
        if mode == 'jitter':
            return self.forward_with_pred_jitter(inputs)
        end_points = self.forward_backbone(inputs)
        center, size, heading = self.calculate_bbox(end_points)
        if iou_opt:
            center.retain_grad()
            size.retain_grad()
            if heading.requires_grad:
                heading.retain_grad()
            return self.grid_conv(center, size, heading, end_points)
        return self.grid_conv(center.detach(), size.detach(), heading.detach(), end_points)

    def forward_with_pred_jitter(self, inputs):
        """Predicted boxes + one jittered copy of each go through the IoU branch together
        (votenet_iou_branch.py:157-181): K -> 2K boxes."""
        end_points = self.forward_backbone(inputs)
        # inputs['jitter_noise']: the two standard-normal (B, K, 3) tensors of this pass, drawn by
        # the caller (a step runner that replays two forward graphs side by side draws them ahead
        # of both: graph replays share one device-side offset per generator, votenet/step.py)
        noise = inputs.get('jitter_noise')
        fused = self._bbox_jitter_fused(end_points, noise)
        if fused is not None:
            return fused
        center, size, heading = self.calculate_bbox(end_points)
        k = heading.shape[1]
        if noise is None:  # the two draws of the reference, in its order
            noise = (torch.randn(size.shape, device=size.device),
                     torch.randn(size.shape, device=size.device))
        center_jitter = center + size * noise[0] * 0.3
        size_jitter = size + size * noise[1] * 0.3
        size_jitter = torch.clamp(size_jitter, min=1e-8)
        heading_jitter = heading.clone()
        all_center = torch.cat([center, center_jitter], dim=1)
        all_size = torch.cat([size, size_jitter], dim=1)
        all_heading = torch.cat([heading, heading_jitter], dim=1)
        end_points = self.grid_conv(all_center.detach(), all_size.detach(), all_heading.detach(),
                                    end_points)
        end_points['iou_scores'], end_points['iou_scores_jitter'] = torch.split(
            end_points['iou_scores'], [k, end_points['iou_scores'].shape[1] - k], dim=1)
        end_points['jitter_center'] = center_jitter
        end_points['jitter_size'] = size_jitter * 2
        end_points['jitter_heading'] = heading_jitter
        return end_points

    def _bbox_jitter_fused(self, end_points, noise=None):
        """calculate_bbox + the jitter as ONE kernel (votenet_bbox_jitter) on the GPU: nothing flows
        back through these tensors in the training forward (they feed the detached IoU branch and
        the IoU labels).  Returns the completed end_points, or None where the kernel is not there."""
        center = end_points['center']
        if not center.is_cuda:
            return None
        cfg = self.dataset_config
        from .config import DatasetConfig
        if not getattr(cfg, 'fused_heading_decode', False) or \
                type(cfg).class2angle_gpu is not DatasetConfig.class2angle_gpu:
            return None  # a custom heading decoding: the tensor-op path calls it
        from .heads import _fused_front_end
        _L = _fused_front_end()
        if _L is None or not hasattr(_L.lib, "votenet_bbox_jitter"):
            return None
        size_scores, size_residuals = end_points['size_scores'], end_points['size_residuals']
        b, k, ns = size_scores.shape
        nh = self.num_heading_bin
        dev = center.device
        # the two draws of the reference, in its order (votenet_iou_branch.py:161-162)
        if noise is None:
            noise_c = torch.randn((b, k, 3), device=dev)
            noise_s = torch.randn((b, k, 3), device=dev)
        else:
            noise_c, noise_s = noise
            for t in (noise_c, noise_s):
                if t.shape != (b, k, 3) or t.dtype != torch.float32 or t.device != dev or \
                        not t.is_contiguous():
                    raise ValueError("jitter_noise: two contiguous float32 (%d, %d, 3) tensors on %s"
                                     % (b, k, dev))
        c = center.detach().contiguous()
        ss, sr = size_scores.detach().contiguous(), size_residuals.detach().contiguous()
        hs = end_points['heading_scores'].detach().contiguous()
        hr = end_points['heading_residuals'].detach().contiguous()
        size = torch.empty((b, k, 3), dtype=torch.float32, device=dev)
        heading = torch.empty((b, k), dtype=torch.float32, device=dev)
        all_center = torch.empty((b, 2 * k, 3), dtype=torch.float32, device=dev)
        all_size = torch.empty((b, 2 * k, 3), dtype=torch.float32, device=dev)
        all_heading = torch.empty((b, 2 * k), dtype=torch.float32, device=dev)
        jitter_size2 = torch.empty((b, k, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _L.check(_L.lib.votenet_bbox_jitter(
                b, k, ns, nh, c.data_ptr(), ss.data_ptr(), sr.data_ptr(), hs.data_ptr(), hr.data_ptr(),
                self._mean_size.data_ptr(), noise_c.data_ptr(), noise_s.data_ptr(), size.data_ptr(),
                heading.data_ptr(), all_center.data_ptr(), all_size.data_ptr(), all_heading.data_ptr(),
                jitter_size2.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                "votenet_bbox_jitter")
        end_points['size'], end_points['heading'] = size, heading
        end_points = self.grid_conv(all_center, all_size, all_heading, end_points)
        end_points['iou_scores'], end_points['iou_scores_jitter'] = torch.split(
            end_points['iou_scores'], [k, end_points['iou_scores'].shape[1] - k], dim=1)
        end_points['jitter_center'] = all_center[:, k:]
        end_points['jitter_size'] = jitter_size2
        end_points['jitter_heading'] = all_heading[:, k:]
        return end_points

    def forward_onlyiou_faster(self, end_points, center, size, heading):
        return self.grid_conv(center, size, heading, end_points)
