"""PointNet++ backbone of VoteNet (4 set-abstraction + 2 feature-propagation layers).

Host-side mirror of the reference models/backbone_module.py (hyper-parameters :35-72, forward
:105-133): same attribute names (sa1..sa4, fp1, fp2 -> same state_dict keys), same end_points
keys, int32 index tensors.  Every custom operator underneath is a gfx950 HIP kernel.
"""
import torch
import torch.nn as nn

from pointnet2.pointnet2_modules import PointnetFPModule, PointnetSAModuleVotes

# (npoint, radius, nsample, mlp widths after the input width) -- backbone_module.py:38-69
SA_SPECS = (
    (2048, 0.2, 64, (64, 64, 128)),
    (1024, 0.4, 32, (128, 128, 256)),
    (512, 0.8, 16, (128, 128, 256)),
    (256, 1.2, 16, (128, 128, 256)),
)


class Pointnet2Backbone(nn.Module):
    """input (B, N, 3 + input_feature_dim) -> end_points with sa{1..4}_{xyz,features,inds},
    fp2_{features,xyz,inds}; fp2 = 1024 seeds with 256 channels."""

    def __init__(self, input_feature_dim=0):
        super().__init__()
        width = input_feature_dim
        for i, (npoint, radius, nsample, mlp) in enumerate(SA_SPECS, start=1):
            setattr(self, "sa%d" % i, PointnetSAModuleVotes(
                npoint=npoint, radius=radius, nsample=nsample, mlp=[width, *mlp],
                use_xyz=True, normalize_xyz=True))
            width = mlp[-1]
        self.fp1 = PointnetFPModule(mlp=[256 + 256, 256, 256])
        self.fp2 = PointnetFPModule(mlp=[256 + 256, 256, 256])

    @staticmethod
    def _break_up_pc(pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    @torch.no_grad()
    def compute_geometry(self, pointcloud):
        """Everything in the backbone that depends on the point COORDINATES only (not on any
        learned weight): the four furthest-point samplings and the four ball queries.  Being data-only, it can be computed
        for the NEXT batch on a side stream while the current step trains (votenet/step.py),
        which takes the strictly serial FPS rounds (a handful of busy CUs) off the critical
        path.  Returns {"sa<i>_inds", "sa<i>_ball_idx"} (int32), the centroid coordinates
        "sa<i>_new_xyz" and the interpolation ("fp<j>_idx", "fp<j>_weight") of the two FP layers.
        The first layer's features are network INPUTS as well, so on large clouds its whole
        grouped tensor "sa1_grouped" (B, 3+C, npoint, nsample) comes out of the same ONE kernel
        that answers its ball queries (the north-star pair of bench.py's roofline)."""
        from pointnet2 import pointnet2_utils
        xyz, in_features = self._break_up_pc(pointcloud)
        geometry = {}
        first_tie = None
        for i in range(1, 5):
            sa = getattr(self, "sa%d" % i)
            # (large clouds: the sampling kernel leaves the cloud's cell lists behind and the
            #  ball query runs on them -- no separate cell-list build in the chain.  Layers 2..4
            #  sample the centroids of the layer before, in pick order: the reference's kernel
            #  finds 0, 1, 2, ... again unless the first run met an exact tie, which it records)
            inds, lists, first_tie = pointnet2_utils.sample_chain(
                xyz, sa.npoint, sa.radius, first_tie, head=i > 1 and first_tie is not None)
            new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(),
                                                       inds).transpose(1, 2).contiguous()
            geometry["sa%d_inds" % i] = inds
            geometry["sa%d_new_xyz" % i] = new_xyz
            if lists is not None:  # the centroids are this sampling call's picks: its query plans apply
                lists.mark_centroids(new_xyz, inds)
            if i == 1 and lists is not None and sa.nsample <= 256 and sa.use_xyz and \
                    sa.pooling == 'max' and (in_features is None or in_features.shape[1] <= 8):
                idx, grouped = pointnet2_utils._ext.query_and_group(
                    new_xyz, xyz, in_features, sa.radius, sa.nsample, sa.normalize_xyz, None, lists)
                geometry["sa1_ball_idx"], geometry["sa1_grouped"] = idx, grouped
            elif lists is not None and sa.nsample <= 256:
                geometry["sa%d_ball_idx" % i] = pointnet2_utils._ext.ball_query_prebuilt(
                    new_xyz, xyz, sa.radius, sa.nsample, lists)
            else:
                geometry["sa%d_ball_idx" % i] = pointnet2_utils.ball_query(sa.radius, sa.nsample,
                                                                           xyz, new_xyz)
            # inverse index for the backward scatter-add of the layer's feature gradient (the
            # first layer's features are network inputs: no gradient, no inverse)
            make_inv = getattr(pointnet2_utils._ext, "group_inverse", None)
            if i > 1 and make_inv is not None and xyz.is_cuda:
                inv = make_inv(geometry["sa%d_ball_idx" % i], xyz.shape[1])
                if inv is not None:
                    geometry["sa%d_ball_inv" % i] = inv
            xyz = new_xyz
        # the two feature-propagation layers interpolate sa4 -> sa3 and sa3 -> sa2
        for name, unknown, known in (("fp1", "sa3", "sa4"), ("fp2", "sa2", "sa3")):
            idx, weight = self.fp1.interpolation(geometry[unknown + "_new_xyz"],
                                                 geometry[known + "_new_xyz"])
            geometry[name + "_idx"], geometry[name + "_weight"] = idx, weight.contiguous()
        return geometry

    def forward(self, pointcloud, end_points=None, geometry=None):
        end_points = end_points if end_points else {}
        xyz, features = self._break_up_pc(pointcloud)
        for i in range(1, 5):
            given = geometry["sa%d_inds" % i] if geometry is not None else None
            ball = geometry.get("sa%d_ball_idx" % i) if geometry is not None else None
            centroids = geometry.get("sa%d_new_xyz" % i) if geometry is not None else None
            inverse = geometry.get("sa%d_ball_inv" % i) if geometry is not None else None
            grouped = geometry.get("sa%d_grouped" % i) if geometry is not None else None
            xyz, features, inds = getattr(self, "sa%d" % i)(xyz, features, given, ball, centroids,
                                                            inverse, grouped)
            end_points["sa%d_xyz" % i] = xyz
            end_points["sa%d_features" % i] = features
            if i <= 2:
                end_points["sa%d_inds" % i] = inds
        interp = [None, None]
        if geometry is not None and "fp1_idx" in geometry:
            interp = [(geometry["fp%d_idx" % j], geometry["fp%d_weight" % j]) for j in (1, 2)]
        features = self.fp1(end_points["sa3_xyz"], end_points["sa4_xyz"],
                            end_points["sa3_features"], end_points["sa4_features"], interp[0])
        features = self.fp2(end_points["sa2_xyz"], end_points["sa3_xyz"],
                            end_points["sa2_features"], features, interp[1])
        end_points["fp2_features"] = features
        end_points["fp2_xyz"] = end_points["sa2_xyz"]
        num_seed = end_points["fp2_xyz"].shape[1]
        # seeds are the first num_seed FPS picks of SA1: indices into the input cloud
        end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:num_seed]
        return end_points
