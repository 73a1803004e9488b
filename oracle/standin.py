"""TEST INFRASTRUCTURE: a CPU stand-in for `pointnet2._ext` backed by the oracle.

Used (a) by tests/golden/make_*_golden.py to drive the REFERENCE's Python layers on the
CPU (the reference extension is CUDA-only), and (b) by the CPU tests of this repository's
host-side mirror modules (their autograd glue and module composition) and by bench.py cpu_baseline, so that host logic is
covered without a GPU.  Never imported by the product.
"""
import types

import numpy as np
import torch


def make(oracle):
    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def n(x):
        return x.detach().cpu().numpy()

    m = types.ModuleType("oracle_backed_ext")
    m.gather_points = lambda points, idx: t(oracle.gather_points(n(points), n(idx)))
    m.gather_points_grad = lambda g, idx, nn: t(oracle.gather_points_grad(n(g), n(idx), nn))
    m.furthest_point_sampling = lambda p, k: t(oracle.furthest_point_sampling(n(p), k))

    def three_nn(u, k):
        d2, i = oracle.three_nn(n(u), n(k))
        return [t(d2), t(i)]

    m.three_nn = three_nn
    m.three_interpolate = lambda p, i, w: t(oracle.three_interpolate(n(p), n(i), n(w)))
    # the TRUE gradient (the reference's dispatch bug is documented separately)
    m.three_interpolate_grad = lambda g, i, w, mm: t(
        oracle.three_interpolate_grad(n(g), n(i), n(w), mm))
    m.ball_query = lambda new_xyz, xyz, r, ns: t(oracle.ball_query(n(new_xyz), n(xyz), r, ns))
    m.group_points = lambda p, i: t(oracle.group_points(n(p), n(i)))
    m.group_points_grad = lambda g, i, nn: t(oracle.group_points_grad(n(g), n(i), nn))

    def query_and_group(new_xyz, xyz, features, radius, nsample, normalize_xyz, idx=None):
        # the reference composition, pointnet2_utils.py:335-358
        if idx is None:
            idx = m.ball_query(new_xyz, xyz, radius, nsample)
        gx = m.group_points(xyz.detach().transpose(1, 2).contiguous(), idx)
        gx = gx - new_xyz.detach().transpose(1, 2).unsqueeze(-1)
        if normalize_xyz:
            gx = gx / radius
        if features is not None:
            gx = torch.cat([gx, m.group_points(features.detach().contiguous(), idx)], dim=1)
        return idx, gx

    m.query_and_group = query_and_group
    return m
