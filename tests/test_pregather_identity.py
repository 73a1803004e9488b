"""CPU check of the algebra behind csrc/mlp_pregather.hip, on the oracle's own grouping: a 1x1
convolution applied to the grouped tensor the reference forms (pointnet2_utils.py:335-358:
[(xyz[idx] - new_xyz) * s ; features[idx]]) equals the gather of the convolution of the packed
point-major operand minus the centroid term -- forward, and the three gradient identities the
backward kernels rely on (scatter over idx, minus group sums, two small GEMMs)."""
import numpy as np
import pytest

from oracle.oracle import Oracle


@pytest.mark.parametrize("b,n,c,m,ns,scale", [(2, 300, 5, 40, 8, 1.0), (1, 257, 16, 33, 4, 1.0 / 0.3),
                                              (3, 64, 3, 64, 16, 2.0)])
def test_first_layer_commutes_with_the_gather(b, n, c, m, ns, scale):
    rng = np.random.default_rng(b * 100 + n + c)
    o = Oracle()
    xyz = rng.uniform(-1, 1, (b, n, 3)).astype(np.float32)
    feats = rng.standard_normal((b, c, n)).astype(np.float32)
    inds = np.stack([rng.permutation(n)[:m] for _ in range(b)]).astype(np.int32)
    new_xyz = np.take_along_axis(xyz, inds[..., None].astype(np.int64), axis=1)
    idx = o.ball_query(new_xyz, xyz, 0.6, ns)                                  # (b, m, ns) int32
    w = rng.standard_normal((7, 3 + c)).astype(np.float64)

    # the reference's data flow: group, then convolve
    gx = o.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx)     # (b, 3, m, ns)
    gx = (gx - new_xyz.transpose(0, 2, 1)[..., None]) * np.float32(scale)
    gf = o.group_points(feats, idx)                                            # (b, c, m, ns)
    grouped = np.concatenate([gx, gf], axis=1).astype(np.float64)
    y_ref = np.einsum("ok,bkms->boms", w, grouped)

    # the pre-gather form: src_ext = [xyz*s | new_xyz*s ; feat | 0], z_ext = W . src_ext
    src_ext = np.zeros((b, 3 + c, n + m))
    src_ext[:, :3, :n] = xyz.transpose(0, 2, 1) * scale
    src_ext[:, :3, n:] = new_xyz.transpose(0, 2, 1) * scale
    src_ext[:, 3:, :n] = feats
    z_ext = np.einsum("ok,bkp->bop", w, src_ext)
    li = idx.astype(np.int64)
    y = np.stack([z_ext[i][:, li[i]] for i in range(b)]) - z_ext[:, :, n:][..., None]
    assert np.abs(y - y_ref).max() <= 1e-5 * max(1.0, np.abs(y_ref).max())  # fp32 rel -> double product

    # backward: dz_ext[:, :n] = scatter-add of dy over idx, dz_ext[:, n + j] = -sum_t dy[:, j, t];
    # then dW = dz_ext . src_ext^T and d src_ext = W^T . dz_ext reproduce the reference's gradients
    dy = rng.standard_normal(y.shape)
    dz_ext = np.zeros_like(z_ext)
    for i in range(b):
        np.add.at(dz_ext[i].T, li[i].reshape(-1), dy[i].reshape(7, -1).T)
    dz_ext[:, :, n:] = -dy.sum(axis=3)
    dw = np.einsum("bop,bkp->ok", dz_ext, src_ext)
    dw_ref = np.einsum("boms,bkms->ok", dy, grouped)
    # (the reference rounds xyz[idx] - new_xyz to fp32 before the product: 1e-7 on the xyz columns)
    assert np.abs(dw - dw_ref).max() <= 1e-6 * max(1.0, np.abs(dw_ref).max())
    dsrc = np.einsum("ok,bop->bkp", w, dz_ext)
    dgrouped = np.einsum("ok,boms->bkms", w, dy)
    dfeat_ref = np.zeros((b, c, n))
    dxyz_ref = np.zeros((b, 3, n))
    for i in range(b):
        np.add.at(dfeat_ref[i].T, li[i].reshape(-1), dgrouped[i, 3:].reshape(c, -1).T)
        np.add.at(dxyz_ref[i].T, li[i].reshape(-1), dgrouped[i, :3].reshape(3, -1).T * scale)
    dnew_ref = -dgrouped[:, :3].sum(axis=3) * scale
    assert np.abs(dsrc[:, 3:, :n] - dfeat_ref).max() <= 1e-9 * max(1.0, np.abs(dfeat_ref).max())
    assert np.abs(dsrc[:, :3, :n] * scale - dxyz_ref).max() <= 1e-9 * max(1.0, np.abs(dxyz_ref).max())
    assert np.abs(dsrc[:, :3, n:] * scale - dnew_ref).max() <= 1e-9 * max(1.0, np.abs(dnew_ref).max())
