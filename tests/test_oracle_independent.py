"""CPU tests: a SECOND derivation of every pointnet2 operator, written from the reference's
source semantics with different machinery (numpy fancy indexing, np.add.at, stable argsort,
masked argmax) than oracle/pn2_oracle.c (literal C loops), and a hypothesis-driven differential
test oracle <-> numpy over random shapes, radii, duplicates and near-origin points.

Why: the reference has no CPU implementation and no tests of these ops (ball_query.cpp:33,
sampling.cpp:87, interpolate.cpp:41 raise "CPU not supported"), so the restatement cannot be
pinned to reference outputs in this container; two independent derivations that agree bit for
bit are the strongest pin available (oracle/README.md lists which functions have one).

  furthest_point_sampling : sampling_gpu.cu:94-177 (skip |p|^2 <= 1e-3, running min, tree tie order)
  gather_points(+grad)    : sampling_gpu.cu:13-62
  ball_query              : ball_query_gpu.cu:14-49
  group_points(+grad)     : group_points_gpu.cu:13-80
  three_nn                : interpolate_gpu.cu:14-64
  three_interpolate(+grad): interpolate_gpu.cu:77-159 (intended scatter-add for the gradient)
"""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _bitrev(v, nbits):
    r = 0
    for _ in range(nbits):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


def sq(a, b):
    """the reference's squared distance, fp32, left to right"""
    d = (a - b).astype(np.float32)
    return ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]).astype(np.float32)


# ------------------------------------------------------------------ numpy derivations
def np_fps(oracle, xyz, m):
    out = np.zeros((xyz.shape[0], m), np.int32)
    for b, p in enumerate(xyz):
        n = p.shape[0]
        bs = int(oracle.lib.pn2o_opt_n_threads(n))
        lb = bs.bit_length() - 1
        mag = ((p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]).astype(np.float32) + p[:, 2] * p[:, 2]).astype(np.float32)
        skip = mag.astype(np.float64) <= 1e-3
        key = np.array([(_bitrev(k % bs, lb) << 22) | k for k in range(n)], np.int64)
        temp = np.full(n, 1e10, np.float32)
        old = 0
        for j in range(1, m):
            temp = np.where(skip, temp, np.minimum(sq(p, p[old]), temp))
            cand = np.where(skip, np.float32(-1), temp)
            best = cand.max()
            if best < 0:
                old = 0
            else:
                ties = np.nonzero(cand == best)[0]
                old = int(ties[np.argmin(key[ties])])
            out[b, j] = old
    return out


def np_ball_query(new_xyz, xyz, radius, nsample):
    b, m, _ = new_xyz.shape
    r2 = np.float32(radius) * np.float32(radius)
    idx = np.zeros((b, m, nsample), np.int32)
    for bi in range(b):
        d2 = sq(new_xyz[bi][:, None, :], xyz[bi][None, :, :])          # (m, n)
        for j in range(m):
            hits = np.flatnonzero(d2[j] < r2)[:nsample]
            if hits.size:
                idx[bi, j] = hits[0]
                idx[bi, j, :hits.size] = hits
    return idx


def np_three_nn(unknown, known):
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.full((b, n, 3), 0, np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    for bi in range(b):
        d = sq(unknown[bi][:, None, :], known[bi][None, :, :]).astype(np.float64)  # exact in double
        order = np.argsort(d, axis=1, kind="stable")[:, :3]                       # first index wins ties
        k = order.shape[1]
        best = np.take_along_axis(d, order, axis=1)
        full = np.full((n, 3), 1e40)
        full[:, :k] = best
        ii = np.zeros((n, 3), np.int32)
        ii[:, :k] = order
        with np.errstate(over='ignore'):   # 1e40 (fewer than three known points) -> inf, as in C
            dist2[bi] = full.astype(np.float32)
        idx[bi] = ii
    return dist2, idx


# ------------------------------------------------------------------ fixed cases
@pytest.mark.parametrize("n,m,seed", [(700, 64, 1), (64, 64, 2), (300, 17, 3)])
def test_fps_second_derivation(oracle, synth, n, m, seed):
    xyz = synth.cloud_edge_cases(2, n, 1.3, seed=seed, near_origin=5, duplicates=min(30, n // 10))
    assert np.array_equal(oracle.furthest_point_sampling(xyz, m), np_fps(oracle, xyz, m))


def test_gather_and_grad_second_derivation(oracle):
    g = np.random.default_rng(3)
    b, c, n, m = 2, 5, 90, 40
    pts = g.standard_normal((b, c, n)).astype(np.float32)
    idx = g.integers(0, n, (b, m)).astype(np.int32)
    idx[:, :7] = idx[:, 7:14]                                   # repeated indices: sums in the gradient
    want = np.stack([pts[bi][:, idx[bi]] for bi in range(b)])
    assert np.array_equal(bits(oracle.gather_points(pts, idx)), bits(want))
    go = g.standard_normal((b, c, m)).astype(np.float32)
    grad = np.zeros((b, c, n), np.float64)
    for bi in range(b):
        for ch in range(c):
            np.add.at(grad[bi, ch], idx[bi], go[bi, ch].astype(np.float64))
    got = oracle.gather_points_grad(go, idx, n)
    assert np.allclose(got, grad, rtol=0, atol=1e-5)
    untouched = np.ones((b, n), bool)
    for bi in range(b):
        untouched[bi, idx[bi]] = False
    assert np.all(got[np.broadcast_to(untouched[:, None, :], got.shape)] == 0)


def test_group_and_grad_second_derivation(oracle, synth):
    g = np.random.default_rng(4)
    b, c, n, m, ns = 2, 6, 300, 25, 9
    xyz = synth.cloud_uniform(b, n, 1.0, seed=9)
    cen = xyz[:, g.permutation(n)[:m]].copy()
    idx = np_ball_query(cen, xyz, 0.25, ns)
    assert np.array_equal(idx, oracle.ball_query(cen, xyz, 0.25, ns))
    feats = g.standard_normal((b, c, n)).astype(np.float32)
    want = np.stack([feats[bi][:, idx[bi]] for bi in range(b)])       # (b, c, m, ns) by fancy indexing
    assert np.array_equal(bits(oracle.group_points(feats, idx)), bits(want))
    go = g.standard_normal((b, c, m, ns)).astype(np.float32)
    grad = np.zeros((b, c, n), np.float64)
    for bi in range(b):
        for ch in range(c):
            np.add.at(grad[bi, ch], idx[bi].reshape(-1), go[bi, ch].reshape(-1).astype(np.float64))
    assert np.allclose(oracle.group_points_grad(go, idx, n), grad, rtol=0, atol=1e-4)


def test_three_interpolate_and_grad_second_derivation(oracle, synth):
    g = np.random.default_rng(5)
    b, c, m, n = 2, 7, 40, 130
    known = synth.cloud_uniform(b, m, 1.0, seed=2)
    unknown = synth.cloud_uniform(b, n, 1.0, seed=3)
    unknown[:, :5] = known[:, :5]                                  # zero distances
    d2, idx = oracle.three_nn(unknown, known)
    wd2, widx = np_three_nn(unknown, known)
    assert np.array_equal(idx, widx) and np.array_equal(bits(d2), bits(wd2))
    w = g.random((b, n, 3)).astype(np.float32)
    pts = g.standard_normal((b, c, m)).astype(np.float32)
    want = np.zeros((b, c, n), np.float32)
    for bi in range(b):
        p = pts[bi][:, idx[bi]]                                     # (c, n, 3)
        # interpolate_gpu.cu:103-105: p0*w0 + p1*w1 + p2*w2, left to right in fp32
        want[bi] = ((p[..., 0] * w[bi][None, :, 0]).astype(np.float32) +
                    (p[..., 1] * w[bi][None, :, 1]).astype(np.float32)).astype(np.float32) + \
            (p[..., 2] * w[bi][None, :, 2]).astype(np.float32)
    assert np.allclose(oracle.three_interpolate(pts, idx, w), want, rtol=0, atol=2e-6)
    go = g.standard_normal((b, c, n)).astype(np.float32)
    grad = np.zeros((b, c, m), np.float64)
    for bi in range(b):
        for ch in range(c):
            for q in range(3):
                np.add.at(grad[bi, ch], idx[bi][:, q], (go[bi, ch] * w[bi][:, q]).astype(np.float64))
    assert np.allclose(oracle.three_interpolate_grad(go, idx, w, m), grad, rtol=0, atol=1e-4)


# ------------------------------------------------------------------ differential, hypothesis
@st.composite
def clouds(draw):
    n = draw(st.integers(1, 220))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    scale = draw(st.sampled_from([0.02, 0.3, 1.0, 7.0]))
    g = np.random.default_rng(seed)
    p = (g.random((1, n, 3), dtype=np.float32) - np.float32(draw(st.sampled_from([0.0, 0.5])))) * np.float32(scale)
    for _ in range(draw(st.integers(0, 3))):                      # exact duplicates
        i, j = g.integers(0, n, 2)
        p[0, i] = p[0, j]
    for _ in range(draw(st.integers(0, 3))):                      # near-origin points (the FPS skip)
        p[0, g.integers(0, n)] = (g.random(3, dtype=np.float32) - 0.5) * np.float32(0.02)
    return p


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(xyz=clouds(), frac=st.floats(0.05, 1.0), rsel=st.sampled_from([0.0, 1e-3, 0.05, 0.3, 2.0, 50.0]),
       ns=st.integers(1, 40), seed=st.integers(0, 1000))
def test_differential_oracle_vs_numpy(oracle, xyz, frac, rsel, ns, seed):
    n = xyz.shape[1]
    m = max(1, int(round(frac * n)))
    inds = oracle.furthest_point_sampling(xyz, m)
    assert np.array_equal(inds, np_fps(oracle, xyz, m))
    cen = np.take_along_axis(xyz, inds[..., None].astype(np.int64), axis=1)
    idx = oracle.ball_query(cen, xyz, rsel, ns)
    assert np.array_equal(idx, np_ball_query(cen, xyz, rsel, ns))
    g = np.random.default_rng(seed)
    feats = g.standard_normal((1, 3, n)).astype(np.float32)
    assert np.array_equal(bits(oracle.group_points(feats, idx)), bits(feats[0][:, idx[0]][None]))
    d2, nn = oracle.three_nn(xyz, cen)
    wd2, wnn = np_three_nn(xyz, cen)
    assert np.array_equal(nn, wnn) and np.array_equal(bits(d2), bits(wd2))
