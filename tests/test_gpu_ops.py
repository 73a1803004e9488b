"""GPU parity tests (run with -m gpu on an MI355X): every HIP operator, called through the
C ABI by the drop-in modules, against the CPU oracle on the same seeded inputs and against the
committed golden vectors.

Bars (BASELINE.json north_star): indices bit-exact for FPS / ball_query / three_nn; gathers
bit-exact (pure copies); IoU, interpolated features and the atomic scatter-adds within 1e-4
(tolerance stated in each test).
"""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------- FPS
@pytest.mark.parametrize("key", ["U", "E", "n100", "n600", "n1500", "allskip"])
def test_fps_golden(ext, key):
    g = golden("ops_fps.npz")
    got = ext.furthest_point_sampling(dev(g["xyz_" + key]), int(g["m_" + key]))
    assert got.dtype == torch.int32
    assert np.array_equal(got.cpu().numpy(), g["idx_" + key])


@pytest.mark.parametrize("n,m", [(1, 1), (2, 2), (63, 10), (64, 64), (511, 60), (512, 100),
                                 (513, 100), (1024, 128), (2048, 256), (3000, 200),
                                 (5000, 128), (9000, 96), (16384, 64), (20000, 48),
                                 (40000, 40), (65536, 32), (66000, 32), (80000, 24)])
def test_fps_sizes_vs_oracle(ext, oracle, synth, n, m):
    # exercises every kernel instantiation (register tiers, the bucketed tiers up to 65536 points
    # and the streaming tier beyond) with duplicates (ties) and near-origin (skipped) points present
    xyz = synth.cloud_edge_cases(2, n, 1.0, seed=100 + n, near_origin=min(4, n // 8),
                                 duplicates=min(32, n // 8)) if n >= 16 else \
        synth.cloud_uniform(2, n, 1.0, seed=n)
    want = oracle.furthest_point_sampling(xyz, m)
    got = ext.furthest_point_sampling(dev(xyz), m).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shift", [0.3, 1.1, 5.0])
def test_fps_negative_coordinates(ext, oracle_omp, synth, shift):
    """Clouds that straddle / lie below the origin (real scans are centred arbitrarily) on the
    bucketed tier: the per-bucket bounding boxes prune correctly for negative coordinates
    (round-1 regression: a mis-compiled wave minimum made them too tight there)."""
    xyz = synth.cloud_uniform(2, 12000, 2.2, seed=41) - np.float32(shift)
    want = oracle_omp.furthest_point_sampling(xyz, 600)
    assert np.array_equal(ext.furthest_point_sampling(dev(xyz), 600).cpu().numpy(), want)
    room = synth.cloud_room(2, 40000, seed=3) - np.array([3.0, 2.5, 1.0], np.float32)
    want = oracle_omp.furthest_point_sampling(room, 1024)
    assert np.array_equal(ext.furthest_point_sampling(dev(room), 1024).cpu().numpy(), want)


def test_fps_all_ties(ext, oracle):
    # every point identical: the winner is decided purely by the reduction-tree order
    xyz = np.ones((1, 1024, 3), np.float32)
    want = oracle.furthest_point_sampling(xyz, 16)
    assert np.array_equal(ext.furthest_point_sampling(dev(xyz), 16).cpu().numpy(), want)
    # two far clusters of exact duplicates
    xyz = np.ones((1, 2000, 3), np.float32)
    xyz[0, 1000:] = 5.0
    want = oracle.furthest_point_sampling(xyz, 32)
    assert np.array_equal(ext.furthest_point_sampling(dev(xyz), 32).cpu().numpy(), want)


def test_fps_config2_properties(ext, synth):
    """BASELINE config 2 size (B=8, N=40000 -> 2048): size-independent properties -- index 0
    first, all indices distinct and in range, and the defining greedy property checked on a
    strided subset of rounds with an fp32 re-computation."""
    xyz = synth.cloud_uniform(8, 40000, synth.cube_side(40000, 0.2, 64), seed=1)
    idx = ext.furthest_point_sampling(dev(xyz), 2048).cpu().numpy()
    assert idx.shape == (8, 2048) and np.all(idx[:, 0] == 0)
    assert idx.min() >= 0 and idx.max() < 40000
    for b in range(8):
        assert len(np.unique(idx[b])) == 2048
    b = 3
    p = xyz[b]
    temp = np.full(40000, 1e10, np.float32)
    for j in range(1, 40):
        d = p - p[idx[b, j - 1]]
        dd = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        temp = np.minimum(temp, dd)
        assert temp[idx[b, j]] == temp.max()


# --------------------------------------------------------------------------- gather / group
def test_gather_golden(ext):
    g = golden("ops_gather.npz")
    out = ext.gather_points(dev(g["points"]), dev(g["idx"]))
    assert np.array_equal(bits(out.cpu().numpy()), bits(g["out"]))
    grad = ext.gather_points_grad(dev(g["grad_out"]), dev(g["idx"]), 500).cpu().numpy()
    np.testing.assert_allclose(grad, g["grad"], rtol=0, atol=1e-4)  # atomics: order-free sum


def test_ballquery_group_golden(ext):
    g = golden("ops_ballquery_group.npz")
    for t in range(5):
        r, ns = float(g["r_%d" % t]), int(g["ns_%d" % t])
        xyz, new_xyz = dev(g["xyz_%d" % t]), dev(g["new_xyz_%d" % t])
        idx = ext.ball_query(new_xyz, xyz, r, ns)
        assert idx.dtype == torch.int32
        assert np.array_equal(idx.cpu().numpy(), g["idx_%d" % t]), t
        if "feats_%d" % t in g:
            grouped = ext.group_points(dev(g["feats_%d" % t]), idx)
            assert np.array_equal(bits(grouped.cpu().numpy()), bits(g["grouped_%d" % t]))
            gg = ext.group_points_grad(dev(g["gout_%d" % t]), idx, xyz.shape[1]).cpu().numpy()
            np.testing.assert_allclose(gg, g["ggrad_%d" % t], rtol=0, atol=1e-4)


@pytest.mark.parametrize("b,n,m,r,ns", [(2, 4096, 512, 0.2, 32), (3, 1000, 77, 0.3, 16),
                                        (1, 130, 130, 0.5, 7), (2, 64, 5, 10.0, 128),
                                        (1, 5000, 300, 0.15, 64), (2, 2048, 1024, 0.4, 32),
                                        (1, 70, 9, 0.0, 4)])
def test_ballquery_vs_oracle(ext, oracle, synth, b, n, m, r, ns):
    # ragged sizes (n, m not multiples of 64), radius 0 (no hit anywhere -> zero rows),
    # huge radius (every row full after the first 128 points)
    xyz = synth.cloud_uniform(b, n, synth.cube_side(n, max(r, 0.1), ns), seed=n + m)
    new_xyz = xyz[:, :m].copy() if m <= n else synth.cloud_uniform(b, m, 1.0, seed=5)
    want = oracle.ball_query(new_xyz, xyz, r, ns)
    got = ext.ball_query(dev(new_xyz), dev(xyz), r, ns).cpu().numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("c,n,m,ns", [(1, 500, 33, 7), (3, 4096, 512, 32), (37, 700, 64, 16),
                                      (128, 2048, 256, 32),
                                      # LDS-staged tier: 64 KB and 160 KB rows, ragged n / m*ns,
                                      # the largest row that fits and the first that does not
                                      (3, 40000, 128, 64), (2, 40960, 100, 41), (1, 30001, 99, 43),
                                      (5, 16385, 64, 64), (2, 40961, 128, 32), (200, 1024, 128, 32),
                                      # many rows of a large cloud: one whole row per workgroup (b*c >= 128);
                                      # the few-row cases above run the range-partitioned kernel
                                      (70, 20000, 64, 32)])
def test_group_vs_oracle(ext, oracle, c, n, m, ns):
    g = np.random.default_rng(c + n)
    pts = g.standard_normal((2, c, n)).astype(np.float32)
    idx = g.integers(0, n, (2, m, ns)).astype(np.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]  # first-hit style padding: repeated indices
    out = ext.group_points(dev(pts), dev(idx)).cpu().numpy()
    assert np.array_equal(bits(out), bits(oracle.group_points(pts, idx)))
    gout = g.standard_normal(out.shape).astype(np.float32)
    got = ext.group_points_grad(dev(gout), dev(idx), n).cpu().numpy()
    np.testing.assert_allclose(got, oracle.group_points_grad(gout, idx, n), rtol=0, atol=1e-4)


@pytest.mark.parametrize("c,n,m,ns", [(4, 40000, 2048, 64), (3, 20000, 1024, 16), (8, 40000, 512, 32)])
def test_group_grad_few_rows_hot_spots(ext, oracle, c, n, m, ns):
    """The few-row scatter-add (ranges of destination points per workgroup, equal neighbours merged):
    ball-query style rows, a cloud whose indices alternate between two points (nothing to merge,
    every element lands in ONE range), a cloud on one point, points nobody refers to.  Sums in
    another order than the oracle's: bounded by the float64 sum and the mass added."""
    g = np.random.default_rng(c + n + m)
    idx = g.integers(0, n - n // 4, (3, m, ns)).astype(np.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]
    idx[1] = (np.arange(m * ns, dtype=np.int32) % 2).reshape(m, ns) * 7 + 3
    idx[2] = n - 1
    gout = g.standard_normal((3, c, m, ns)).astype(np.float32)
    got = ext.group_points_grad(dev(gout), dev(idx), n).cpu().numpy()
    np.testing.assert_allclose(got[:1], oracle.group_points_grad(gout[:1], idx[:1], n), rtol=0, atol=1e-4)
    truth = np.zeros((3, c, n)); mass = np.zeros((3, c, n))
    for bi in range(3):
        np.add.at(truth[bi], (slice(None), idx[bi].reshape(-1)), gout[bi].reshape(c, -1).astype(np.float64))
        np.add.at(mass[bi], (slice(None), idx[bi].reshape(-1)), np.abs(gout[bi].reshape(c, -1)).astype(np.float64))
    assert np.all(np.abs(got - truth) <= 1e-5 + 2e-7 * mass)


@pytest.mark.parametrize("c,n,m,ns", [(128, 2048, 1024, 32), (256, 1024, 512, 16), (256, 512, 256, 16),
                                      (7, 4096, 511, 64), (1, 1, 3, 5), (3, 100, 7, 128),
                                      (130, 257, 256, 128)])
def test_group_grad_through_inverse_index(ext, oracle, c, n, m, ns):
    """group_points_grad with the inverse of idx built once: every gradient element added once,
    same result as the oracle's scatter-add (group_points_gpu.cu:48-69).  Ball-query style idx
    (padding repeats the first hit) plus points nobody refers to."""
    g = np.random.default_rng(c * n + ns)
    idx = g.integers(0, max(1, n - n // 4), (3, m, ns)).astype(np.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]
    idx[1] = 0 if n < 3 else n - 1  # one cloud: every position on one point
    gout = g.standard_normal((3, c, m, ns)).astype(np.float32)
    inv = ext.group_inverse(dev(idx), n)
    entries = inv.shape[1]
    chunk = entries // 1024
    assert inv is not None and entries in (4096, 8192, 16384, 32768) and m * ns <= entries < max(2 * m * ns, 4097)
    # stored lane-interleaved: sorted entry s sits at [(s % chunk) * 1024 + s // chunk]
    packed = inv.cpu().numpy().view(np.uint32).reshape(3, chunk, 1024).transpose(0, 2, 1).reshape(3, entries)
    assert np.all(packed[:, m * ns:] == 0xFFFFFFFF)
    packed = packed[:, :m * ns]
    assert np.array_equal(np.sort(packed & 0xFFFF, axis=1),
                          np.broadcast_to(np.arange(m * ns, dtype=np.uint32), (3, m * ns)))
    assert np.all(np.diff((packed >> 16).astype(np.int64), axis=1) >= 0)
    assert np.array_equal(packed >> 16, np.take_along_axis(
        idx.reshape(3, -1).astype(np.uint32), (packed & 0xFFFF).astype(np.int64), axis=1))
    got = ext.group_points_grad_sorted(dev(gout), inv, n).cpu().numpy()
    for bi in (0, 2):
        np.testing.assert_allclose(got[bi:bi + 1], oracle.group_points_grad(gout[bi:bi + 1], idx[bi:bi + 1], n),
                                   rtol=0, atol=1e-4)
    # fp32 sums in another order than the oracle's: bound by the float64 sum and the mass added
    # (cloud 1 puts up to 32768 terms on one point)
    truth = np.zeros((3, c, n)); mass = np.zeros((3, c, n))
    for bi in range(3):
        np.add.at(truth[bi], (slice(None), idx[bi].reshape(-1)), gout[bi].reshape(c, -1).astype(np.float64))
        np.add.at(mass[bi], (slice(None), idx[bi].reshape(-1)), np.abs(gout[bi].reshape(c, -1)).astype(np.float64))
    assert np.all(np.abs(got - truth) <= 1e-5 + 2e-7 * mass)
    # the same channels taken in place from a wider tensor (the gradient of a grouped tensor);
    # runs that straddle lanes meet in LDS atomics, so two launches agree to rounding only
    wide = torch.cat([torch.full((3, 3, m, ns), 7.0, device=DEV), dev(gout)], dim=1)
    again = ext.group_points_grad_sorted(wide, inv, n, 3).cpu().numpy()
    assert np.all(np.abs(again - truth) <= 1e-5 + 2e-7 * mass)


def test_group_inverse_range(ext):
    """Outside n <= 4096 / m*ns <= 32768 there is no inverse: callers keep the atomic kernel."""
    idx = torch.zeros(1, 2048, 64, dtype=torch.int32, device=DEV)
    assert ext.group_inverse(idx, 40000) is None
    assert ext.group_inverse(idx[:, :1024], 4097) is None
    assert ext.group_inverse(idx[:, :512], 4096) is not None


def test_query_and_group_fused(ext, oracle, synth):
    """Fused front end == reference composition (pointnet2_utils.py:335-358) done with the
    oracle: idx exact, gathered features exact, relative xyz within 1 ulp-level 1e-6."""
    b, n, m, c, r, ns = 2, 3000, 200, 5, 0.25, 24
    xyz = synth.cloud_uniform(b, n, synth.cube_side(n, r, ns), seed=9)
    new_xyz = xyz[:, :m].copy()
    feats = np.random.default_rng(3).standard_normal((b, c, n)).astype(np.float32)
    want_idx = oracle.ball_query(new_xyz, xyz, r, ns)
    gx = oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), want_idx)
    gx = gx - new_xyz.transpose(0, 2, 1)[..., None]
    gf = oracle.group_points(feats, want_idx)
    for normalize in (False, True):
        idx, out = ext.query_and_group(dev(new_xyz), dev(xyz), dev(feats), r, ns, normalize)
        assert np.array_equal(idx.cpu().numpy(), want_idx)
        out = out.cpu().numpy()
        assert out.shape == (b, 3 + c, m, ns)
        assert np.array_equal(bits(out[:, 3:]), bits(gf))
        ref_xyz = gx / np.float32(r) if normalize else gx
        np.testing.assert_allclose(out[:, :3], ref_xyz, rtol=0, atol=1e-6)
    idx, out = ext.query_and_group(dev(new_xyz), dev(xyz), None, r, ns, False)
    assert out.shape == (b, 3, m, ns)


def test_north_star_pair_config2_properties(ext, synth):
    """B=8, N=40000, m=2048, r=0.2, ns=64 (the roofline target): properties that do not need
    the oracle at full size -- rows ascending then padded with the first hit, every listed
    index inside the ball, nothing smaller missed (checked on sampled rows), group == fancy
    indexing."""
    b, n, m, r, ns = 8, 40000, 2048, 0.2, 64
    xyz = synth.cloud_uniform(b, n, synth.cube_side(n, r, ns), seed=1)
    d_xyz = dev(xyz)
    fps = ext.furthest_point_sampling(d_xyz, m)
    new_xyz = ext.gather_points(d_xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    idx = ext.ball_query(new_xyz, d_xyz, r, ns).cpu().numpy()
    cen = new_xyz.cpu().numpy()
    r2 = np.float32(r) * np.float32(r)
    rng = np.random.default_rng(0)
    for bi, j in zip(rng.integers(0, b, 200), rng.integers(0, m, 200)):
        d = cen[bi, j] - xyz[bi]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        hits = np.nonzero(d2 < r2)[0][:ns]
        want = np.full(ns, hits[0])
        want[:len(hits)] = hits
        assert np.array_equal(idx[bi, j], want)
    feats = d_xyz.transpose(1, 2).contiguous()
    grouped = ext.group_points(feats, dev(idx))
    want = torch.gather(feats.unsqueeze(2).expand(b, 3, m, n), 3,
                        dev(idx).long().unsqueeze(1).expand(b, 3, m, ns))
    assert torch.equal(grouped, want)


@pytest.mark.parametrize("kind", ["step", "R"])
def test_north_star_pair_on_the_step_and_room_clouds(ext, oracle_omp, synth, kind):
    """The fused query + gather at the north-star shape on the two other clouds the roofline is
    quoted on (bench.pair_cloud): the xyz of the timed train step's batch (votenet/data.py, seed
    100: half the points inside <= 12 object boxes -> rows of > 64 records, balls with > 192
    hits: the general path with its adaptive cut) and cloud R (room shell; walls at x = 0 put
    centroids on the lattice seam).  Two clouds at full size against the oracle, idx and grouped
    tensor bit-exact, with and without the cell lists the sampling kernel leaves behind."""
    import bench
    b, n, m, r, ns = 2, 40000, 2048, 0.2, 64
    xyz = bench.pair_cloud(kind)[:b].numpy().copy()
    d_xyz = dev(xyz)
    fps, lists = ext.furthest_point_sampling_with_grid(d_xyz, m, r)
    assert np.array_equal(fps.cpu().numpy(), oracle_omp.furthest_point_sampling(xyz, m))
    flipped = d_xyz.transpose(1, 2).contiguous()
    new_xyz = ext.gather_points(flipped, fps).transpose(1, 2).contiguous()
    cen = new_xyz.cpu().numpy()
    want_idx = oracle_omp.ball_query(cen, xyz, r, ns)
    feat = np.random.default_rng(5).standard_normal((b, 1, n)).astype(np.float32)
    gx = oracle_omp.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), want_idx)
    gx = (gx - cen.transpose(0, 2, 1)[..., None]) * (np.float32(1.0) / np.float32(r))
    gf = oracle_omp.group_points(feat, want_idx)
    for grid in (lists, None):
        idx, out = ext.query_and_group(new_xyz, d_xyz, dev(feat), r, ns, True, None, grid)
        assert np.array_equal(idx.cpu().numpy(), want_idx)
        out = out.cpu().numpy()
        assert np.array_equal(bits(out[:, :3]), bits(gx.astype(np.float32)))
        assert np.array_equal(bits(out[:, 3:]), bits(gf))
    # how dense this cloud is (the test is only worth its name if the dense paths ran)
    if kind == "step":
        d2 = ((cen[0, :, None, :] - xyz[0, None, ::8, :]) ** 2).sum(-1)
        assert ((d2 < r * r).sum(1) * 8).max() > 192


def test_launch_order_left_by_the_sampling_kernel(ext, oracle_omp, synth):
    """The sampling kernel leaves, next to the cell lists, the order in which the query kernel
    answers the centroids it picked: a permutation of 0..m-1 per cloud, sorted by the query's
    cost class (0: every x-row of the 3x3 neighbourhood is shorter than a wave; otherwise the number of
    64-record chunks of the general path), longest first -- recomputed here from the cell
    offsets.  Lists built any other way carry no order; a query for another number of
    centroids ignores it; rows are the oracle's in every case."""
    import bench
    b, n, m, r, ns = 2, 40000, 2048, 0.2, 64
    xyz = bench.pair_cloud("step")[:b].numpy().copy()
    d_xyz = dev(xyz)
    fps, lists = ext.furthest_point_sampling_with_grid(d_xyz, m, r)
    order_for, order, start = (t.cpu().numpy() for t in lists.launch_order())
    assert list(order_for) == [m] * b and start.shape[1] == 32 ** 3 + 1
    new_xyz = ext.gather_points(d_xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    cen = new_xyz.cpu().numpy()
    inv_side = np.float32(1.0) / (np.float32(r) * np.float32(1.001))
    heavy = 0
    for i in range(b):
        assert np.array_equal(np.sort(order[i, :m]), np.arange(m))
        g = np.floor(cen[i] * inv_side).astype(np.int64)
        gx, gy, gz = g[:, 0] & 31, g[:, 1], g[:, 2]
        xa, xb = np.maximum(gx - 1, 0), np.minimum(gx + 1, 31)
        seam = (gx == 0) | (gx == 31)
        fast = np.ones(m, bool)
        chunks = np.zeros(m, np.int64)
        for rz in range(3):
            for ry in range(3):
                base = (((gz + rz - 1) & 31) * 32 + ((gy + ry - 1) & 31)) * 32
                ln = start[i][base + xb + 1] - start[i][base + xa]
                wrap = base + np.where(gx == 0, 31, 0)
                lw = np.where(seam, start[i][wrap + 1] - start[i][wrap], 0)
                fast &= ln + lw < 64
                chunks += (ln + 63) // 64 + (lw + 63) // 64
        cls = np.where(fast, 0, np.clip(chunks, 1, 63))
        along = cls[order[i, :m]]
        assert np.all(along[:-1] >= along[1:])
        heavy += int((cls > 0).sum())
    assert heavy > 100   # (the dense clusters of this cloud are what the order is for)
    want = oracle_omp.ball_query(cen, xyz, r, ns)
    assert np.array_equal(ext.ball_query_prebuilt(new_xyz, d_xyz, r, ns, lists).cpu().numpy(), want)
    fewer = new_xyz[:, :777].contiguous()
    assert np.array_equal(ext.ball_query_prebuilt(fewer, d_xyz, r, ns, lists).cpu().numpy(), want[:, :777])
    assert list(ext.build_grid(d_xyz, r).launch_order()[0].cpu().numpy()) == [0] * b


@pytest.mark.parametrize("kind,shift,ns,c", [("step", 0.0, 64, 1), ("U", 0.0, 64, 1), ("R", 0.0, 64, 3),
                                             ("step", -1.7, 64, 0), ("U", -0.3, 128, 1), ("step", 0.0, 16, 8)])
def test_query_plans_left_by_the_sampling_kernel(ext, oracle_omp, synth, kind, shift, ns, c):
    """include/pn2_hip.h pn2_query_and_group_picks: when the centroids ARE the picks of the sampling
    call that left the cell lists (what a set-abstraction layer computes, pointnet2_modules.py:
    236-250), every query wave starts from the 64-byte plan that kernel wrote -- row offsets, row
    lengths, centroid, launch position.  Rows and grouped tensor: bit-for-bit those of the call
    that knows nothing about the centroids, and the oracle's (ball_query_gpu.cu:14-49,
    group_points_gpu.cu:13-33).  Clouds with dense clusters (general path inside a planned launch),
    walls at x = 0 and negative coordinates (wrapped seam cells), nsample 16 / 64 / 128."""
    import bench
    b, n, m, r = 2, 40000, 2048, 0.2
    xyz = bench.pair_cloud(kind)[:b].numpy().copy() + np.float32(shift)
    d_xyz = dev(xyz)
    feat = torch.rand(b, c, n, device=d_xyz.device) if c else None
    fps, lists = ext.furthest_point_sampling_with_grid(d_xyz, m, r)
    new_xyz = ext.gather_points(d_xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    assert not lists.centroids_are_picks(new_xyz)
    lists.mark_centroids(new_xyz, fps)
    assert lists.centroids_are_picks(new_xyz) and not lists.centroids_are_picks(new_xyz.clone())
    idx_p, grouped_p = ext.query_and_group(new_xyz, d_xyz, feat, r, ns, True, None, lists)   # plans
    other = new_xyz.clone()
    idx_o, grouped_o = ext.query_and_group(other, d_xyz, feat, r, ns, True, None, lists)     # no plans
    want = oracle_omp.ball_query(new_xyz.cpu().numpy(), xyz, r, ns)
    assert np.array_equal(idx_o.cpu().numpy(), want)
    assert np.array_equal(idx_p.cpu().numpy(), want)
    assert np.array_equal(bits(grouped_p.cpu().numpy()), bits(grouped_o.cpu().numpy()))
    # a centroid tensor modified in place is no longer "the picks": the ordinary path answers it
    new_xyz[:, 5] += 0.01
    assert not lists.centroids_are_picks(new_xyz)
    idx_m, _ = ext.query_and_group(new_xyz, d_xyz, feat, r, ns, True, None, lists)
    assert np.array_equal(idx_m.cpu().numpy(), oracle_omp.ball_query(new_xyz.cpu().numpy(), xyz, r, ns))


@pytest.mark.parametrize("ns", [64, 128])
def test_query_plans_dense_balls(ext, oracle_omp, synth, ns):
    """Balls with far more hits than the hit list holds, answered from the plans: blobs of 2000 and
    600 points with indices spread over the cloud (the sample's index histogram gives a cut, all
    passes run again below it), one blob of 2500 CONSECUTIVE indices (one index bucket alone
    overflows the list: the general path's bucket refinement decides) and one at the lattice seam
    (x around 0: wrapped cells AND long rows).  Rows = the oracle's (ball_query_gpu.cu:24-47 stops
    at cnt == nsample: only the nsample smallest indices matter)."""
    g = np.random.default_rng(77 + ns)
    b, n, m, r = 2, 40000, 2048, 0.2
    xyz = synth.cloud_uniform(b, n, 3.0, seed=9)
    perm = g.permutation(n)
    xyz[0, perm[:2000]] = 1.0 + g.random((2000, 3), dtype=np.float32) * 0.1
    xyz[0, perm[2000:2600]] = np.array([2.0, 1.5, 0.7], np.float32) + g.random((600, 3), dtype=np.float32) * 0.15
    xyz[1, 5000:7500] = 1.7 + g.random((2500, 3), dtype=np.float32) * 0.12
    xyz[1, perm[:900]] = np.array([-0.05, 1.0, 1.0], np.float32) + g.random((900, 3), dtype=np.float32) * 0.12
    d_xyz = dev(xyz)
    fps, lists = ext.furthest_point_sampling_with_grid(d_xyz, m, r)
    new_xyz = ext.gather_points(d_xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    lists.mark_centroids(new_xyz, fps)
    assert lists.centroids_are_picks(new_xyz)
    feat = torch.rand(b, 1, n, device=d_xyz.device)
    idx, grouped = ext.query_and_group(new_xyz, d_xyz, feat, r, ns, False, None, lists)
    cen = new_xyz.cpu().numpy()
    want = oracle_omp.ball_query(cen, xyz, r, ns)
    # (the blobs hold picks: some balls have hundreds of hits)
    d2 = ((xyz[0][None, :, :] - cen[0][:, None, :]) ** 2).sum(-1)
    assert int(((d2 < r * r).sum(1) > 256).sum()) >= 1
    assert np.array_equal(idx.cpu().numpy(), want)
    ref = np.take_along_axis(xyz[:, None, :, :], want[:, :, :, None].astype(np.int64), axis=2) - cen[:, :, None, :]
    assert np.array_equal(bits(grouped[:, :3].permute(0, 2, 3, 1).cpu().numpy()), bits(ref.astype(np.float32)))


def test_query_plans_need_room(ext, oracle_omp, synth):
    """More centroids than the object has plans for (m > n / 8): the sampling kernel writes none and
    pn2_query_and_group_picks is pn2_query_and_group_prebuilt."""
    b, n, m, r, ns = 2, 8192, 2048, 0.25, 32
    xyz = synth.cloud_uniform(b, n, synth.cube_side(n, r, ns), seed=5)
    d_xyz = dev(xyz)
    fps, lists = ext.furthest_point_sampling_with_grid(d_xyz, m, r)
    if lists is None:
        pytest.skip("no cell lists for this cloud size")
    new_xyz = ext.gather_points(d_xyz.transpose(1, 2).contiguous(), fps).transpose(1, 2).contiguous()
    lists.mark_centroids(new_xyz, fps)
    idx, _ = ext.query_and_group(new_xyz, d_xyz, None, r, ns, True, None, lists)
    assert np.array_equal(idx.cpu().numpy(), oracle_omp.ball_query(new_xyz.cpu().numpy(), xyz, r, ns))


def test_ballquery_general_path_bucket_refinement(ext, oracle_omp, synth):
    """n = 80 000 (index buckets of 2048) with 3000 CONSECUTIVE indices packed into one cell and
    2000 more into its neighbour: one bucket alone overflows the hit list, so the adaptive cut
    fails and the counting sweeps split the bucket (width 2048 -> 32) before the list is
    collected.  nsample 64 and 128."""
    g = np.random.default_rng(23)
    b, n, m, r = 2, 80000, 256, 0.2
    xyz = synth.cloud_uniform(b, n, 3.5, seed=41)
    xyz[0, 1000:4000] = 1.05 + g.random((3000, 3), dtype=np.float32) * 0.08
    xyz[0, 50000:52000] = np.array([1.25, 1.05, 1.05], np.float32) + g.random((2000, 3), dtype=np.float32) * 0.08
    xyz[1, :2500] = 2.0 + g.random((2500, 3), dtype=np.float32) * 0.02
    cen = xyz[:, g.permutation(n)[:m]].copy()
    cen[0, :64] = xyz[0, 1000:4000:47][:64]     # centroids inside the blobs
    cen[1, :32] = xyz[1, 0:2500:79][:32]
    for ns in (64, 128):
        want = oracle_omp.ball_query(cen, xyz, r, ns)
        got = ext.ball_query(dev(cen), dev(xyz), r, ns).cpu().numpy()
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]


def test_stress_config5_sa1_front_end(ext, oracle_omp, synth):
    """BASELINE configs[4] (stress): B=32, N=80000, nsample=128, the SA1 front end end to end at
    full size -- FPS on the bucketed tier (two metadata sets per lane beyond 65 536 points),
    ball query + gathers on the cell-list tier with 64 < nsample <= 128.  FPS and the ball query
    of two clouds against the oracle; sampled rows of all clouds against a direct numpy scan;
    the fused grouped tensor against torch.gather of the same indices."""
    b, n, m, r, ns = 32, 80000, 2048, 0.2, 128
    xyz = synth.cloud_uniform(b, n, synth.cube_side(n, r, ns), seed=5)
    d_xyz = dev(xyz)
    fps = ext.furthest_point_sampling(d_xyz, m)
    got_fps = fps.cpu().numpy()
    assert np.array_equal(got_fps[:2], oracle_omp.furthest_point_sampling(xyz[:2], m))
    assert np.all(got_fps[:, 0] == 0) and all(len(np.unique(row)) == m for row in got_fps)
    flipped = d_xyz.transpose(1, 2).contiguous()
    new_xyz = ext.gather_points(flipped, fps).transpose(1, 2).contiguous()
    feat = torch.rand(b, 1, n, device=DEV)
    idx, grouped = ext.query_and_group(new_xyz, d_xyz, feat, r, ns, True)
    idx_np, cen = idx.cpu().numpy(), new_xyz.cpu().numpy()
    assert np.array_equal(idx_np[:2], oracle_omp.ball_query(cen[:2], xyz[:2], r, ns))
    r2 = np.float32(r) * np.float32(r)
    rng = np.random.default_rng(1)
    for bi, j in zip(rng.integers(0, b, 150), rng.integers(0, m, 150)):
        d = cen[bi, j] - xyz[bi]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        hits = np.nonzero(d2 < r2)[0][:ns]
        want = np.full(ns, hits[0])
        want[:len(hits)] = hits
        assert np.array_equal(idx_np[bi, j], want)
    assert torch.equal(ext.ball_query(new_xyz, d_xyz, r, ns), idx)
    both = torch.cat([flipped, feat], dim=1)
    want = torch.gather(both.unsqueeze(2).expand(b, 4, m, n), 3,
                        idx.long().unsqueeze(1).expand(b, 4, m, ns))
    want[:, :3] = (want[:, :3] - new_xyz.transpose(1, 2).unsqueeze(-1)) * (np.float32(1.0) / np.float32(r))
    assert torch.equal(grouped, want)
    # the reference-surface operators on the same indices (group_points falls back from the
    # LDS-staged tier: a channel row of 80 000 floats does not fit 160 KB)
    assert torch.equal(ext.group_points(feat, idx), want[:, 3:])
    g = ext.group_points_grad(torch.ones(b, 1, m, ns, device=DEV), idx, n)
    assert abs(float(g.sum()) - b * m * ns) < 1e-3 * b * m * ns


def test_stress_config5_iou_matrix_and_nms(oracle_omp, synth):
    """BASELINE configs[4]: the 1024 x 1024 oriented 3-D IoU matrix against the oracle (1e-4)."""
    load = __import__("importlib").import_module
    ut = load("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    a, bb = synth.boxes_pair(1024, seed=11)
    got = ut.boxes_iou3d_gpu(dev(a), dev(bb)).cpu().numpy()
    np.testing.assert_allclose(got, oracle_omp.boxes_iou3d(a, bb), rtol=0, atol=1e-4)


# ------------------------------------------------------------------------------ three_nn etc.
def test_interp_golden(ext):
    g = golden("ops_interp.npz")
    d2, idx = ext.three_nn(dev(g["unknown"]), dev(g["known"]))
    assert np.array_equal(idx.cpu().numpy(), g["idx"])
    assert np.array_equal(bits(d2.cpu().numpy()), bits(g["dist2"]))
    d2s, idxs = ext.three_nn(dev(g["unknown"][:, :9]), dev(g["known"][:, :2]))
    assert np.array_equal(bits(d2s.cpu().numpy()), bits(g["dist2_small"]))
    assert np.array_equal(idxs.cpu().numpy(), g["idx_small"])
    out = ext.three_interpolate(dev(g["feats"]), dev(g["idx"]), dev(g["weight"]))
    np.testing.assert_allclose(out.cpu().numpy(), g["interp"], rtol=0, atol=1e-4)
    assert np.array_equal(bits(out.cpu().numpy()), bits(g["interp"]))  # same op order: exact
    grad = ext.three_interpolate_grad(dev(g["grad_out"]), dev(g["idx"]), dev(g["weight"]), 300)
    np.testing.assert_allclose(grad.cpu().numpy(), g["grad"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("b,c,m,n", [(2, 20, 700, 4096), (1, 9, 2048, 2048), (2, 33, 50, 2052),
                                     (1, 5, 2049, 4096), (2, 17, 1024, 3000), (1, 40, 1500, 900),
                                     (2, 256, 1024, 8192)])
def test_three_interpolate_every_kernel_vs_oracle(ext, oracle, b, c, m, n):
    """Forward (LDS-staged rows for m <= 2048 and n >= 2048, global gathers otherwise) bit-exact
    against the oracle; backward (LDS-privatised rows for m <= 1024, global atomics otherwise)
    within the fp32 reordering of the scatter-add."""
    g = np.random.default_rng(c * 1000 + m + n)
    feats = g.standard_normal((b, c, m)).astype(np.float32)
    idx = g.integers(0, m, (b, n, 3)).astype(np.int32)
    idx[:, ::7, 1] = idx[:, ::7, 0]  # repeated neighbours: same-address accumulation
    w = g.random((b, n, 3)).astype(np.float32)
    w = (w / w.sum(axis=2, keepdims=True)).astype(np.float32)
    out = ext.three_interpolate(dev(feats), dev(idx), dev(w)).cpu().numpy()
    assert np.array_equal(bits(out), bits(oracle.three_interpolate(feats, idx, w)))
    go = g.standard_normal((b, c, n)).astype(np.float32)
    grad = ext.three_interpolate_grad(dev(go), dev(idx), dev(w), m).cpu().numpy()
    want = oracle.three_interpolate_grad(go, idx, w, m)
    np.testing.assert_allclose(grad, want, rtol=0, atol=1e-4 * max(1.0, np.abs(want).max()))


@pytest.mark.parametrize("b,c,m,n,ctot,c0", [(2, 20, 700, 4096, 31, 3), (1, 9, 2048, 2048, 9, 0),
                                             (2, 33, 50, 2052, 40, 7), (1, 5, 3000, 37, 6, 1),
                                             (8, 256, 1024, 16384, 259, 3)])
def test_interpolate_on_channel_slices(ext, b, c, m, n, ctot, c0):
    """three_interpolate written into / its gradient read out of a channel slice of a wider
    tensor (the feature-propagation concatenation without the copy) == the plain operators on
    separate tensors, bit for bit (forward) / to the atomics' summation order (backward)."""
    g = torch.Generator().manual_seed(b * 100 + c + m + n)
    pts = torch.randn(b, c, m, generator=g).to(DEV)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32).to(DEV)
    w = torch.rand(b, n, 3, generator=g).to(DEV)
    want = ext.three_interpolate(pts, idx, w)
    out = torch.full((b, ctot, n), -7.0, device=DEV)
    ext.three_interpolate_into(pts, idx, w, out, c0)
    assert torch.equal(out[:, c0:c0 + c], want)
    assert bool((out[:, :c0] == -7.0).all()) and bool((out[:, c0 + c:] == -7.0).all())
    grad = torch.randn(b, ctot, n, generator=g).to(DEV)
    want_g = ext.three_interpolate_grad(grad[:, c0:c0 + c].contiguous(), idx, w, m)
    got_g = ext.three_interpolate_grad_from(grad, c0, c, idx, w, m)
    torch.testing.assert_close(got_g, want_g, rtol=0, atol=1e-4)


@pytest.mark.parametrize("b,n,m", [(2, 4096, 1024), (1, 1000, 1500), (3, 37, 5), (1, 2500, 2049),
                                   (8, 9216, 1024), (2, 300, 2), (1, 64, 1023), (40, 8192, 512)])
def test_three_nn_vs_oracle(ext, oracle, synth, b, n, m):
    unk = synth.cloud_uniform(b, n, 2.0, seed=n)
    kn = synth.cloud_uniform(b, m, 2.0, seed=m + 1)
    kn[:, m // 2] = kn[:, 0]  # duplicate known point -> distance ties
    wd, wi = oracle.three_nn(unk, kn)
    d2, idx = ext.three_nn(dev(unk), dev(kn))
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.array_equal(bits(d2.cpu().numpy()), bits(wd))


@pytest.mark.parametrize("b,blocks,m,spread", [(2, 64, 1024, 0.4), (1, 33, 2048, 0.05), (3, 20, 256, 1.5),
                                                (1, 16, 700, 0.0)])
def test_three_nn_pruned_blocks_vs_oracle(ext, oracle, synth, b, blocks, m, spread):
    """Queries that arrive in spatially coherent blocks of 128 (the grid points of a proposal box):
    the kernel prunes the known set per block by a triangle-inequality bound -- results must stay
    bit-identical to the full scan (indices incl. the first-wins tie rule, squared distances);
    duplicate known points, a block of identical queries (spread 0), a ragged last block."""
    rng = np.random.default_rng(blocks * 7 + m)
    kn = synth.cloud_uniform(b, m, 3.0, seed=m + 2)
    kn[:, m // 2] = kn[:, 0]
    kn[:, m // 3] = kn[:, 1]
    centres = rng.uniform(0.0, 3.0, (b, blocks, 1, 3))
    unk = (centres + rng.uniform(-spread, spread, (b, blocks, 128, 3))).reshape(b, blocks * 128, 3)
    unk = unk[:, :blocks * 128 - 37].astype(np.float32).copy()
    unk[:, 5] = kn[:, 0]  # a query sitting exactly on a duplicated known point
    wd, wi = oracle.three_nn(unk, kn)
    d2, idx = ext.three_nn(dev(unk), dev(kn))
    assert np.array_equal(idx.cpu().numpy(), wi)
    assert np.array_equal(bits(d2.cpu().numpy()), bits(wd))


def test_gridconv_three_nn_properties(ext, synth):
    """GridConv size (B=8, 32768 x 1024): distances sorted, indices valid, and exact against a
    torch fp32 top-3 on sampled queries."""
    unk = synth.cloud_uniform(8, 32768, 3.0, seed=2)
    kn = synth.cloud_uniform(8, 1024, 3.0, seed=3)
    d2, idx = ext.three_nn(dev(unk), dev(kn))
    assert bool((d2[..., 0] <= d2[..., 1]).all()) and bool((d2[..., 1] <= d2[..., 2]).all())
    assert int(idx.min()) >= 0 and int(idx.max()) < 1024
    u = dev(unk[:, :256])
    k = dev(kn)
    diff = u[:, :, None, :] - k[:, None, :, :]
    dd = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    want = torch.sort(dd, dim=2, stable=True)
    assert torch.equal(idx[:, :256].long(), want.indices[..., :3])


# ------------------------------------------------------------------------------------ IoU / NMS
@pytest.mark.parametrize("tag", ["oriented", "aligned", "kat"])
def test_iou_vs_reference_golden(iou_ext, tag):
    """Device BEV overlap / IoU against vectors from the REFERENCE's compiled CPU code;
    tolerance 1e-4 (device sin/cos/atan2 differ from glibc in the last ulps)."""
    g = golden("iou_bev_cpu_ref.npz")
    a, b = dev(g["a_" + tag]), dev(g["b_" + tag])
    ov = torch.zeros(a.shape[0], b.shape[0], device=DEV)
    iou_ext.boxes_overlap_bev_gpu(a, b, ov)
    np.testing.assert_allclose(ov.cpu().numpy(), g["overlap_" + tag], rtol=0, atol=1e-4)
    iou = torch.zeros_like(ov)
    iou_ext.boxes_iou_bev_gpu(a, b, iou)
    np.testing.assert_allclose(iou.cpu().numpy(), g["iou_bev_" + tag], rtol=0, atol=1e-4)


@pytest.mark.parametrize("na,nb,seed,aligned", [(64, 64, 0, False), (256, 256, 1, False),
                                                (256, 256, 2, True), (33, 70, 3, False),
                                                (1, 1, 4, False), (17, 300, 5, True)])
def test_iou3d_vs_oracle(oracle, synth, na, nb, seed, aligned):
    import importlib
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    n = max(na, nb)
    a, b = synth.boxes_pair(n, seed=seed, axis_aligned=aligned)
    a, b = a[:na], b[:nb]
    got = ut.boxes_iou3d_gpu(dev(a), dev(b)).cpu().numpy()
    want = oracle.boxes_iou3d(a, b)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    assert (want > 0).any() or n == 1


def test_iou3d_known_answers_and_empty(synth):
    import importlib
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    a, b = synth.box_kats()
    iou = ut.boxes_iou3d_gpu(dev(a), dev(b)).cpu().numpy()
    assert abs(iou[0, 0] - 0.125) < 1e-5 and iou[1, 1] == 0.0 and iou[5, 5] == 0.0
    g = golden("iou3d_nms_oracle.npz")
    np.testing.assert_allclose(iou, g["iou3d_kat"], rtol=0, atol=1e-4)
    empty = ut.boxes_iou3d_gpu(dev(a[:0]), dev(b))
    assert tuple(empty.shape) == (0, len(b))


def test_train_step_iou_shape_properties(synth):
    """(B*K) x (B*64) = 2048 x 512 as compute_iou_labels issues it: IoU in [0,1], symmetric
    under swapping the operands, and identical boxes give ~1."""
    import importlib
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    a, b = synth.boxes_pair(2048, seed=8)
    a_d, b_d = dev(a), dev(b[:512])
    iou = ut.boxes_iou3d_gpu(a_d, b_d)
    assert float(iou.min()) >= 0.0 and float(iou.max()) <= 1.0 + 1e-4
    iou_t = ut.boxes_iou3d_gpu(b_d, a_d)
    assert float((iou - iou_t.t()).abs().max()) <= 1e-4
    self_iou = ut.boxes_iou3d_gpu(a_d[:256], a_d[:256]).diagonal()
    assert float((self_iou - 1).abs().max()) < 5e-2


@pytest.mark.parametrize("scenes,p,g,seed", [(8, 256, 64, 0), (3, 33, 7, 1), (2, 16, 100, 2),
                                             (1, 1, 1, 3), (4, 50, 0, 4)])
def test_scene_best_iou3d_vs_oracle(oracle, synth, scenes, p, g, seed):
    """Per-scene best IoU + first arg-max == block diagonal of the oracle's all-pairs matrix, bit
    for bit the same kernel arithmetic as boxes_iou3d_gpu; empty GT sets give zeros."""
    import importlib
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    a, b = synth.boxes_pair(max(scenes * p, scenes * max(g, 1)), seed=seed)
    a = a[:scenes * p].reshape(scenes, p, 7)
    b = b[:scenes * g].reshape(scenes, g, 7)
    a[:, p // 2:] += 50.0  # half of the predictions overlap nothing: all-zero rows -> index 0
    best, idx = ut.boxes_iou3d_scene_max_gpu(dev(a), dev(b))
    assert best.shape == (scenes, p) and idx.dtype == torch.int64
    if g == 0:
        assert float(best.abs().max()) == 0.0 and int(idx.abs().max()) == 0
        return
    full = ut.boxes_iou3d_gpu(dev(a.reshape(-1, 7)), dev(b.reshape(-1, 7))).cpu().numpy()
    want = oracle.boxes_iou3d(a.reshape(-1, 7), b.reshape(-1, 7))
    for s in range(scenes):
        blk_gpu = full[s * p:(s + 1) * p, s * g:(s + 1) * g]
        blk = want[s * p:(s + 1) * p, s * g:(s + 1) * g]
        np.testing.assert_array_equal(best[s].cpu().numpy(), blk_gpu.max(axis=1))
        np.testing.assert_array_equal(idx[s].cpu().numpy(), blk_gpu.argmax(axis=1))
        np.testing.assert_allclose(best[s].cpu().numpy(), blk.max(axis=1), rtol=0, atol=1e-4)


def _mask_words(t):
    return t.cpu().numpy().view(np.uint64)


def test_nms_vs_oracle(iou_ext, oracle, synth):
    import importlib
    L = importlib.import_module("3dioumatch_amd._lib")
    g = golden("iou3d_nms_oracle.npz")
    boxes = g["nms_boxes"]
    n = len(boxes)
    d = dev(boxes)
    cb = (n + 63) // 64
    for thr, tag in ((0.25, "0p25"), (0.5, "0p5")):
        for normal, kkey, mkey in ((False, "keep_", "mask_"), (True, "keepn_", "maskn_")):
            keep = torch.empty(n, dtype=torch.int64)
            fn = iou_ext.nms_normal_gpu if normal else iou_ext.nms_gpu
            num = fn(d, keep, thr)
            assert np.array_equal(keep[:num].numpy(), g[kkey + tag]), (thr, normal)
            keep32 = torch.empty(n, dtype=torch.int32)
            assert fn(d, keep32, thr) == num and np.array_equal(keep32[:num].numpy(), g[kkey + tag])
            # the full mask through the launcher-level ABI, bit for bit
            mask = torch.zeros(n * cb, dtype=torch.int64, device=DEV)
            mfn = L.lib.iou3d_nms_normal_mask if normal else L.lib.iou3d_nms_mask
            L.check(mfn(d.data_ptr(), mask.data_ptr(), n, thr,
                        torch.cuda.current_stream().cuda_stream), "mask")
            torch.cuda.synchronize()
            want = g[mkey + tag]
            got = _mask_words(mask).reshape(n, cb)
            flips = np.count_nonzero(got != want)
            assert flips == 0, "%d mask words differ (IoU within ulps of the threshold?)" % flips


def test_nms_wrapper_and_stress_size(oracle, synth):
    import importlib
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    boxes, scores = synth.boxes_scored(1024, seed=12, spread=8.0)
    perm = np.random.default_rng(1).permutation(1024)
    keep, _ = ut.nms_gpu(dev(boxes[perm]), dev(scores[perm]), 0.25)
    want, _ = oracle.nms(boxes, 0.25)
    inv = np.argsort(perm)  # index in the shuffled array of sorted box i is where perm == i
    got_sorted = np.array([np.nonzero(perm[k] == np.arange(1024))[0][0] for k in keep.cpu().numpy()])
    assert len(got_sorted) == len(want)
    assert np.array_equal(got_sorted, want)
    assert inv.shape == (1024,)
    keep_n, _ = ut.nms_normal_gpu(dev(boxes), dev(scores), 0.5)
    want_n, _ = oracle.nms_normal(boxes, 0.5)
    assert np.array_equal(keep_n.cpu().numpy(), want_n)
    k1, _ = ut.nms_gpu(dev(boxes), dev(scores), 0.25, pre_maxsize=100)
    w1, _ = oracle.nms(boxes[:100], 0.25)
    assert np.array_equal(k1.cpu().numpy(), w1)


@pytest.mark.parametrize("n,spread", [(1, 8.0), (63, 2.0), (64, 2.0), (65, 2.0), (130, 1.0), (1000, 8.0),
                                      (1024, 3.0), (1025, 8.0), (1500, 4.0), (2300, 6.0)])
def test_nms_scan_block_by_block(iou_ext, oracle, synth, n, spread):
    """The device scan decides a block of 64 boxes per step (fixed point of the suppression inside
    the block, rows of the survivors ORed into the removed set): the reference's one-box-at-a-time
    loop (iou3d_nms.cpp:121-134) gives the same list.  Sizes around the block width, both forms
    of the kernel (rows in registers up to 1024 boxes, any size beyond), sparse scenes and dense
    ones (small spread: long suppression chains inside a block)."""
    boxes, scores = synth.boxes_scored(n, seed=100 + n, spread=spread)
    d = dev(boxes)
    for thr in (0.1, 0.25):
        for normal in (False, True):
            keep, num = iou_ext.nms_device(d, thr, normal)
            want, _ = (oracle.nms_normal if normal else oracle.nms)(boxes, thr)
            assert num == len(want), (n, thr, normal, num, len(want))
            assert np.array_equal(keep[:num].cpu().numpy(), want)
            host = torch.empty(n, dtype=torch.int64)
            assert (iou_ext.nms_normal_gpu if normal else iou_ext.nms_gpu)(d, host, thr) == num
            assert np.array_equal(host[:num].numpy(), want)


def test_streams_and_second_device_context(ext, oracle, synth):
    """Kernels run on torch's CURRENT stream (the reference's iou3d launches use the legacy
    default stream): results issued on a side stream are ordered with that stream."""
    xyz = synth.cloud_uniform(2, 2000, 1.5, seed=77)
    want = oracle.furthest_point_sampling(xyz, 64)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d = dev(xyz)
        got = ext.furthest_point_sampling(d, 64)
        idx = ext.ball_query(d[:, :64].contiguous(), d, 0.3, 16)
    s.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(idx.cpu().numpy(), oracle.ball_query(xyz[:, :64], xyz, 0.3, 16))


def _first_tie_reference(xyz, inds):
    """Per cloud: the first round of the reference's FPS (picks = inds) whose maximum running
    distance is held by two or more participating points, or is zero (nothing left to take);
    len(picks) if none.  fp32, every operation rounded, the reference's order
    (sampling_gpu.cu:100-118)."""
    out = []
    m = inds.shape[1]
    for b in range(xyz.shape[0]):
        p = xyz[b].astype(np.float32)
        mag = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]
        part = mag.astype(np.float64) > 1e-3
        td = np.where(part, np.float32(1e10), np.float32(-1.0)).astype(np.float32)
        first = m if part.any() else 0
        for j in range(1, m):
            if not part.any():
                break
            d = p - p[inds[b, j - 1]]
            d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            td = np.where(part, np.minimum(d, td), td)
            best = td[part].max()
            assert td[inds[b, j]] == best
            if (td[part] == best).sum() > 1 or not best > 0:
                first = j
                break
        out.append(first)
    return np.array(out, np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["uniform", "room", "duplicates", "identical", "mixed", "two_points",
                                  "few_points"])
def test_fps_ties_and_prefix(ext, oracle_omp, synth, case):
    """Sampling a sampled cloud (SA2..SA4): the picks of a run, in order, sample to 0, 1, 2, ...
    as long as the run met no exact tie (include/pn2_hip.h).  The kernel's first_tie against a
    numpy restatement, and furthest_point_sampling_prefix against the oracle's FPS of the head --
    both where the shortcut applies and where a tie forces the real sampling."""
    n, m1, m2 = 9000, 700, 300
    if case == "uniform":
        xyz = synth.cloud_uniform(2, n, 3.0, seed=11)
    elif case == "room":
        xyz = synth.cloud_room(2, n, seed=12)
    elif case == "duplicates":
        xyz = synth.cloud_edge_cases(2, n, 2.0, seed=13, near_origin=8, duplicates=400)
    elif case == "identical":
        xyz = np.ones((2, n, 3), np.float32)
        xyz[1, : n // 2] = 4.0
    elif case == "mixed":
        xyz = synth.cloud_uniform(2, n, 3.0, seed=14)
        xyz[1] = synth.cloud_edge_cases(1, n, 2.0, seed=15, near_origin=8, duplicates=400)[0]
    else:
        # almost every point inside the skip radius: the run exhausts its candidates after one
        # (two_points) or 40 (few_points) rounds and repeats picks from then on -- the picked
        # sequence holds the same point many times and ties with itself when sampled as a cloud
        g = np.random.default_rng(16)
        xyz = (g.random((2, n, 3), dtype=np.float32) - 0.5) * 0.02
        keep = 1 if case == "two_points" else 40
        xyz[:, 0] = 3.0
        xyz[:, 100:100 + keep] = g.random((2, keep, 3), dtype=np.float32) * 2.0 + 1.0
    xyz = np.ascontiguousarray(xyz, np.float32)
    t = torch.from_numpy(xyz).cuda()
    inds, lists, ties = ext.furthest_point_sampling_ties(t, m1, 0.3)
    assert ties is not None and lists is not None
    want1 = oracle_omp.furthest_point_sampling(xyz, m1)
    assert np.array_equal(inds.cpu().numpy(), want1)
    plain, _, ties2 = ext.furthest_point_sampling_ties(t, m1)      # without the cell lists
    assert torch.equal(plain, inds) and torch.equal(ties, ties2)
    ref_tie = _first_tie_reference(xyz, want1)
    assert np.array_equal(ties.cpu().numpy(), ref_tie), (ties.cpu().numpy(), ref_tie)
    if case in ("uniform", "room"):
        assert (ref_tie == m1).all()
    if case in ("duplicates", "identical"):
        assert (ref_tie < m1).all()
    head = np.ascontiguousarray(np.take_along_axis(xyz, want1[:, :, None].astype(np.int64), axis=1))
    for m, nh in ((m2, m1), (m2, m2 + 50), (100, 200)):      # heads of the sequence, and of heads
        sub = np.ascontiguousarray(head[:, :nh])
        got = ext.furthest_point_sampling_prefix(torch.from_numpy(sub).cuda(), m, ties).cpu().numpy()
        want = oracle_omp.furthest_point_sampling(sub, m)
        assert np.array_equal(got, want), (case, m, nh)
        for b in range(2):
            if ref_tie[b] >= nh:
                assert np.array_equal(got[b], np.arange(m))
    # no record of ties: the plain sampling
    got = ext.furthest_point_sampling_prefix(torch.from_numpy(head).cuda(), m2, None).cpu().numpy()
    assert np.array_equal(got, oracle_omp.furthest_point_sampling(head, m2))


@pytest.mark.gpu
def test_backbone_sampling_chain_matches_reference_chain(oracle_omp, synth):
    """compute_geometry's four samplings (the last three answered from the first run's tie
    record) against four independent oracle samplings, on a cloud large enough for the bucketed
    tier, with and without early ties."""
    import importlib
    backbone = importlib.import_module("3dioumatch_amd.votenet.backbone")
    net = backbone.Pointnet2Backbone(input_feature_dim=1).cuda()
    for seed, dup in ((21, 0), (22, 600)):
        xyz = synth.cloud_edge_cases(2, 12000, 3.0, seed=seed, near_origin=4, duplicates=dup) if dup \
            else synth.cloud_room(2, 12000, seed=seed)
        pc = np.concatenate([xyz, np.zeros((2, 12000, 1), np.float32)], axis=2).astype(np.float32)
        geo = net.compute_geometry(torch.from_numpy(pc).cuda())
        cur = np.ascontiguousarray(xyz, np.float32)
        for i, m in enumerate((2048, 1024, 512, 256), start=1):
            want = oracle_omp.furthest_point_sampling(cur, m)
            assert np.array_equal(geo["sa%d_inds" % i].cpu().numpy(), want), (seed, i)
            cur = np.ascontiguousarray(np.take_along_axis(cur, want[:, :, None].astype(np.int64), axis=1))


@pytest.mark.gpu
def test_module_chain_recognises_its_own_centroids(oracle_omp, synth):
    """The reference's call sequence -- every SA module samples the new_xyz the module before
    returned (backbone_module.py:97-112) -- through the drop-in modules: the second to fourth
    samplings are answered from the first run's tie record, found through the tensor's identity;
    a copy of the same tensor is sampled the ordinary way.  Centroids against the oracle chain."""
    import importlib
    backbone = importlib.import_module("3dioumatch_amd.votenet.backbone")
    utils = importlib.import_module("pointnet2.pointnet2_utils")
    ext = importlib.import_module("pointnet2._ext")
    net = backbone.Pointnet2Backbone(input_feature_dim=1).cuda().eval()
    xyz = synth.cloud_room(2, 12000, seed=31)
    pc = torch.from_numpy(np.concatenate([xyz, np.zeros((2, 12000, 1), np.float32)], axis=2)).cuda()
    with torch.no_grad():
        end_points = net(pc)
    cur = np.ascontiguousarray(xyz, np.float32)
    for i, m in enumerate((2048, 1024, 512, 256), start=1):
        want = oracle_omp.furthest_point_sampling(cur, m)
        cur = np.ascontiguousarray(np.take_along_axis(cur, want[:, :, None].astype(np.int64), axis=1))
        assert np.array_equal(end_points["sa%d_xyz" % i].cpu().numpy(), cur), i
        if i <= 2:
            assert np.array_equal(end_points["sa%d_inds" % i].cpu().numpy(), want)
    head = end_points["sa1_xyz"]
    ties = utils.head_record(head)
    assert ties is not None and int(ties.min()) == 2048
    assert utils.head_record(head.clone()) is None            # another tensor: no record
    got = ext.furthest_point_sampling(head.clone(), 1024)      # ... and the ordinary sampling agrees
    assert torch.equal(got.cpu(), torch.arange(1024, dtype=torch.int32).expand(2, -1))


def test_fps_bucket_tier_small_clouds_subprocess(synth):
    """Force the bucketed (spatially pruned) FPS tier onto SMALL clouds, where the oracle is
    cheap, including heavy ties, skipped points, all-skipped and n not a multiple of 64.
    The tier threshold is read once per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    code = r'''
import importlib, sys, numpy as np, torch
sys.path.insert(0, %r)
importlib.import_module("3dioumatch_amd")
ext = importlib.import_module("pointnet2._ext")
synth = importlib.import_module("3dioumatch_amd.synth")
from oracle.oracle import Oracle
o = Oracle(omp=True)
cases = []
for n, m, seed in [(64, 20, 1), (65, 30, 2), (700, 200, 3), (1000, 128, 4), (3000, 300, 5),
                   (5000, 512, 6), (12000, 256, 7)]:
    cases.append((synth.cloud_edge_cases(2, n, 1.0, seed=seed, near_origin=min(6, n // 8),
                                         duplicates=min(48, n // 6)), m))
cases.append((np.ones((1, 900, 3), np.float32), 40))                 # every point identical
two = np.ones((1, 2000, 3), np.float32); two[0, 1000:] = 5.0
cases.append((two, 64))                                               # two clusters of duplicates
cases.append((np.zeros((1, 300, 3), np.float32), 16))                 # every point skipped
cases.append((synth.cloud_room(2, 6000, seed=9), 400))                # points on planes
for xyz, m in cases:
    got = ext.furthest_point_sampling(torch.from_numpy(xyz).cuda(), m).cpu().numpy()
    want = o.furthest_point_sampling(xyz, m)
    assert np.array_equal(got, want), (xyz.shape, m, np.nonzero(got != want))
print("BUCKET_TIER_OK", len(cases))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PN2_FPS_BUCKET_MIN_N="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "BUCKET_TIER_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("case", ["aliasing", "overflow", "overflow_list_full", "clumps",
                                  "dense_hits", "negative", "ns_small", "ns_over_64",
                                  "ns_128_dense", "ns_128_sparse", "ns_over_128", "tiny_radius"])
def test_ballquery_cell_list_tier_edge_cases(ext, oracle_omp, synth, case):
    """The cell-list tier (n >= 4096, nsample <= 128) against the oracle where its special
    paths trigger: lattice aliasing (cloud wider than 32 cells), cell overflow (> 64 points
    in one cell with CONSECUTIVE indices -> the general path's bucket refinement), > 384 hits in a
    ball (histogram select of the nsample smallest indices), negative
    coordinates, small nsample, 64 < nsample <= 128 (second selection pass; balls with fewer /
    more than 128 hits), nsample > 128 (tier declines), radius so small that balls hold only
    their own centroid."""
    g = np.random.default_rng(11)
    b, n, m, r, ns = 2, 6000, 300, 0.2, 64
    xyz = synth.cloud_uniform(b, n, 2.0, seed=21)
    if case == "aliasing":
        xyz = synth.cloud_uniform(b, n, 30.0, seed=22)       # 150 cells wide: aliases 4-5x
        xyz[:, ::2] *= 0.1                                      # plus a dense core
    elif case == "overflow":
        xyz[0, :500] = 0.5 + g.random((500, 3), dtype=np.float32) * 0.05   # 500 points in one cell
    elif case == "overflow_list_full":
        xyz[1, :3000] = 0.7 + g.random((3000, 3), dtype=np.float32) * 0.05  # > cell + list capacity
    elif case == "clumps":
        for q in range(6):                                      # object-like dense clumps
            lo = 400 * q
            xyz[:, lo:lo + 400] = g.random(3, dtype=np.float32) * 1.5 + \
                g.random((b, 400, 3), dtype=np.float32) * 0.25
    elif case == "dense_hits":
        r, ns = 0.6, 64                                        # ~ 680 hits per ball
    elif case == "negative":
        xyz = xyz - 5.0
        xyz[1] *= -1.0
    elif case == "ns_small":
        ns = 5
    elif case == "ns_over_64":
        ns = 100
    elif case == "ns_128_dense":
        r, ns = 0.45, 128                                      # ~ 290 hits per ball
    elif case == "ns_128_sparse":
        r, ns = 0.25, 128                                      # ~ 50 hits: mostly padding
    elif case == "ns_over_128":
        r, ns = 0.45, 200
    elif case == "tiny_radius":
        r = 1e-4
    cen = xyz[:, g.permutation(n)[:m]].copy()
    cen[:, -1] = 1000.0                                        # empty ball -> zero row
    want = oracle_omp.ball_query(cen, xyz, r, ns)
    got = ext.ball_query(dev(cen), dev(xyz), r, ns).cpu().numpy()
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    assert np.all(got[:, -1] == 0)


@pytest.mark.parametrize("case", ["plain", "c0", "c12", "ns_100", "ns_200", "ns_256_dense", "dense_hits",
                                  "negative", "aliasing", "clumps", "ragged"])
def test_query_and_group_fused_cell_list(ext, oracle_omp, synth, case):
    """The fused ball query + gather kernel of the cell-list tier (n >= 4096): idx, gathered
    features (bit-exact) and relative xyz against the reference composition
    (pointnet2_utils.py:335-358) done with the oracle -- with 0 / 1 / 12 feature channels (more
    than 8 go through the channel-parallel gather), 64 < nsample <= 256, balls with more hits than the list holds
    (general path: adaptive cut / histogram select, then the gather), the lattice seam (negative coordinates put
    centroids in cells 0 and 31), aliasing, clumps, and n / m that are not multiples of 4."""
    g = np.random.default_rng(17)
    b, n, m, c, r, ns = 2, 6000, 300, 1, 0.2, 64
    xyz = synth.cloud_uniform(b, n, 2.0, seed=31)
    if case == "c0":
        c = 0
    elif case == "c12":
        c = 12
    elif case == "ns_100":
        ns, r = 100, 0.3
    elif case == "ns_200":      # 128 < nsample <= 256: the 512-entry hit list
        ns, r = 200, 0.4
    elif case == "ns_256_dense":  # more than 512 hits in most balls: the general path's select
        ns, r = 256, 0.65
    elif case == "dense_hits":
        r = 0.6
    elif case == "negative":
        xyz = xyz - 1.0
        xyz[1] *= -3.0
    elif case == "aliasing":
        xyz = synth.cloud_uniform(b, n, 30.0, seed=32)
        xyz[:, ::2] *= 0.1
    elif case == "clumps":
        for q in range(6):
            lo = 400 * q
            xyz[:, lo:lo + 400] = g.random(3, dtype=np.float32) * 1.5 + \
                g.random((b, 400, 3), dtype=np.float32) * 0.25
    elif case == "ragged":
        n, m, c, ns = 4099, 203, 3, 33
        xyz = synth.cloud_uniform(b, n, 1.7, seed=33)
    cen = xyz[:, g.permutation(n)[:m]].copy()
    cen[:, -1] = 1000.0                                        # empty ball -> zero row
    feats = g.standard_normal((b, c, n)).astype(np.float32) if c else None
    want_idx = oracle_omp.ball_query(cen, xyz, r, ns)
    gx = oracle_omp.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), want_idx)
    gx = gx - cen.transpose(0, 2, 1)[..., None]
    for normalize in (False, True):
        idx, out = ext.query_and_group(dev(cen), dev(xyz), dev(feats) if c else None, r, ns,
                                       normalize)
        assert np.array_equal(idx.cpu().numpy(), want_idx), np.argwhere(idx.cpu().numpy() != want_idx)[:5]
        out = out.cpu().numpy()
        assert out.shape == (b, 3 + c, m, ns)
        if c:
            assert np.array_equal(bits(out[:, 3:]), bits(oracle_omp.group_points(feats, want_idx)))
        ref_xyz = gx * (np.float32(1.0) / np.float32(r)) if normalize else gx
        assert np.array_equal(bits(out[:, :3]), bits(ref_xyz.astype(np.float32)))
    # and the fused result equals the unfused operators of the reference surface
    idx2 = ext.ball_query(dev(cen), dev(xyz), r, ns)
    assert torch.equal(idx2, idx)


def test_reference_surface_under_inference_mode(ext, oracle_omp, synth):
    """Tensors created under torch.inference_mode() keep no version counter: the per-cloud caches
    behind furthest_point_sampling / ball_query (cell lists, tie records) must step aside instead
    of raising, and the answers stay the oracle's.  Also an SA-style chain whose second sampling
    asks for MORE points than the cloud holds (the reference repeats picks)."""
    import pointnet2.pointnet2_utils as pu
    b, n, m, r, ns = 2, 6000, 300, 0.2, 32
    xyz_np = synth.cloud_uniform(b, n, 2.0, seed=51)
    with torch.inference_mode():
        xyz = dev(xyz_np)
        inds = ext.furthest_point_sampling(xyz, m)
        new_xyz = pu._sample_centroids(xyz, m)[0] if hasattr(pu, "_sample_centroids") and False else \
            ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        idx = ext.ball_query(new_xyz, xyz, r, ns)
        idx2 = ext.ball_query(new_xyz, xyz, r, ns)          # (would be a cache hit outside inference mode)
    want_inds = oracle_omp.furthest_point_sampling(xyz_np, m)
    assert np.array_equal(inds.cpu().numpy(), want_inds)
    cen = np.take_along_axis(xyz_np, want_inds[..., None].astype(np.int64), axis=1)
    want = oracle_omp.ball_query(cen, xyz_np, r, ns)
    assert np.array_equal(idx.cpu().numpy(), want) and torch.equal(idx, idx2)
    # more samples than points through the prefix entry point: plain sampling semantics
    small = dev(xyz_np[:, :100].copy())
    got = ext.furthest_point_sampling(small, 130).cpu().numpy()
    assert np.array_equal(got, oracle_omp.furthest_point_sampling(xyz_np[:, :100].copy(), 130))


def test_cell_list_cache_behind_the_reference_surface(ext, oracle_omp, synth):
    """The reference's layer calls furthest_point_sampling(xyz, npoint) and then
    ball_query(new_xyz, xyz, r, ns) on the same cloud (pointnet2_modules.py:236-250); the pybind
    surface cannot pass cell lists between the two, so _ext keeps them per cloud tensor.  A hit
    must be exactly what a fresh build answers, and a cloud that changed (in place, or a new tensor
    that may reuse the address) must never hit."""
    b, n, m, r, ns = 2, 9000, 256, 0.25, 32
    xyz_np = synth.cloud_uniform(b, n, 2.0, seed=91)
    xyz = dev(xyz_np)
    stats = ext.cache_stats
    inds = ext.furthest_point_sampling(xyz, m)
    new_xyz = ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    h0, m0, l0 = stats["hits"], stats["misses"], stats["left_by_sampling"]
    a = ext.ball_query(new_xyz, xyz, r, ns)                      # miss (or lists left by sampling)
    bq = ext.ball_query(new_xyz, xyz, r, ns)                     # hit
    assert stats["hits"] >= h0 + 1 and torch.equal(a, bq)
    assert np.array_equal(a.cpu().numpy(), oracle_omp.ball_query(new_xyz.cpu().numpy(), xyz_np, r, ns))
    # the radius is now known for clouds of this size: sampling a NEW cloud leaves its lists behind
    xyz2_np = synth.cloud_uniform(b, n, 2.0, seed=92)
    xyz2 = dev(xyz2_np)
    inds2 = ext.furthest_point_sampling(xyz2, m)
    assert stats["left_by_sampling"] == l0 + 1
    assert np.array_equal(inds2.cpu().numpy(), oracle_omp.furthest_point_sampling(xyz2_np, m))
    new2 = ext.gather_points(xyz2.transpose(1, 2).contiguous(), inds2).transpose(1, 2).contiguous()
    h1, m1 = stats["hits"], stats["misses"]
    c = ext.ball_query(new2, xyz2, r, ns)
    assert stats["hits"] == h1 + 1 and stats["misses"] == m1
    assert np.array_equal(c.cpu().numpy(), oracle_omp.ball_query(new2.cpu().numpy(), xyz2_np, r, ns))
    # in-place change: the version moves, the stale lists must not be used
    xyz2.mul_(0.5)
    m2 = stats["misses"]
    d = ext.ball_query(new2, xyz2, r, ns)
    assert stats["misses"] == m2 + 1
    assert np.array_equal(d.cpu().numpy(), oracle_omp.ball_query(new2.cpu().numpy(), xyz2_np * np.float32(0.5), r, ns))
    # another radius on the same cloud: its own lists
    e = ext.ball_query(new2, xyz2, 0.4, ns)
    assert np.array_equal(e.cpu().numpy(), oracle_omp.ball_query(new2.cpu().numpy(), xyz2_np * np.float32(0.5), 0.4, ns))
    # entries die with their cloud
    key = id(xyz2)
    assert key in ext._LISTS
    del xyz2
    import gc
    gc.collect()
    assert key not in ext._LISTS


@pytest.mark.parametrize("case", ["uniform", "skipped_points", "negative", "ns_100", "small_m"])
def test_cell_lists_from_fps_and_standalone(ext, oracle_omp, synth, case):
    """Cell lists as an object (include/pn2_hip.h pn2_grid_*): built by the two-kernel build or
    left behind by the furthest-point-sampling kernel, then queried -- same indices as the oracle
    (and as the self-contained operators), same FPS indices as the plain sampling.  Includes
    points the sampling skips (|p|^2 <= 1e-3: never sampled, but inside balls) and the lattice
    seam."""
    g = np.random.default_rng(23)
    b, n, m, r, ns = 2, 12000, 600, 0.2, 64
    xyz = synth.cloud_uniform(b, n, 2.2, seed=41)
    if case == "skipped_points":
        xyz[:, 100:400] = (g.random((b, 300, 3), dtype=np.float32) - 0.5) * 0.05  # |p|^2 <= 1e-3
    elif case == "negative":
        xyz = xyz - 1.1
    elif case == "ns_100":
        ns, r = 100, 0.3
    elif case == "small_m":
        m = 37
    d_xyz = dev(xyz)
    assert ext.grid_supported(b, n)
    inds, lists = ext.furthest_point_sampling_with_grid(d_xyz, m, r)
    assert lists is not None
    assert torch.equal(inds, ext.furthest_point_sampling(d_xyz, m))
    assert np.array_equal(inds.cpu().numpy(), oracle_omp.furthest_point_sampling(xyz, m))
    new_xyz = ext.gather_points(d_xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    want = oracle_omp.ball_query(new_xyz.cpu().numpy(), xyz, r, ns)
    standalone = ext.build_grid(d_xyz, r)
    for which in (lists, standalone):
        assert np.array_equal(ext.ball_query_prebuilt(new_xyz, d_xyz, r, ns, which).cpu().numpy(), want)
    feats = torch.randn(b, 3, n, device=DEV)
    idx_a, out_a = ext.query_and_group(new_xyz, d_xyz, feats, r, ns, True)
    idx_b, out_b = ext.query_and_group(new_xyz, d_xyz, feats, r, ns, True, None, lists)
    assert torch.equal(idx_a, idx_b) and torch.equal(out_a, out_b)
    assert np.array_equal(idx_b.cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        ext.ball_query_prebuilt(new_xyz, d_xyz, r * 2, ns, lists)  # lists are per radius


def test_sa_module_uses_cell_lists_of_the_sampling_kernel(ext, oracle_omp, synth):
    """PointnetSAModuleVotes on a large cloud: the layer samples and groups through the cell
    lists left behind by the sampling kernel (one fused query + gather kernel) and returns exactly
    what the precomputed-index path returns."""
    mods = __import__("importlib").import_module("pointnet2.pointnet2_modules")
    torch.manual_seed(0)
    sa = mods.PointnetSAModuleVotes(npoint=512, radius=0.2, nsample=32, mlp=[1, 16, 16, 32],
                                    use_xyz=True, normalize_xyz=True).to(DEV).eval()
    xyz = dev(synth.cloud_uniform(2, 9000, 2.0, seed=8))
    feats = torch.randn(2, 1, 9000, device=DEV)
    with torch.no_grad():
        new_xyz, out, inds = sa(xyz, feats)
        want_inds = ext.furthest_point_sampling(xyz, 512)
        assert torch.equal(inds, want_inds)
        ball = ext.ball_query(new_xyz, xyz, 0.2, 32)
        new_xyz2, out2, _ = sa(xyz, feats, want_inds, ball, new_xyz)
    assert torch.equal(new_xyz, new_xyz2) and torch.equal(out, out2)


def test_lhs_nms_vs_reference_golden_and_oracle(oracle, synth):
    """Device pseudo-label NMS == the reference's numpy lhs_3d_faster_samecls (committed vectors)
    and == the oracle on a batch of crowded random scenes."""
    import importlib
    nms = importlib.import_module("3dioumatch_amd.votenet.pseudo_nms")
    g = golden("lhs_nms_ref.npz")
    for k in range(int(g["num_cases"])):
        get = lambda name: g["c%d_%s" % (k, name)]  # noqa: E731
        picked = nms.lhs_nms_samecls_gpu(
            dev(get("center")[None]), torch.from_numpy(get("size").astype(np.float64)[None]).cuda(),
            torch.from_numpy(get("heading").astype(np.float64)[None]).cuda(), dev(get("score")[None]),
            torch.from_numpy(get("cls")[None]).cuda(), float(get("thresh")), bool(get("old")))
        np.testing.assert_array_equal(picked[0].cpu().numpy().astype(np.int32), get("pick"))
    rng = np.random.default_rng(5)
    s, n = 12, 64
    clump = rng.uniform(-2, 2, (s, 4, 3))
    center = (clump[np.arange(s)[:, None], rng.integers(0, 4, (s, n))] +
              rng.normal(0, 0.2, (s, n, 3))).astype(np.float32)
    size = rng.uniform(0.3, 1.5, (s, n, 3)).astype(np.float32).astype(np.float64)
    heading = np.zeros((s, n))
    heading[s // 2:] = rng.uniform(-3, 3, (s - s // 2, n))
    score = rng.uniform(0, 1, (s, n)).astype(np.float32)
    score[:, 10:14] = score[:, 9:10]  # ties
    cls = rng.integers(0, 2, (s, n)).astype(np.int64)
    got = nms.lhs_nms_samecls_gpu(dev(center), torch.from_numpy(size).cuda(),
                                  torch.from_numpy(heading).cuda(), dev(score),
                                  torch.from_numpy(cls).cuda(), 0.25).cpu().numpy()
    for i in range(s):
        aabb = oracle.camera_aabb(center[i], size[i], heading[i])
        want = oracle.lhs_nms_samecls(aabb, score[i], cls[i], 0.25)
        np.testing.assert_array_equal(got[i].astype(np.int32), want)
    assert 0 < got.sum() < s * n


def test_eval_nms_vs_reference_golden(oracle):
    """Device NMS of the evaluation path (n up to 256, one workgroup per scene) == the reference's
    numpy nms_3d_faster[_samecls]; batched call == per-scene oracle."""
    import importlib
    nms = importlib.import_module("3dioumatch_amd.votenet.pseudo_nms")
    g = golden("lhs_nms_ref.npz")
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).cuda()  # noqa: E731
    for k in range(int(g["num_eval_cases"])):
        get = lambda name: g["e%d_%s" % (k, name)]  # noqa: E731
        picked = nms.nms3d_aabb_gpu(dev(get("center")[None]), t64(get("size")[None]),
                                    t64(get("heading")[None]), dev(get("score")[None]),
                                    torch.from_numpy(get("cls")[None]).cuda(), float(get("thresh")),
                                    bool(get("old")), bool(get("same")))
        np.testing.assert_array_equal(picked[0].cpu().numpy().astype(np.int32), get("pick"))
    for s, n in ((5, 256), (2, 257), (3, 700), (1, 1024)):
        _check_aabb_nms_random(nms, oracle, s, n)
    with pytest.raises(RuntimeError):
        nms.nms3d_aabb_gpu(torch.zeros(1, 1025, 3, device=DEV), torch.ones(1, 1025, 3, device=DEV).double(),
                           torch.zeros(1, 1025, device=DEV).double(), torch.zeros(1, 1025, device=DEV),
                           torch.zeros(1, 1025, device=DEV).long(), 0.25)


def _check_aabb_nms_random(nms, oracle, s, n):
    t64 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).to(DEV)  # noqa: E731
    rng = np.random.default_rng(6 + n)
    clump = rng.uniform(-2, 2, (s, 6, 3))
    center = (clump[np.arange(s)[:, None], rng.integers(0, 6, (s, n))] +
              rng.normal(0, 0.2, (s, n, 3))).astype(np.float32)
    size = rng.uniform(0.3, 1.5, (s, n, 3)).astype(np.float32).astype(np.float64)
    heading = rng.uniform(-3, 3, (s, n))
    score = rng.uniform(0, 1, (s, n)).astype(np.float32)
    score[:, 100:104] = score[:, 99:100]
    cls = rng.integers(0, 3, (s, n)).astype(np.int64)
    for same in (True, False):
        got = nms.nms3d_aabb_gpu(dev(center), t64(size), t64(heading), dev(score),
                                 torch.from_numpy(cls).cuda(), 0.25, False, same).cpu().numpy()
        for i in range(s):
            aabb = oracle.camera_aabb(center[i], size[i], heading[i])
            want = oracle.nms3d_aabb(aabb, score[i], cls[i], 0.25, False, same)
            np.testing.assert_array_equal(got[i].astype(np.int32), want)


def test_multi_copy_one_launch():
    """pn2_multi_copy: many device-to-device copies from a pointer table in one launch -- sizes
    that are no multiple of 16 bytes, a misaligned pair (byte path), an empty row."""
    import importlib
    L = importlib.import_module("3dioumatch_amd._lib")
    g = torch.Generator().manual_seed(0)
    sizes = [1, 3, 4, 17, 1000, 32768 // 4, 32768 // 4 + 5, 1234567, 0]
    srcs = [torch.randn(max(n, 1), generator=g).to(DEV)[:n] for n in sizes]
    dsts = [torch.full((max(n, 1),), -1.0, device=DEV)[:n] for n in sizes]
    big_src = torch.randn(4099, generator=g).to(DEV)
    big_dst = torch.zeros(4099, device=DEV)
    srcs.append(big_src[1:4097])   # 4 bytes off 16-byte alignment
    dsts.append(big_dst[2:4098])
    rows = [[s.data_ptr(), d.data_ptr(), s.numel() * 4] for s, d in zip(srcs, dsts)]
    table = torch.tensor(rows, dtype=torch.int64, device=DEV)
    L.check(L.lib.pn2_multi_copy(len(rows), table.data_ptr(), max(r[2] for r in rows),
                                 torch.cuda.current_stream().cuda_stream), "pn2_multi_copy")
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)
    assert float(big_dst[0]) == 0.0 and float(big_dst[1]) == 0.0 and float(big_dst[-1]) == 0.0
