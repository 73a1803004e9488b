"""Seeded synthetic inputs shared by tests/ and bench.py.

Everything is generated on the CPU with numpy's PCG64 so that the CPU oracle and the GPU
path see bit-identical inputs.  No dataset is read; shapes follow the ScanNet / SUN RGB-D
configurations of the reference (pretrain.py:45-52, backbone_module.py:35-72).
"""
import math

import numpy as np


def cube_side(n, radius, nsample):
    """Side L of the cube for which a uniform cloud of n points has an expected in-ball count
    of nsample: about half of the balls fill up (the realistic regime)."""
    return (n * (4.0 / 3.0) * math.pi * radius ** 3 / nsample) ** (1.0 / 3.0)


def cloud_uniform(b, n, side, seed=0):
    """Cloud U(L): xyz ~ U[0, side)^3, float32 (B, N, 3)."""
    g = np.random.default_rng(seed)
    return (g.random((b, n, 3), dtype=np.float32) * np.float32(side)).astype(np.float32)


def cloud_room(b, n, seed=0):
    """Cloud R: points on the six faces of a 6 x 5 x 3 room with 1 cm jitter."""
    g = np.random.default_rng(seed)
    ext = np.array([6.0, 5.0, 3.0], np.float32)
    p = g.random((b, n, 3), dtype=np.float32) * ext
    face = g.integers(0, 6, (b, n))
    for axis in range(3):
        lo = face == 2 * axis
        hi = face == 2 * axis + 1
        p[..., axis][lo] = 0.0
        p[..., axis][hi] = ext[axis]
    p += g.normal(0, 0.01, p.shape).astype(np.float32)
    return p.astype(np.float32)


def cloud_edge_cases(b, n, side, seed=0, near_origin=16, duplicates=64):
    """Cloud E (parity only): U(L) with `near_origin` points inside |p|^2 <= 1e-3 (the FPS
    skip branch), `duplicates` exact duplicate points (FPS ties) and point 0 of each cloud
    moved far away (a centroid placed there has an isolated ball)."""
    g = np.random.default_rng(seed)
    p = cloud_uniform(b, n, side, seed)
    for bi in range(b):
        where = g.choice(np.arange(1, n), near_origin + 2 * duplicates, replace=False)
        no = where[:near_origin]
        p[bi, no] = (g.random((near_origin, 3), dtype=np.float32) * 0.03 - 0.015) * 0.5
        src = where[near_origin:near_origin + duplicates]
        dst = where[near_origin + duplicates:]
        p[bi, dst] = p[bi, src]
        p[bi, 0] = np.float32(side) * 3.0
    return p.astype(np.float32)


def boxes_pair(n, seed=0, axis_aligned=False):
    """Boxes a and a perturbed permutation b of them, rows (x,y,z,dx,dy,dz,heading)."""
    g = np.random.default_rng(seed)
    a = np.concatenate([g.random((n, 3)) * 4.0, g.random((n, 3)) * 1.5 + 0.2,
                        (g.random((n, 1)) - 0.5) * 2.0 * np.pi], axis=1).astype(np.float32)
    b = a[g.permutation(n)].copy()
    b[:, :6] += g.normal(0, 0.15, (n, 6)).astype(np.float32)
    b[:, 3:6] = np.maximum(b[:, 3:6], 1e-6)
    if axis_aligned:
        a[:, 6] = 0.0
        b[:, 6] = 0.0
    return a, b


def boxes_scored(n, seed=0, spread=6.0):
    """n boxes sorted by a random score (descending) for the NMS tests."""
    g = np.random.default_rng(seed)
    boxes = np.concatenate([g.random((n, 3)) * spread, g.random((n, 3)) * 1.5 + 0.3,
                            (g.random((n, 1)) - 0.5) * 2.0 * np.pi], axis=1).astype(np.float32)
    scores = g.random(n).astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    return boxes[order].copy(), scores[order].copy()


def box_kats():
    """Known-answer boxes: the axis-aligned cases asserted by the reference's only
    assertion-bearing smoke block (utils/metric_util.py:126-144: unit cube vs 2-cube -> 1/8,
    disjoint -> 0) plus identical / contained / touching / degenerate pairs."""
    a = np.array([
        [0.5, 0.5, 0.5, 1, 1, 1, 0],      # unit cube [0,1]^3
        [0.5, 0.5, 0.5, 1, 1, 1, 0],
        [0.0, 0.0, 0.0, 2, 2, 2, 0.3],    # identical rotated boxes
        [0.0, 0.0, 0.0, 4, 4, 4, 0.0],    # contains a small rotated box
        [0.0, 0.0, 0.0, 1, 1, 1, 0.0],    # touching faces
        [0.0, 0.0, 0.0, 1, 1, 1, 0.0],    # z-disjoint
        [0.0, 0.0, 0.0, 1e-6, 1e-6, 1e-6, 0.0],  # near zero size
        [-1000.0, -1000.0, -1000.0, 1e-6, 1e-6, 1e-6, 0.0],  # masked-GT sentinel, loss_helper_iou.py:58
    ], np.float32)
    b = np.array([
        [1.0, 1.0, 1.0, 2, 2, 2, 0],      # 2-cube [0,2]^3 -> IoU 1/8
        [5.0, 5.0, 5.0, 1, 1, 1, 0],      # disjoint -> 0
        [0.0, 0.0, 0.0, 2, 2, 2, 0.3],
        [0.2, -0.1, 0.0, 1, 0.5, 1, 0.7],
        [1.0, 0.0, 0.0, 1, 1, 1, 0.0],
        [0.0, 0.0, 5.0, 1, 1, 1, 0.0],
        [0.0, 0.0, 0.0, 1, 1, 1, 0.0],
        [1.0, 1.0, 1.0, 1, 1, 1, 0.4],
    ], np.float32)
    return a, b
