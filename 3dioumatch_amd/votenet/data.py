"""Synthetic, seeded train batches with the key set and shapes of the reference's ScanNet
loader (scannet/scannet_ssl_dataset.py:157-182).  No dataset is read.

Scenes are `room`-like clouds with up to MAX_NUM_OBJ axis-aligned (ScanNet) or oriented
(SUN RGB-D) GT boxes; points inside a box carry a vote towards its centre.
"""
import numpy as np
import torch

MAX_NUM_OBJ = 64  # scannet_ssl_dataset.py:20


def make_batch(batch_size, num_points, config, seed=0, num_objects=12, device="cpu",
               with_height=True):
    g = np.random.default_rng(seed)
    ext = np.array([6.0, 5.0, 3.0], np.float32)
    pc = np.zeros((batch_size, num_points, 4 if with_height else 3), np.float32)
    center = np.zeros((batch_size, MAX_NUM_OBJ, 3), np.float32)
    size_cls = np.zeros((batch_size, MAX_NUM_OBJ), np.int64)
    size_res = np.zeros((batch_size, MAX_NUM_OBJ, 3), np.float32)
    head_cls = np.zeros((batch_size, MAX_NUM_OBJ), np.int64)
    head_res = np.zeros((batch_size, MAX_NUM_OBJ), np.float32)
    sem_cls = np.zeros((batch_size, MAX_NUM_OBJ), np.int64)
    mask = np.zeros((batch_size, MAX_NUM_OBJ), np.float32)
    vote = np.zeros((batch_size, num_points, 9), np.float32)
    vote_mask = np.zeros((batch_size, num_points), np.int64)
    for b in range(batch_size):
        k = int(g.integers(max(1, num_objects // 2), num_objects + 1))
        cls = g.integers(0, config.num_size_cluster, k)
        sz = config.mean_size_arr[cls] + g.normal(0, 0.05, (k, 3)).astype(np.float32)
        sz = np.maximum(sz, 0.1)
        ctr = g.random((k, 3), dtype=np.float32) * (ext - sz) + sz / 2
        # half the points on the room shell, half inside objects
        pts = g.random((num_points, 3), dtype=np.float32) * ext
        face = g.integers(0, 6, num_points)
        for axis in range(3):
            pts[face == 2 * axis, axis] = 0.0
            pts[face == 2 * axis + 1, axis] = ext[axis]
        owner = g.integers(0, k, num_points)
        inside = g.random(num_points) < 0.5
        local = (g.random((num_points, 3), dtype=np.float32) - 0.5) * sz[owner]
        pts[inside] = (ctr[owner] + local)[inside]
        pc[b, :, :3] = pts
        if with_height:
            pc[b, :, 3] = pts[:, 2] - np.percentile(pts[:, 2], 0.99)
        center[b, :k] = ctr
        size_cls[b, :k] = cls
        size_res[b, :k] = sz - config.mean_size_arr[cls]
        sem_cls[b, :k] = cls % config.num_class
        mask[b, :k] = 1
        if config.num_heading_bin > 1:
            head_cls[b, :k] = g.integers(0, config.num_heading_bin, k)
            head_res[b, :k] = (g.random(k) - 0.5) * (np.pi / config.num_heading_bin)
        v = (ctr[owner] - pts) * inside[:, None]
        vote[b] = np.tile(v, (1, 3))
        vote_mask[b] = inside.astype(np.int64)
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return {
        'point_clouds': t(pc), 'center_label': t(center), 'heading_class_label': t(head_cls),
        'heading_residual_label': t(head_res), 'size_class_label': t(size_cls),
        'size_residual_label': t(size_res), 'sem_cls_label': t(sem_cls), 'box_label_mask': t(mask),
        'vote_label': t(vote), 'vote_label_mask': t(vote_mask),
        'supervised_mask': torch.ones(batch_size, dtype=torch.int64, device=device),
        'scan_idx': torch.arange(batch_size, dtype=torch.int64, device=device),
    }
