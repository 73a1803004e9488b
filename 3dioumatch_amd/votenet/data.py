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


def make_semi_batch(labeled, unlabeled, num_points, config, seed=0, num_objects=12, device="cpu"):
    """A stage-2 batch as train.py:321-325 collates it: `labeled` + `unlabeled` scenes.
    point_clouds / ema_point_clouds / the augmentation keys / supervised_mask / scan_idx have
    batch labeled+unlabeled (labeled first), the label tensors batch `labeled`.  The student sees
    the flipped / rotated / scaled cloud (and, for labeled scenes, labels in that frame), the EMA
    teacher an independent, un-augmented subsample of the same scene
    (scannet_ssl_dataset.py:90-128,177)."""
    total = labeled + unlabeled
    base = make_batch(total, num_points, config, seed=seed, num_objects=num_objects)
    g = np.random.default_rng(seed + 7919)
    pc = base['point_clouds'].numpy()
    ema_pc = np.stack([pc[b][g.permutation(num_points)] for b in range(total)])
    flip_x = g.integers(0, 2, total).astype(np.int64)
    flip_y = g.integers(0, 2, total).astype(np.int64)
    rot_angle = ((g.random(total) - 0.5) * (np.pi / 18)).astype(np.float32)  # +-5 degrees
    scale = (g.random((total, 1, 3)) * 0.3 + 0.85).astype(np.float32)
    c, s = np.cos(rot_angle), np.sin(rot_angle)
    rot_mat = np.zeros((total, 3, 3), np.float32)
    rot_mat[:, 0, 0], rot_mat[:, 0, 1], rot_mat[:, 1, 0], rot_mat[:, 1, 1], rot_mat[:, 2, 2] = \
        c, -s, s, c, 1.0

    def to_student(xyz, b):  # (n,3) teacher frame -> student frame
        out = xyz.copy()
        if flip_x[b]:
            out[:, 0] = -out[:, 0]
        if flip_y[b]:
            out[:, 1] = -out[:, 1]
        return (out @ rot_mat[b].T) * scale[b]

    student_pc = pc.copy()
    center = base['center_label'].numpy().copy()
    size_res = base['size_residual_label'].numpy().copy()
    vote = base['vote_label'].numpy().copy()
    for b in range(total):
        student_pc[b, :, :3] = to_student(pc[b, :, :3], b)
        if student_pc.shape[2] > 3:
            student_pc[b, :, 3] = student_pc[b, :, 2] - np.percentile(student_pc[b, :, 2], 0.99)
        if b < labeled:
            k = int(base['box_label_mask'][b].sum())
            center[b, :k] = to_student(center[b, :k], b)
            cls = base['size_class_label'][b, :k].numpy()
            size_res[b, :k] = (config.mean_size_arr[cls] + size_res[b, :k]) * scale[b] \
                - config.mean_size_arr[cls]
            target = pc[b, :, :3] + vote[b, :, :3]          # voted centre, teacher frame
            moved = to_student(target, b) - student_pc[b, :, :3]
            vote[b] = np.tile(moved * base['vote_label_mask'][b].numpy()[:, None], (1, 3))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    lab = slice(0, labeled)
    out = {
        'point_clouds': t(student_pc), 'ema_point_clouds': t(ema_pc),
        'flip_x_axis': t(flip_x), 'flip_y_axis': t(flip_y), 'rot_mat': t(rot_mat),
        'rot_angle': t(rot_angle), 'scale': t(scale),
        'supervised_mask': t(np.array([1] * labeled + [0] * unlabeled, np.int64)),
        'scan_idx': torch.arange(total, dtype=torch.int64, device=device),
        'center_label': t(center[lab]), 'size_residual_label': t(size_res[lab]),
        'vote_label': t(vote[lab]),
    }
    for k in ('heading_class_label', 'heading_residual_label', 'size_class_label', 'sem_cls_label',
              'box_label_mask', 'vote_label_mask'):
        out[k] = base[k][lab].to(device)
    return out
