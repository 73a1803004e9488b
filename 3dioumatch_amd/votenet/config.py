"""Dataset constants the detector heads and losses need (class / heading-bin / size-cluster
counts, mean box sizes, class<->angle/size decoding).

Mirrors the interface of the reference's ScannetDatasetConfig (scannet/model_util_scannet.py:19-83:
18 classes, 1 heading bin -> heading == 0, 18 size clusters) and SunrgbdDatasetConfig
(sunrgbd/model_util_sunrgbd.py:19-129: 10 classes, 12 heading bins, 10 size clusters).  The
reference loads its mean sizes from dataset files; there is no dataset here, so the mean sizes
are SYNTHETIC (seeded), which is all the throughput metric needs.
"""
import numpy as np
import torch


class DatasetConfig(object):
    # the fused box decode + jitter kernel (votenet_bbox_jitter, detector._bbox_jitter_fused)
    # hard-codes THIS class's class2angle_gpu; a subclass that overrides the decoding must set it False
    fused_heading_decode = True

    def __init__(self, num_class, num_heading_bin, num_size_cluster, seed=0):
        self.num_class = num_class
        self.num_heading_bin = num_heading_bin
        self.num_size_cluster = num_size_cluster
        g = np.random.default_rng(seed)
        self.mean_size_arr = g.uniform(0.3, 1.8, (num_size_cluster, 3)).astype(np.float32)
        self._mean_size_dev = {}

    def mean_size(self, device):
        key = str(device)
        if key not in self._mean_size_dev:
            self._mean_size_dev[key] = torch.from_numpy(self.mean_size_arr).to(device)
        return self._mean_size_dev[key]

    def class2angle_gpu(self, pred_cls, residual, to_label_format=True):
        """heading class + residual -> angle.  One bin (ScanNet): always 0
        (model_util_scannet.py:50-54); otherwise class*2pi/N + residual wrapped to (-pi, pi]
        (model_util_sunrgbd.py class2angle_gpu)."""
        if self.num_heading_bin == 1:
            return torch.zeros(pred_cls.shape, device=pred_cls.device)
        angle_per_class = 2 * np.pi / float(self.num_heading_bin)
        angle = pred_cls.float() * angle_per_class + residual
        if to_label_format:
            angle = angle - (angle > np.pi).float() * (2 * np.pi)
        return angle

    def class2angle_f64(self, pred_cls, residual):
        """The numpy decoding used by the pseudo-label filter (class2angle in
        model_util_scannet.py:60-64 / model_util_sunrgbd.py:110-120): float64."""
        if self.num_heading_bin == 1:
            return torch.zeros(pred_cls.shape, dtype=torch.float64, device=pred_cls.device)
        angle = pred_cls.double() * (2 * np.pi / float(self.num_heading_bin)) + residual.double()
        return angle - 2 * np.pi * (angle > np.pi).double()

    def angle2class_gpu(self, angle):
        """angle -> (heading class, residual) with class*(2pi/N) + residual == angle
        (model_util_sunrgbd.py:62-78)."""
        if self.num_heading_bin == 1:
            return torch.zeros_like(angle, dtype=torch.int32), angle
        angle = angle % (2 * np.pi)
        per = 2 * np.pi / float(self.num_heading_bin)
        shifted = (angle + per / 2) % (2 * np.pi)
        class_id = (shifted / per).int()
        return class_id, shifted - (class_id * per + per / 2)

    # numpy forms, as the reference's dataset configs offer them (used by its host-side filter)
    def class2angle(self, pred_cls, residual, to_label_format=True):
        if self.num_heading_bin == 1:
            return np.zeros(np.shape(pred_cls))
        angle = pred_cls * (2 * np.pi / float(self.num_heading_bin)) + residual
        if to_label_format:
            angle = angle - 2 * np.pi * (angle > np.pi)
        return angle

    def class2size(self, pred_cls, residual):
        return self.mean_size_arr[pred_cls, :] + residual

    def class2size_gpu(self, pred_cls, residual):
        """size class + residual -> box size (model_util_scannet.py:56-58)."""
        return self.mean_size(residual.device)[pred_cls, :] + residual


def scannet_config():
    return DatasetConfig(num_class=18, num_heading_bin=1, num_size_cluster=18, seed=18)


def sunrgbd_config():
    return DatasetConfig(num_class=10, num_heading_bin=12, num_size_cluster=10, seed=10)
