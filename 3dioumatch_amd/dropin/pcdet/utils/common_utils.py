"""The one helper of OpenPCDet's common_utils that the iou3d_nms wrappers need
(reference OpenPCDet/pcdet/utils/common_utils.py:14-17)."""
import numpy as np
import torch


def check_numpy_to_torch(x):
    """numpy array -> (float32 CPU tensor, True); anything else -> (x, False)."""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False
