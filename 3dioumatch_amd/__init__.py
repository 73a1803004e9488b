"""3dioumatch_amd -- MI355X (gfx950) native hot path of 3DIoUMatch's VoteNet-IoU detector:
PointNet++ set-abstraction operators and rotated 3-D IoU / NMS, as hand-written HIP kernels
behind a C ABI (include/pn2_hip.h, include/iou3d_hip.h).

The directory name is not a Python identifier; import it with
    importlib.import_module("3dioumatch_amd")
Importing it puts `3dioumatch_amd/dropin` on sys.path, which provides the reference's module
names unchanged:
    pointnet2._ext                              (pointnet2/_ext_src/src/bindings.cpp:11-24)
    pointnet2.pointnet2_utils / pointnet2_modules / pytorch_utils
    pcdet.ops.iou3d_nms.iou3d_nms_cuda          (iou3d_nms_api.cpp:11-17)
    pcdet.ops.iou3d_nms.iou3d_nms_utils
so that models/backbone_module.py, models/voting_module.py and models/loss_helper_iou.py of the
reference run unchanged on PyTorch-ROCm.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
DROPIN = os.path.join(_HERE, "dropin")
if DROPIN not in sys.path:
    sys.path.insert(0, DROPIN)


def build(force=False):
    """Compile the HIP shared library in-tree (hipcc --offload-arch=gfx950)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_3dioumatch_amd_build",
                                                  os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force)
