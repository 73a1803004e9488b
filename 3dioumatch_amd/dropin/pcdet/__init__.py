"""Minimal `pcdet` namespace: only the iou3d_nms operator package that 3DIoUMatch imports
(utils/box_util.py:17-20 of the reference).  Nothing else of OpenPCDet is provided."""
__version__ = "0.3.0+3dioumatch_amd"
