"""CPU tests of the drop-in boundary: the C-ABI shared library loads, exports every symbol the
headers in include/ declare (no compute without a GPU), every declaration cites the reference
interface it replaces, and the Python drop-in modules expose the reference's pybind surfaces."""
import ctypes
import importlib
import os
import re

import pytest

from conftest import ROOT, load_pkg

HEADERS = ["pn2_hip.h", "iou3d_hip.h", "mlp_hip.h", "lhs_hip.h", "loss_hip.h"]
DECL = re.compile(r"^(?:int|size_t|const char \*)\s*\*?\s*((?:pn2|iou3d|mlp|lhs|votenet)_\w+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    return text, DECL.findall(text)


def test_headers_declare_expected_entry_points():
    names = sum((declared(h)[1] for h in HEADERS), [])
    for must in ["pn2_furthest_point_sampling", "pn2_gather_points", "pn2_gather_points_grad",
                 "pn2_ball_query", "pn2_group_points", "pn2_group_points_grad", "pn2_three_nn",
                 "pn2_three_interpolate", "pn2_three_interpolate_grad",
                 "iou3d_boxes_overlap_bev", "iou3d_boxes_iou_bev", "iou3d_nms_mask",
                 "iou3d_nms_normal_mask", "iou3d_nms", "iou3d_boxes_iou_bev_cpu"]:
        assert must in names, must


@pytest.mark.parametrize("header", HEADERS)
def test_library_exports_every_declared_symbol(header):
    pkg = load_pkg()
    so = os.path.join(os.path.dirname(pkg.__file__), "lib3dioumatch_hip.so")
    assert os.path.exists(so), "build with python 3dioumatch_amd/build.py"
    lib = ctypes.CDLL(so)
    _, names = declared(header)
    assert names
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n


@pytest.mark.parametrize("header", HEADERS)
def test_every_declaration_cites_the_reference(header):
    text, names = declared(header)
    for mt in DECL.finditer(text):
        n = mt.group(1)
        comment = text[:mt.start()].rsplit("/*", 1)[-1]
        assert re.search(r"\.(cpp|cu|py|h):\d+", comment), "no reference file:line above %s" % n


def test_ctypes_binding_covers_the_headers():
    load_pkg()
    L = importlib.import_module("3dioumatch_amd._lib")
    names = sum((declared(h)[1] for h in HEADERS), [])
    assert set(names) == set(L.EXPORTS)


def test_dropin_surfaces_match_reference_pybind_names():
    load_pkg()
    ext = importlib.import_module("pointnet2._ext")
    # pointnet2/_ext_src/src/bindings.cpp:11-24
    for n in ["gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
              "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
              "group_points_grad"]:
        assert callable(getattr(ext, n))
    cu = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_cuda")
    # OpenPCDet/pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17
    for n in ["boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu",
              "boxes_iou_bev_cpu"]:
        assert callable(getattr(cu, n))
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    # OpenPCDet/pcdet/ops/iou3d_nms/iou3d_nms_utils.py:12-116
    for n in ["boxes_bev_iou_cpu", "boxes_iou_bev", "boxes_iou3d_gpu", "nms_gpu", "nms_normal_gpu"]:
        assert callable(getattr(ut, n))


def test_no_cpu_fallback_in_device_entry_points():
    """Device ops must refuse CPU tensors (the reference raises 'CPU not supported',
    e.g. ball_query.cpp:33); nothing silently computes on the host."""
    import torch
    load_pkg()
    ext = importlib.import_module("pointnet2._ext")
    xyz = torch.rand(1, 32, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.ball_query(xyz[:, :4].contiguous(), xyz, 0.2, 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.furthest_point_sampling(xyz, 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.three_nn(xyz, xyz)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.group_points(torch.rand(1, 2, 32), torch.zeros(1, 4, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="float tensor"):
        ext.furthest_point_sampling(xyz.double(), 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        ext.furthest_point_sampling(xyz.transpose(1, 2), 4)
    cu = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_cuda")
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        cu.boxes_overlap_bev_gpu(torch.rand(2, 7), torch.rand(2, 7), torch.zeros(2, 2))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under 3dioumatch_amd/ may reference it."""
    pkg_dir = os.path.join(ROOT, "3dioumatch_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert "liboracle" not in text and "pn2o_" not in text and "iou3do_" not in text, f


def test_boxes_bev_iou_cpu_matches_reference_golden():
    """The API-mandated CPU operator (iou3d_cpu.cpp:232-252) of the PRODUCT, bit-exact against
    vectors produced by the reference's compiled code."""
    import numpy as np
    from conftest import golden
    load_pkg()
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    g = golden("iou_bev_cpu_ref.npz")
    for tag in ("oriented", "aligned", "kat"):
        got = ut.boxes_bev_iou_cpu(g["a_" + tag], g["b_" + tag])  # numpy in -> numpy out
        assert isinstance(got, np.ndarray)
        assert np.array_equal(got.view(np.uint32), g["iou_bev_" + tag].view(np.uint32)), tag


def test_host_side_shape_gates_of_the_round4_entry_points():
    """The sizing / gating helpers of the pre-gather first layer and of the paired small backward
    are host functions (no device needed): their answers at the network's shapes and just outside."""
    pkg = load_pkg()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(pkg.__file__), "lib3dioumatch_hip.so"))
    ok = lib.mlp_pregather_supported
    assert ok(8, 128, 2048, 1024, 32) == 1      # SA2
    assert ok(8, 128, 1024, 512, 16) == 1       # SA3
    assert ok(8, 128, 1024, 256, 16) == 1       # vote aggregation
    assert ok(8, 128, 4097, 1024, 16) == 0      # source row beyond the LDS stage
    assert ok(8, 128, 2048, 2048, 32) == 0      # m * ns beyond the inverse index
    assert ok(8, 128, 2048, 1024, 3) == 0       # nsample not a power of two
    assert ok(8, 130, 2048, 1024, 32) == 0      # channels not a multiple of 4
    assert ok(0, 128, 2048, 1024, 32) == 0
    small = lib.mlp_gemm_backward_small_supported
    assert small(8, 256, 256, 1024, 2, 1) == 1  # a vote-head layer, gradient operand on the fly
    assert small(8, 128, 259, 768, 0, 0) == 1   # SA4's pre-gather layer, plain operands
    assert small(8, 256, 256, 4096, 2, 1) == 0  # 32768 columns: not the small regime
    assert small(8, 256, 256, 1024, 3, 1) == 0  # pooled gradient operand: other kernels
    assert small(8, 256, 256, 16, 2, 1) == 0    # fewer columns than one chunk
    ws = lib.mlp_gemm_wgrad_workspace_floats
    ws.restype = ctypes.c_size_t
    for b, m, k, r in ((8, 256, 256, 1024), (8, 128, 259, 768), (2, 79, 128, 256), (8, 128, 131, 3072)):
        need = ws(b, m, k, r)
        assert need >= b * m * k and need % (m * k) == 0  # whole partial blocks, one per cloud at least


def test_weight_reduction_queue_host_logic():
    """The queue behind mlp_defer_weight_reductions is host state: with nothing queued, switching it
    on and off launches nothing and returns 0 (no device needed), and the Python context restores
    the immediate mode whatever happens inside it."""
    pkg = load_pkg()
    lib = ctypes.CDLL(os.path.join(os.path.dirname(pkg.__file__), "lib3dioumatch_hip.so"))
    assert lib.mlp_flush_weight_reductions() == 0
    assert lib.mlp_defer_weight_reductions(1) == 0
    assert lib.mlp_flush_weight_reductions() == 0
    assert lib.mlp_defer_weight_reductions(0) == 0
    K = importlib.import_module("pointnet2._mlp_ext")
    with K.deferred_weight_reductions():
        assert K._queued_workspaces == []
        with K.deferred_weight_reductions():       # nested: the outer context owns the queue
            assert K._queued_workspaces == []
    assert K._queued_workspaces is None
    with pytest.raises(RuntimeError):
        with K.deferred_weight_reductions():
            raise RuntimeError("inside")
    assert K._queued_workspaces is None            # back to immediate launches
    with K.deferred_weight_reductions(False):      # disabled: nothing is queued
        assert K._queued_workspaces is None


def test_pregather_path_steps_aside_on_the_cpu():
    """SharedMLP.pregather_ok is False for CPU tensors (the module then groups and convolves as the
    reference does), without touching the HIP library."""
    import torch
    load_pkg()
    pt = importlib.import_module("pointnet2.pytorch_utils")
    mlp = pt.SharedMLP([3 + 8, 16, 16], bn=True)
    xyz, new_xyz, feats = torch.rand(2, 64, 3), torch.rand(2, 16, 3), torch.rand(2, 8, 64)
    assert mlp.pregather_ok(xyz, new_xyz, feats, 16, 8) is False
    assert mlp.pregather_ok(xyz, new_xyz, None, 16, 8) is False
