"""Evaluation-path prediction parsing (votenet/eval_helper.py) against the REFERENCE's
models/ap_helper.py:parse_predictions (vectors from tests/golden/make_eval_golden.py): kept-box
mask identical, the per-class / per-box lists in the same order with the same corners and
confidences.  CPU: the NMS kernel is replaced by the oracle; GPU: everything on the device."""
import importlib

import numpy as np
import pytest
import torch

from conftest import golden, load_pkg


def _oracle_nms(oracle):
    def fn(center, size, heading, score, cls, thresh, old_type, same_class):
        out = []
        for i in range(center.shape[0]):
            aabb = oracle.camera_aabb(center[i].cpu().numpy(), size[i].cpu().numpy(),
                                      heading[i].cpu().numpy())
            out.append(oracle.nms3d_aabb(aabb, score[i].cpu().numpy(), cls[i].cpu().numpy(), thresh,
                                         old_type, same_class))
        return torch.from_numpy(np.stack(out)).bool().to(center.device)
    return fn


@pytest.mark.parametrize("tag", ["scannet", "sunrgbd", "nocls"])
@pytest.mark.parametrize("use_gpu", [pytest.param(False, id="cpu-hostlogic"),
                                     pytest.param(True, id="gpu-hip", marks=pytest.mark.gpu)])
def test_parse_predictions_matches_reference(use_gpu, tag, oracle, monkeypatch):
    load_pkg()
    V = importlib.import_module("3dioumatch_amd.votenet")
    E = importlib.import_module("3dioumatch_amd.votenet.eval_helper")
    g = golden("eval_parse_ref.npz")
    dev = torch.device("cuda:0" if use_gpu else "cpu")
    if not use_gpu:
        monkeypatch.setattr(E, "_nms3d", _oracle_nms(oracle))
    cfg = V.sunrgbd_config() if tag == "sunrgbd" else V.scannet_config()
    cls_nms, use_iou = (bool(v) for v in g[tag + "_flags"])
    ep = {k.split("::", 1)[1]: torch.from_numpy(g[k]).to(dev) for k in g.files
          if k.startswith(tag + "_in::")}
    config_dict = {"dataset_config": cfg, "remove_empty_box": False, "use_3d_nms": True,
                   "nms_iou": 0.25, "use_old_type_nms": False, "cls_nms": cls_nms,
                   "use_iou_for_nms": use_iou, "per_class_proposal": cls_nms, "conf_thresh": 0.05}
    batch = E.parse_predictions(ep, config_dict)
    assert isinstance(ep["pred_mask"], np.ndarray)  # the reference's layout: (B,K) numpy 0/1
    np.testing.assert_array_equal(ep["pred_mask"].astype(np.int32), g[tag + "_pred_mask"])
    size64, heading64 = E.decode_boxes(ep, cfg)
    corners = E.corners_upright_camera(ep["center"], size64, heading64).cpu().numpy()
    np.testing.assert_allclose(corners, g[tag + "_corners"], rtol=0, atol=2e-6)
    for i, cur in enumerate(batch):
        want_cls, want_j, want_conf = (g["%s_%s_%d" % (tag, n, i)] for n in ("cls", "j", "conf"))
        assert len(cur) == len(want_cls)
        assert [c for c, _, _ in cur] == want_cls.tolist()
        for (c, box, conf), j, wc in zip(cur, want_j, want_conf):
            assert np.allclose(box, g[tag + "_corners"][i, j], rtol=0, atol=2e-6)
            assert abs(conf - wc) <= 2e-6
