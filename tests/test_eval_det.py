"""AP evaluation (votenet/eval_det.py, eval_helper.APCalculator / parse_groundtruths) against the
REFERENCE's utils/box_util.py:box3d_iou, utils/eval_det.py and models/ap_helper.py (vectors from
tests/golden/make_evaldet_golden.py).

CPU: the oracle restatement is pinned to the reference's IoUs, and the host logic (grouping,
score-ordered marking, VOC AP) is checked with the oracle standing in for the kernel.
GPU: the kernel against the oracle (float64, same operation order) and the whole evaluation on the
device against the reference's numbers.
"""
import importlib

import numpy as np
import pytest
import torch

from conftest import golden, load_pkg

IOU_RTOL, IOU_ATOL = 1e-9, 1e-12   # float64 IoU: qhull area (reference) vs shoelace area


def _mods():
    load_pkg()
    return (importlib.import_module("3dioumatch_amd.votenet"),
            importlib.import_module("3dioumatch_amd.votenet.eval_det"),
            importlib.import_module("3dioumatch_amd.votenet.eval_helper"))


def _detection_set(g):
    preds, gts = [], []
    for i in range(int(g["num_scans"])):
        preds.append([(int(c), b, s) for c, b, s in
                      zip(g["det_%d_cls" % i], g["det_%d_box" % i], g["det_%d_score" % i])])
        gts.append([(int(c), b) for c, b in zip(g["gt_%d_cls" % i], g["gt_%d_box" % i])])
    return preds, gts


def _oracle_best_match(oracle):
    def fn(det, gt_begin, gt_count, gt, device):
        return oracle.best_match(det, gt_begin, gt_count, gt)
    return fn


def test_oracle_box3d_iou_matches_reference(oracle):
    g = golden("evaldet_ref.npz")
    want = g["pair_iou"]
    got = oracle.box3d_iou_matrix(g["pair_a"], g["pair_b"])
    ok = ~np.isnan(want)                       # the reference raised (qhull) on the others
    assert ok.sum() >= want.size - 4
    np.testing.assert_allclose(got[ok], want[ok], rtol=IOU_RTOL, atol=IOU_ATOL)
    assert (want[ok] > 0.05).sum() > 100       # the set really overlaps
    # identical boxes: every vertex lies ON a clipping edge and the strict inside() decides.  The
    # axis-aligned copy gives 1.0; for the rotated copy a parallel-edge intersection divides by
    # zero, the reference hands NaN vertices to qhull and raises -- the restatement returns NaN.
    assert got[40, 36] == want[40, 36] == 1.0
    assert np.isnan(want[41, 37]) and np.isnan(got[41, 37])


@pytest.mark.parametrize("use_gpu", [pytest.param(False, id="cpu-hostlogic"),
                                     pytest.param(True, id="gpu-hip", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("thr", [0.25, 0.5])
def test_eval_det_matches_reference(use_gpu, thr, oracle, monkeypatch):
    _, D, E = _mods()
    g = golden("evaldet_ref.npz")
    if not use_gpu:
        monkeypatch.setattr(D, "_best_match", _oracle_best_match(oracle))
    preds, gts = _detection_set(g)
    calc = E.APCalculator(thr, None, device="cuda:0" if use_gpu else None)
    calc.step(preds, gts)
    rec, prec, ap = D.eval_det(calc.pred_map_cls, calc.gt_map_cls, ovthresh=thr,
                               device="cuda:0" if use_gpu else None)
    classes = sorted(int(k.split("_")[-1]) for k in g.files if k.startswith("ap_%g_" % thr))
    assert sorted(ap.keys()) == classes
    for c in classes:
        np.testing.assert_allclose(rec[c], g["rec_%g_%d" % (thr, c)], rtol=0, atol=1e-12)
        np.testing.assert_allclose(prec[c], g["prec_%g_%d" % (thr, c)], rtol=0, atol=1e-12)
        assert ap[c] == pytest.approx(float(g["ap_%g_%d" % (thr, c)]), abs=1e-12)
    metrics = calc.compute_metrics()
    keys = sorted(metrics.keys())
    assert keys == [str(k) for k in g["metrics_%g_keys" % thr]]
    np.testing.assert_allclose([metrics[k] for k in keys], g["metrics_%g_vals" % thr], rtol=0, atol=1e-12)


def test_eval_det_class_without_predictions_and_empty_scans(oracle, monkeypatch):
    _, D, _ = _mods()
    monkeypatch.setattr(D, "_best_match", _oracle_best_match(oracle))
    g = golden("evaldet_ref.npz")
    a = g["pair_a"]
    pred_all = {0: [(0, a[0], 0.9), (0, a[1], 0.8)], 1: [], 2: [(0, a[2], 0.7)]}
    gt_all = {0: [(0, a[0]), (3, a[5])], 1: [(0, a[7])], 2: []}
    rec, prec, ap = D.eval_det(pred_all, gt_all, ovthresh=0.25)
    assert ap[3] == 0 and rec[3] == 0 and prec[3] == 0          # GT class never predicted
    np.testing.assert_allclose(rec[0], [0.5, 0.5, 0.5])           # a[0] found, a[7] missed
    np.testing.assert_allclose(prec[0], [1.0, 0.5, 1.0 / 3])
    # single-class entry point, a scan with detections but no ground truth of the class
    r, p, v = D.eval_det_cls({0: [(a[0], 0.9)], 5: [(a[1], 0.95)]}, {0: [a[0]]}, ovthresh=0.25)
    np.testing.assert_allclose(r, [0.0, 1.0])
    np.testing.assert_allclose(p, [0.0, 0.5])
    assert v == pytest.approx(0.5)


def test_eval_det_duplicate_detections_are_false_positives(oracle, monkeypatch):
    _, D, _ = _mods()
    monkeypatch.setattr(D, "_best_match", _oracle_best_match(oracle))
    a = golden("evaldet_ref.npz")["pair_a"]
    shifted = a[3] + np.float32(0.01)
    r, p, v = D.eval_det_cls({0: [(shifted, 0.6), (a[3] + np.float32(0.02), 0.9)]}, {0: [a[3]]}, ovthresh=0.5)
    np.testing.assert_allclose(r, [1.0, 1.0])                     # best-scored claims the box,
    np.testing.assert_allclose(p, [1.0, 0.5])                     # the second match is a FP


def test_voc_ap_07_metric():
    _, D, _ = _mods()
    rec = np.array([0.1, 0.2, 0.2, 0.5, 1.0])
    prec = np.array([1.0, 1.0, 0.66, 0.75, 0.5])
    want = sum((prec[rec >= t].max() if (rec >= t).any() else 0) / 11. for t in np.arange(0., 1.1, 0.1))
    assert D.voc_ap(rec, prec, use_07_metric=True) == pytest.approx(want)
    assert D.voc_ap(rec, prec) == pytest.approx(0.2 * 1.0 + 0.3 * 0.75 + 0.5 * 0.5)


def test_eval_det_has_no_cpu_path():
    _, D, _ = _mods()
    a = golden("evaldet_ref.npz")["pair_a"]
    with pytest.raises(RuntimeError):
        D.eval_det_cls({0: [(a[0], 0.9)]}, {0: [a[0]]}, device="cpu")
    with pytest.raises(RuntimeError):
        D.corners_iou3d_gpu(torch.from_numpy(a), torch.from_numpy(a))


@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
@pytest.mark.parametrize("use_gpu", [pytest.param(False, id="cpu-hostlogic"),
                                     pytest.param(True, id="gpu", marks=pytest.mark.gpu)])
def test_parse_groundtruths_matches_reference(use_gpu, tag):
    V, _, E = _mods()
    g = golden("evaldet_ref.npz")
    dev = torch.device("cuda:0" if use_gpu else "cpu")
    cfg = V.sunrgbd_config() if tag == "sunrgbd" else V.scannet_config()
    ep = {k.split("::", 1)[1]: torch.from_numpy(g[k]).to(dev) for k in g.files
          if k.startswith(tag + "_lab::")}
    batch = E.parse_groundtruths(ep, {"dataset_config": cfg})
    assert ep["batch_gt_map_cls"] is batch
    for i, cur in enumerate(batch):
        assert [c for c, _ in cur] == g["%s_gtcls_%d" % (tag, i)].tolist()
        got = np.stack([b for _, b in cur])
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, g["%s_gtbox_%d" % (tag, i)], rtol=0, atol=2e-6)


# ---------------------------------------------------------------------------------------------
# GPU: kernel vs oracle
# ---------------------------------------------------------------------------------------------
def _random_corners(rng, n, spread):
    """Boxes in the vertex order of get_3d_box (utils/box_util.py:335-358), float32."""
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1.])
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1.])
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1.])
    ctr = rng.uniform(-spread, spread, (n, 3)) * [1, 0.25, 1]
    l, h, w = (rng.uniform(0.3, 1.6, (n, 1)) for _ in range(3))
    ang = rng.uniform(-np.pi, np.pi, (n, 1))
    x, y, z = sx * l / 2, sy * h / 2, sz * w / 2
    c, s = np.cos(ang), np.sin(ang)
    out = np.stack([c * x + s * z + ctr[:, 0:1], y + ctr[:, 1:2], -s * x + c * z + ctr[:, 2:3]], -1)
    return out.astype(np.float32)


@pytest.mark.gpu
def test_gpu_corners_iou_matrix_matches_oracle_and_reference(oracle):
    _, D, _ = _mods()
    g = golden("evaldet_ref.npz")
    a, b = g["pair_a"], g["pair_b"]
    got = D.corners_iou3d_gpu(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    want = oracle.box3d_iou_matrix(a, b)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13, equal_nan=True)
    ok = ~np.isnan(g["pair_iou"])
    np.testing.assert_allclose(got[ok], g["pair_iou"][ok], rtol=IOU_RTOL, atol=IOU_ATOL)
    rng = np.random.default_rng(5)
    a, b = _random_corners(rng, 700, 2.0), _random_corners(rng, 333, 2.0)
    got = D.corners_iou3d_gpu(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    want = oracle.box3d_iou_matrix(a, b)
    assert (want > 0.05).sum() > 5000
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13, equal_nan=True)
    empty = D.corners_iou3d_gpu(torch.from_numpy(a[:0]).cuda(), torch.from_numpy(b).cuda())
    assert empty.shape == (0, 333)


@pytest.mark.gpu
def test_gpu_best_match_matches_oracle(oracle):
    _, D, _ = _mods()
    rng = np.random.default_rng(6)
    nd, ng = 20000, 3000
    det, gt = _random_corners(rng, nd, 1.5), _random_corners(rng, ng, 1.5)
    count = rng.integers(0, 12, nd).astype(np.int32)
    count[::17] = 0
    begin = rng.integers(0, ng - 12, nd).astype(np.int32)
    ov, jm = D.corners_best_match_gpu(torch.from_numpy(det).cuda(), torch.from_numpy(begin).cuda(),
                                      torch.from_numpy(count).cuda(), torch.from_numpy(gt).cuda())
    wov, wjm = oracle.best_match(det, begin, count, gt)
    np.testing.assert_array_equal(jm.cpu().numpy(), wjm)
    np.testing.assert_allclose(ov.cpu().numpy(), wov, rtol=0, atol=1e-13)
    assert np.isneginf(wov[::17]).all() and (wjm[::17] == -1).all()
    assert (wov > 0.25).sum() > 1000


def _perfect_end_points(V, cfg, rng, scenes=3, k=48):
    """Head outputs that decode to the ground-truth boxes of a synthetic batch (plus far-away
    low-objectness clutter), so that the whole evaluation chain must report AP = 1."""
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    batch = data.make_batch(scenes, 256, cfg, seed=int(rng.integers(1 << 30)), num_objects=6)
    nh, ns, nc = cfg.num_heading_bin, cfg.num_size_cluster, cfg.num_class
    mean = torch.from_numpy(cfg.mean_size_arr.astype(np.float32))
    center = torch.full((scenes, k, 3), 40.0) + torch.arange(k).view(1, k, 1) * 3.0  # clutter, spread out
    obj = torch.tensor([4.0, -4.0]).repeat(scenes, k, 1)                             # "no object"
    h_scores = torch.zeros(scenes, k, nh)
    h_res = torch.zeros(scenes, k, nh)
    s_scores = torch.zeros(scenes, k, ns)
    s_res = torch.zeros(scenes, k, ns, 3)
    sem = torch.zeros(scenes, k, nc)
    for b in range(scenes):
        n = int(batch["box_label_mask"][b].sum())
        for j in range(n):
            # 2 mm off: an EXACT copy of a rotated box makes polygon_clip divide by zero (parallel
            # coincident edges; the reference raises there, see the NaN case above)
            center[b, j] = batch["center_label"][b, j] + 0.002
            obj[b, j] = torch.tensor([-6.0, 6.0])
            h_scores[b, j, batch["heading_class_label"][b, j]] = 8.0
            h_res[b, j, batch["heading_class_label"][b, j]] = batch["heading_residual_label"][b, j]
            s_scores[b, j, batch["size_class_label"][b, j]] = 8.0
            s_res[b, j, batch["size_class_label"][b, j]] = batch["size_residual_label"][b, j]
            sem[b, j, batch["sem_cls_label"][b, j]] = 12.0
    ep = {"center": center, "objectness_scores": obj, "heading_scores": h_scores,
          "heading_residuals": h_res, "size_scores": s_scores, "size_residuals": s_res,
          "sem_cls_scores": sem, "iou_scores": torch.zeros(scenes, k, nc)}
    ep.update(batch)
    return ep


@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
def test_evaluation_chain_perfect_predictions_cpu_hostlogic(tag, oracle, monkeypatch):
    """parse_predictions -> parse_groundtruths -> APCalculator on predictions equal to the ground
    truth: every class with a ground-truth box has AP = 1 and recall = 1 at IoU 0.25 and 0.5
    (the device kernels are replaced by the oracle here; the GPU variant is below)."""
    V, D, E = _mods()
    from test_eval_helper import _oracle_nms
    monkeypatch.setattr(E, "_nms3d", _oracle_nms(oracle))
    monkeypatch.setattr(D, "_best_match", _oracle_best_match(oracle))
    _run_perfect(V, E, tag, torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
def test_evaluation_chain_perfect_predictions_gpu(tag):
    V, D, E = _mods()
    _run_perfect(V, E, tag, torch.device("cuda:0"))


def _run_perfect(V, E, tag, dev):
    cfg = V.sunrgbd_config() if tag == "sunrgbd" else V.scannet_config()
    ep = {k: (v.to(dev) if torch.is_tensor(v) else v)
          for k, v in _perfect_end_points(V, cfg, np.random.default_rng(3)).items()}
    # (NMS threshold close to 1: synthetic ground-truth boxes of one class may overlap, and a
    # suppressed true positive is not what this test is about)
    config_dict = {"dataset_config": cfg, "remove_empty_box": False, "use_3d_nms": True,
                   "nms_iou": 0.999, "use_old_type_nms": False, "cls_nms": True,
                   "use_iou_for_nms": False, "per_class_proposal": True, "conf_thresh": 0.05}
    preds = E.parse_predictions(ep, config_dict)
    gts = E.parse_groundtruths(ep, config_dict)
    assert sum(len(g) for g in gts) == int(ep["box_label_mask"].sum())
    for thr in (0.25, 0.5):
        calc = E.APCalculator(thr, None, device=str(dev) if dev.type == "cuda" else None)
        calc.step(preds, gts)
        metrics = calc.compute_metrics()
        classes = sorted({c for g in gts for c, _ in g})
        for c in classes:
            assert metrics["%d Average Precision" % c] == pytest.approx(1.0, abs=1e-9), (thr, c)
            assert metrics["%d Recall" % c] == pytest.approx(1.0, abs=1e-9), (thr, c)
