"""The fused supervised loss (votenet/fused_loss.py, csrc/votenet_loss.hip, csrc/loss_core.h)
against the tensor-op formulation (votenet/losses.py, itself pinned to the reference's
get_labeled_loss by tests/golden/train_step_ref.npz).

CPU: csrc/loss_core.h -- the functions the kernels call -- is compiled with g++
(tests/loss_host.cpp) and driven through the same Python binding with host pointers, so every
loss term, statistic, label and gradient is checked against autograd without a GPU.
GPU: the kernels themselves against the tensor-op path on the device.
"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import golden, load_pkg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_layer_golden import seeded_state  # noqa: E402
from make_step_golden import B, K, N, STAT_KEYS  # noqa: E402
from test_train_step import _setup, _restore  # noqa: E402,F401  (autouse fixture restores the ops)

LABELS = ("objectness_label", "objectness_mask", "object_assignment")
LOGGED = STAT_KEYS + ["pos_ratio", "neg_ratio", "obj_acc", "cls_acc", "obj_count", "pred_iou_value",
                      "pred_iou_obj_value", "iou_acc", "iou_acc_obj", "jitter_iou_acc",
                      "jitter_iou_acc_obj", "box_loss", "detection_loss"]


@pytest.fixture(scope="module")
def host_build():
    so = os.path.join(HERE, "_loss_host.so")
    src = os.path.join(HERE, "loss_host.cpp")
    core = os.path.join(os.path.dirname(HERE), "3dioumatch_amd", "csrc", "loss_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, src])
    return ctypes.CDLL(so)


def _forward(V, cfg, dev, seed_net=21, seed_batch=33, scenes=B):
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    net = V.VoteNet(cfg.num_class, cfg.num_heading_bin, cfg.num_size_cluster, cfg.mean_size_arr,
                    cfg, input_feature_dim=1, num_proposal=K, sampling="seed_fps")
    seeded_state(net, seed=seed_net)
    net = net.to(dev).train()
    batch = {k: v.to(dev) for k, v in data.make_batch(scenes, N, cfg, seed=seed_batch, num_objects=6).items()}

    def run():
        net.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        noise = [torch.randn(scenes, K, 3), torch.randn(scenes, K, 3)]
        real = torch.randn
        torch.randn = lambda *a, **k: noise.pop(0).to(dev)
        try:
            ep = net(batch, mode="jitter")
        finally:
            torch.randn = real
        ep.update(batch)
        return ep
    return net, run


def _both_paths(V, cfg, dev, monkeypatch, extra=None, **kw):
    fused = importlib.import_module("3dioumatch_amd.votenet.fused_loss")
    net, run = _forward(V, cfg, dev, **kw)
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VOTENET_FUSED_LOSS", flag)
        ep = run()
        ep["all_supervised"] = True  # what SupervisedStep passes for a fully labeled batch
        if extra:
            ep.update(extra)
            if "labeled_num" in extra:
                del ep["all_supervised"]
        loss, ep = V.get_labeled_loss(ep, cfg, {"dataset_config": cfg})
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
        out.append((ep, grads))
    assert fused.enabled()
    return out


def _compare(ref, got, tol):
    (ep0, g0), (ep1, g1) = ref, got
    for key in LABELS + ("pred_bbox",):
        a, b = ep0[key].detach().cpu().numpy(), ep1[key].detach().cpu().numpy()
        if key == "pred_bbox":
            np.testing.assert_allclose(b, a, rtol=0, atol=1e-6)
        else:
            assert np.array_equal(a.astype(np.float64), b.astype(np.float64)), key
    for key in LOGGED + ["loss"]:
        if key in ep0:
            a, b = float(ep0[key].detach()), float(ep1[key].detach())
            assert abs(a - b) <= tol * max(1.0, abs(a)), (key, a, b)
    assert set(g0) == set(g1)
    # (biases in front of a BatchNorm have a mathematically zero gradient: rounding noise in both
    # formulations, hence the floor relative to the whole gradient)
    whole = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g0.values())))
    for name in g0:
        err = float((g0[name] - g1[name]).norm()) / max(0.01 * whole, float(g0[name].norm()))
        assert err <= 20 * tol, (name, err)


@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
def test_host_build_matches_tensor_ops(tag, oracle_omp, host_build, monkeypatch):
    V, dev = _setup(False, oracle_omp)
    fused = importlib.import_module("3dioumatch_amd.votenet.fused_loss")
    monkeypatch.setattr(fused, "_HOST_BUILD", host_build)
    cfg = V.scannet_config() if tag == "scannet" else V.sunrgbd_config()
    calls = []
    real_launch = fused._launch
    monkeypatch.setattr(fused, "_launch", lambda name, a, d: (calls.append(name), real_launch(name, a, d)))
    ref, got = _both_paths(V, cfg, dev, monkeypatch)
    assert calls == ["votenet_loss_decode", "votenet_loss_forward_backward"]  # the fused path ran
    _compare(ref, got, 2e-5)
    # and against the reference's own numbers
    g = golden("train_step_ref.npz")
    for key in STAT_KEYS:
        want = float(g["%s_%s" % (tag, key)])
        assert abs(float(got[0][key]) - want) <= 1e-3 * max(1.0, abs(want)), (key, float(got[0][key]), want)


def test_host_build_labeled_scenes_first(oracle_omp, host_build, monkeypatch):
    """Stage-2 layout: only the first `labeled_num` scenes are supervised; the others get no
    gradient from this loss."""
    V, dev = _setup(False, oracle_omp)
    fused = importlib.import_module("3dioumatch_amd.votenet.fused_loss")
    monkeypatch.setattr(fused, "_HOST_BUILD", host_build)
    cfg = V.scannet_config()
    ref, got = _both_paths(V, cfg, dev, monkeypatch, extra={"labeled_num": B - 1})
    assert got[0]["objectness_label"].shape[0] == B - 1
    _compare(ref, got, 2e-5)


def test_host_build_no_positive_proposals_and_empty_scene(oracle_omp, host_build, monkeypatch):
    """All GT slots empty in one scene, and no proposal within 0.3 m anywhere: the positive-count
    normaliser is 1e-6 and every positive-only term is exactly zero."""
    V, dev = _setup(False, oracle_omp)
    fused = importlib.import_module("3dioumatch_amd.votenet.fused_loss")
    monkeypatch.setattr(fused, "_HOST_BUILD", host_build)
    cfg = V.scannet_config()
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    batch = data.make_batch(B, N, cfg, seed=33, num_objects=6)
    far = batch["center_label"] + 50.0
    mask = batch["box_label_mask"].clone()
    mask[0] = 0
    ref, got = _both_paths(V, cfg, dev, monkeypatch, extra={"center_label": far, "box_label_mask": mask})
    assert float(got[0]["obj_count"]) == 0.0
    _compare(ref, got, 2e-5)


def test_fused_loss_has_no_cpu_path(oracle_omp, monkeypatch):
    V, dev = _setup(False, oracle_omp)
    fused = importlib.import_module("3dioumatch_amd.votenet.fused_loss")
    assert not fused.available(torch.device("cpu"))
    with pytest.raises(RuntimeError):
        fused._launch("votenet_loss_decode", fused.VnLossArgs(), torch.device("cpu"))


def test_args_struct_matches_header():
    """ctypes mirror of VnLossArgs: same field names, in the header's order."""
    import re
    load_pkg()
    fused = importlib.import_module("3dioumatch_amd.votenet.fused_loss")
    text = open(os.path.join(os.path.dirname(HERE), "include", "loss_hip.h")).read()
    body = text[text.index("typedef struct VnLossArgs {"):text.index("} VnLossArgs;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"[A-Za-z_][A-Za-z_0-9]*", part)[-1])
    assert names == [n for n, _ in fused.VnLossArgs._fields_]
    assert ctypes.sizeof(fused.VnLossTensor) == 40


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
def test_gpu_kernels_match_tensor_ops(tag, oracle_omp, monkeypatch):
    V, dev = _setup(True, oracle_omp)
    cfg = V.scannet_config() if tag == "scannet" else V.sunrgbd_config()
    ref, got = _both_paths(V, cfg, dev, monkeypatch)
    _compare(ref, got, 5e-5)


@pytest.mark.gpu
def test_gpu_kernels_labeled_scenes_first(oracle_omp, monkeypatch):
    V, dev = _setup(True, oracle_omp)
    cfg = V.scannet_config()
    ref, got = _both_paths(V, cfg, dev, monkeypatch, extra={"labeled_num": B - 1})
    _compare(ref, got, 5e-5)


def _consistency_both_paths(tag, dev, monkeypatch, nms=None):
    """get_unlabeled_loss on the reference's golden inputs, tensor operations (flag 0) and the
    fused kernels in consistency mode (flag 1): (loss, end_points, input gradients) each."""
    V = importlib.import_module("3dioumatch_amd.votenet")
    U = importlib.import_module("3dioumatch_amd.votenet.losses_unlabeled")
    if nms is not None:
        monkeypatch.setattr(U, "_lhs_nms", nms)
    g = golden("unlabeled_loss_ref.npz")
    cfg = V.scannet_config() if tag == "scannet" else V.sunrgbd_config()
    keys = ("center", "heading_residuals_normalized", "size_residuals_normalized", "sem_cls_scores",
            "heading_scores", "size_scores", "objectness_scores")
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VOTENET_FUSED_LOSS", flag)
        ep = {k.split("::", 1)[1]: torch.from_numpy(g[k]).to(dev) for k in g.files
              if k.startswith(tag + "_in_ep::")}
        ema = {k.split("::", 1)[1]: torch.from_numpy(g[k]).to(dev) for k in g.files
               if k.startswith(tag + "_in_ema::")}
        for k in keys:
            ep[k].requires_grad_(True)
        ep["labeled_num"] = int(ep["supervised_mask"].sum())
        loss, ep = U.get_unlabeled_loss(ep, ema, cfg, U.default_config_dict(cfg, dataset=tag,
                                                                            unlabeled_batch_size=3))
        loss.backward()
        out.append((loss.detach(), ep, {k: (ep[k].grad.detach().clone() if ep[k].grad is not None
                                            else torch.zeros_like(ep[k])) for k in keys}))
    return out


def _compare_consistency(ref, got, tol):
    (l0, e0, g0), (l1, e1, g1) = ref, got
    assert abs(float(l0) - float(l1)) <= tol * max(1.0, abs(float(l0)))
    for key in ("unlabeled_objectness_label", "unlabeled_objectness_mask", "unlabeled_object_assignment"):
        assert np.array_equal(e0[key].cpu().numpy().astype(np.float64), e1[key].cpu().numpy().astype(np.float64)), key
    for key in ("unlabeled_objectness_loss", "unlabeled_pos_ratio", "unlabeled_neg_ratio",
                "unlabeled_center_loss", "unlabeled_heading_cls_loss", "unlabeled_heading_reg_loss",
                "unlabeled_size_cls_loss", "unlabeled_size_reg_loss", "unlabeled_sem_cls_loss",
                "unlabeled_box_loss", "unlabeled_detection_loss"):
        a, b = float(e0[key].detach()), float(e1[key].detach())
        assert abs(a - b) <= tol * max(1.0, abs(a)), (key, a, b)
    assert np.array_equal(e0["unlabeled_center_label"].cpu().numpy(), e1["unlabeled_center_label"].cpu().numpy())
    whole = float(torch.sqrt(sum((v.double() ** 2).sum() for v in g0.values())))
    assert whole > 0
    for name in g0:
        err = float((g0[name] - g1[name]).norm()) / max(0.01 * whole, float(g0[name].norm()))
        assert err <= 20 * tol, (name, err)
    assert float(g1["objectness_scores"].abs().max()) == 0.0  # a statistic only: no gradient


@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
def test_host_build_consistency_loss_matches_tensor_ops(tag, oracle, host_build, monkeypatch):
    """The consistency loss on pseudo labels through the supervised loss' functions in their
    consistency mode (loss_core.h compiled for the host) == losses_unlabeled's tensor operations
    with autograd, on the inputs of the reference's golden vectors: labels, every logged term,
    the loss, and the gradient w.r.t. each head output."""
    from test_unlabeled_loss import _oracle_nms
    load_pkg()
    fused = importlib.import_module("3dioumatch_amd.votenet.fused_loss")
    monkeypatch.setattr(fused, "_HOST_BUILD", host_build)
    calls = []
    real_launch = fused._launch
    monkeypatch.setattr(fused, "_launch", lambda name, a, d: (calls.append(name), real_launch(name, a, d)))
    ref, got = _consistency_both_paths(tag, torch.device("cpu"), monkeypatch, _oracle_nms(oracle))
    assert calls == ["votenet_loss_forward_backward"]  # the fused path ran, once
    _compare_consistency(ref, got, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
def test_gpu_consistency_loss_matches_tensor_ops(tag, oracle_omp, monkeypatch):
    _setup(True, oracle_omp)
    ref, got = _consistency_both_paths(tag, torch.device("cuda:0"), monkeypatch)
    _compare_consistency(ref, got, 5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["scannet", "sunrgbd"])
def test_decode_scores_kernel_matches_tensor_ops(tag, monkeypatch):
    """votenet_decode_scores / _grad == decode_scores in tensor operations
    (models/proposal_module.py:24-54): the nine predictions bit for bit except softplus (1 ulp),
    and the gradient of a random functional of all of them w.r.t. the head output and the
    aggregated vote positions; some outputs unused (their gradient is None), softplus arguments
    beyond its threshold of 20."""
    load_pkg()
    V = importlib.import_module("3dioumatch_amd.votenet")
    heads = importlib.import_module("3dioumatch_amd.votenet.heads")
    dev = torch.device("cuda:0")
    cfg = V.scannet_config() if tag == "scannet" else V.sunrgbd_config()
    nh, ns, nc = cfg.num_heading_bin, cfg.num_size_cluster, cfg.num_class
    g = torch.Generator().manual_seed(nh + ns)
    b, k = 3, 70
    width = 5 + 2 * nh + 4 * ns + nc
    net0 = (torch.randn(b, width, k, generator=g) * 3).to(dev)
    net0[0, 5 + 2 * nh + ns, :5] = 25.0    # softplus' linear branch
    agg0 = torch.randn(b, k, 3, generator=g).to(dev)
    keys = ("objectness_scores", "center", "heading_scores", "heading_residuals_normalized",
            "heading_residuals", "size_scores", "size_residuals_normalized", "size_residuals", "sem_cls_scores")
    weights = {}
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VOTENET_FUSED_DECODE", flag)
        net, agg = net0.clone().requires_grad_(True), agg0.clone().requires_grad_(True)
        ep = heads.decode_scores(net, {"aggregated_vote_xyz": agg}, nc, nh, ns, cfg.mean_size(dev))
        total = 0.0
        for key in keys:
            if key in ("heading_scores", "size_residuals"):
                continue  # unused outputs: no gradient arrives for them
            if key not in weights:
                weights[key] = torch.randn(ep[key].shape, generator=g).to(dev)
            total = total + (ep[key] * weights[key]).sum()
        total.backward()
        res.append(({key: ep[key].detach() for key in keys}, net.grad.clone(), agg.grad.clone()))
    (o0, gn0, ga0), (o1, gn1, ga1) = res
    for key in keys:
        assert o1[key].is_contiguous() and o0[key].shape == o1[key].shape, key
        tol = 2e-6 if key.startswith("size_residuals") else 0.0
        assert float((o0[key] - o1[key]).abs().max()) <= tol * max(1.0, float(o0[key].abs().max())), key
    assert float((gn0 - gn1).abs().max()) <= 2e-6 * float(gn0.abs().max())
    assert torch.equal(ga0, ga1)
