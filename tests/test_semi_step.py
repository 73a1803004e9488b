"""The stage-2 (semi-supervised) step of votenet/step.py: EMA teacher forward + labeled loss +
pseudo-label consistency loss + backward + Adam + EMA update (reference train.py:305-371).
CPU: host logic on the oracle stand-ins.  GPU: HIP kernels, graph replay == eager."""
import importlib
import sys

import numpy as np
import pytest
import torch

from conftest import load_pkg
from test_unlabeled_loss import _oracle_nms

LAB, UNL, N, K = 2, 3, 3000, 64


def _setup(use_gpu, oracle, monkeypatch):
    load_pkg()
    utils = importlib.import_module("pointnet2.pointnet2_utils")
    V = importlib.import_module("3dioumatch_amd.votenet")
    losses = importlib.import_module("3dioumatch_amd.votenet.losses")
    U = importlib.import_module("3dioumatch_amd.votenet.losses_unlabeled")
    if use_gpu:
        return V, torch.device("cuda:0")
    from oracle import standin as oracle_ext
    monkeypatch.setattr(utils, "_ext", oracle_ext.make(oracle))
    monkeypatch.setattr(losses, "boxes_iou3d_gpu", lambda a, b: torch.from_numpy(
        oracle.boxes_iou3d(a.detach().numpy(), b.detach().numpy())))
    monkeypatch.setattr(U, "_lhs_nms", _oracle_nms(oracle))
    return V, torch.device("cpu")


def _loose_filter(V, cfg):
    """Random-weight networks pass no 0.9 thresholds: loosen them so pseudo labels exist."""
    U = importlib.import_module("3dioumatch_amd.votenet.losses_unlabeled")
    d = U.default_config_dict(cfg, unlabeled_batch_size=UNL)
    d.update(obj_threshold=0.3, cls_threshold=0.03, iou_threshold=0.2)
    return d


@pytest.mark.parametrize("use_gpu", [pytest.param(False, id="cpu-hostlogic"),
                                     pytest.param(True, id="gpu-hip", marks=pytest.mark.gpu)])
def test_semi_supervised_step(use_gpu, oracle_omp, monkeypatch):
    V, dev = _setup(use_gpu, oracle_omp, monkeypatch)
    cfg = V.scannet_config()
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    batch = {k: v.to(dev) for k, v in V.make_semi_batch(LAB, UNL, N, cfg, seed=3, num_objects=5).items()}
    runner = V.SemiSupervisedStep(cfg, dev, num_proposal=K, seed=4, graphs=False,
                                  config_dict=_loose_filter(V, cfg))
    student0 = step_mod.flat_params(runner.net).clone()
    assert torch.equal(step_mod.flat_params(runner.teacher), student0)
    bn_t0 = runner.teacher.pnet.bn1.running_mean.clone()
    torch.manual_seed(1)
    loss, ep = runner(dict(batch))
    assert bool(torch.isfinite(loss))
    # loss composition (train.py:348) and the pieces that feed it
    want = ep["detection_loss"] + 2.0 * ep["unlabeled_detection_loss"]
    assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
    assert ep["unlabeled_box_label_mask"].shape == (UNL, 64) and int(ep["unlabeled_box_label_mask"].sum()) > 0
    assert ep["objectness_label"].shape[0] == LAB and ep["unlabeled_objectness_label"].shape[0] == UNL
    # Adam moved the student; the teacher is the EMA with a = min(1 - 1/2, 0.999) = 0.5
    student1 = step_mod.flat_params(runner.net)
    assert float((student1 - student0).abs().max()) > 0
    teacher1 = step_mod.flat_params(runner.teacher)
    assert torch.allclose(teacher1, 0.5 * student0 + 0.5 * student1, rtol=0, atol=2e-6)
    assert all(p.grad is None for p in runner.teacher.parameters())
    # the teacher ran in train mode: its BatchNorm statistics moved, independently of the student's
    assert not torch.equal(runner.teacher.pnet.bn1.running_mean, bn_t0)
    assert not torch.equal(runner.teacher.pnet.bn1.running_mean, runner.net.pnet.bn1.running_mean)
    # second step: a = min(1 - 1/3, 0.999)
    student1 = student1.clone(); teacher1 = teacher1.clone()
    runner(dict(batch))
    student2 = step_mod.flat_params(runner.net)
    a = 1 - 1 / 3
    assert torch.allclose(step_mod.flat_params(runner.teacher), a * teacher1 + (1 - a) * student2,
                          rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_semi_graph_replay_matches_eager(oracle_omp, monkeypatch):
    """Captured stage-2 step (two index chains prefetched, both networks, all losses, device NMS,
    Adam + EMA) == the eager step: same pseudo labels and loss, same student and teacher weights."""
    V, dev = _setup(True, oracle_omp, monkeypatch)
    cfg = V.scannet_config()
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    batches = [{k: v.to(dev) for k, v in V.make_semi_batch(LAB, UNL, N, cfg, seed=s, num_objects=5).items()}
               for s in (5, 6)]
    results = []
    for graphs in (False, True):
        runner = V.SemiSupervisedStep(cfg, dev, num_proposal=K, seed=4, graphs=graphs,
                                      config_dict=_loose_filter(V, cfg))
        # (the zero-gradient parameters stay put in both arms: profiles/r4_step_repeatability.txt)
        step_mod.freeze_shift_invariant_parameters(runner.net)
        torch.manual_seed(9)
        torch.cuda.manual_seed_all(9)
        first, second = dict(batches[0]), dict(batches[1])
        runner.prefetch_geometry(first)
        runner.prefetch_geometry(second)
        loss0, ep0 = runner(first)
        loss0, mask0 = float(loss0), ep0["unlabeled_box_label_mask"].cpu().clone()
        torch.cuda.synchronize()
        loss1, ep1 = runner(second)
        assert runner.graphs == graphs
        results.append((loss0, float(loss1), mask0, step_mod.flat_params(runner.net).cpu(),
                        step_mod.flat_params(runner.teacher).cpu(),
                        runner.teacher.pnet.bn1.running_mean.cpu().clone()))
    eager, graph = results
    assert torch.equal(eager[2], graph[2])
    assert abs(eager[0] - graph[0]) <= 1e-5 * max(1.0, abs(eager[0]))
    assert abs(eager[1] - graph[1]) <= 4e-2 * max(1.0, abs(eager[1]))
    assert float((eager[3] - graph[3]).abs().max()) <= 1.2e-2  # two Adam steps of lr 2e-3
    print("semi params after 2 steps, graph vs eager: student %.2e teacher %.2e" % (
        float((eager[3] - graph[3]).norm() / eager[3].norm()),
        float((eager[4] - graph[4]).norm() / eager[4].norm())))
    assert float((eager[3] - graph[3]).norm() / eager[3].norm()) < 1.2e-2
    assert float((eager[4] - graph[4]).norm() / eager[4].norm()) < 1.2e-2
    # teacher BN running mean: after step 0 the teacher IS the student (EMA weight 0), whose
    # pre-BatchNorm biases carry the +-lr sign noise above; a bias moves the batch mean one to
    # one and the running mean by momentum (0.1) of that: 0.1 * 2 * lr = 4e-4
    assert torch.allclose(eager[5], graph[5], rtol=3e-3, atol=5e-4)


@pytest.mark.gpu
def test_one_loss_node_for_both_losses(oracle_omp, monkeypatch):
    """fused_loss._FusedSemiLoss (the supervised loss on the labeled scenes and the consistency loss
    on the unlabeled ones writing ONE gradient buffer per head output, the consistency rows
    pre-scaled by the loss weight) == the two separate autograd nodes on slices of the head outputs:
    same loss, same logged terms and labels, same parameter gradients (to the order of the fp32
    atomics), from the same initial state and noise."""
    V, dev = _setup(True, oracle_omp, monkeypatch)
    cfg = V.scannet_config()
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    batch = {k: v.to(dev) for k, v in V.make_semi_batch(LAB, UNL, N, cfg, seed=3, num_objects=5).items()}
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("VOTENET_FUSED_SEMI_LOSS", flag)
        runner = V.SemiSupervisedStep(cfg, dev, num_proposal=K, seed=4, graphs=False,
                                      config_dict=_loose_filter(V, cfg))
        torch.manual_seed(1)
        torch.cuda.manual_seed_all(1)
        view = dict(batch)
        view["labeled_num"] = LAB
        loss, ep = runner._forward_backward(view)
        res.append((float(loss), ep, runner.flat_grad.clone()))
    (l0, e0, g0), (l1, e1, g1) = res
    assert int(e1["unlabeled_box_label_mask"].sum()) > 0
    assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0))
    for key in ("detection_loss", "unlabeled_detection_loss", "vote_loss", "objectness_loss", "box_loss",
                "iou_loss", "unlabeled_box_loss", "unlabeled_center_loss", "unlabeled_sem_cls_loss",
                "unlabeled_objectness_loss", "pos_ratio", "unlabeled_pos_ratio"):
        a, b = float(e0[key]), float(e1[key])
        assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (key, a, b)
    for key in ("objectness_label", "object_assignment", "unlabeled_objectness_label",
                "unlabeled_object_assignment"):
        assert torch.equal(e0[key], e1[key]), key
    assert float((g0 - g1).norm() / g0.norm()) < 1e-5


@pytest.mark.gpu
def test_teacher_graph_on_any_stream(oracle_omp, monkeypatch):
    """The kernel library keeps no state per stream (include/mlp_hip.h `tickets`: every module
    owns the counters of its one-launch BatchNorm reductions), so the teacher's graph may be
    captured and replayed on ANY stream -- also the one torch.cuda.graph captures the student's
    graphs on, which the seventh graph runner of a process used to draw (torch.cuda.Stream() deals
    a pool of 32 handles round-robin) and, in round 5, corrupted both passes' statistics with.
    Two captured runners, one with its teacher on the default capture stream, step on the same
    batches: same loss, same student and teacher weights after two steps as the eager runner."""
    V, dev = _setup(True, oracle_omp, monkeypatch)
    cfg = V.scannet_config()
    batches = [{k: v.to(dev) for k, v in V.make_semi_batch(LAB, UNL, N, cfg, seed=5 + i, num_objects=5).items()}
               for i in range(2)]

    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")

    def run(graphs, teacher_stream=None):
        runner = V.SemiSupervisedStep(cfg, dev, num_proposal=K, seed=4, graphs=graphs,
                                      config_dict=_loose_filter(V, cfg))
        step_mod.freeze_shift_invariant_parameters(runner.net)
        if teacher_stream is not None:
            runner._teacher_stream = teacher_stream
            monkeypatch.setenv("STEP_SEMI_TEACHER_PROBE", "0")  # replay where it was captured
        torch.manual_seed(11)
        torch.cuda.manual_seed_all(11)
        loss0, _ = runner(dict(batches[0]))
        loss0 = float(loss0)
        runner(dict(batches[1]))
        if graphs:
            assert runner.graphs and runner._gt is not None
        return loss0, step_mod.flat_params(runner.net).clone(), step_mod.flat_params(runner.teacher).clone(), runner

    l_e, s_e, t_e, _ = run(False)
    default_capture = torch.cuda.graph.default_capture_stream
    for stream in (None, default_capture):
        l_g, s_g, t_g, runner = run(True, stream)
        if stream is not None:
            assert runner._teacher_stream.cuda_stream == default_capture.cuda_stream
        # (teacher and student are two module trees: two sets of counters, zero between launches)
        t_a = runner.teacher.backbone_net.sa1.mlp_module._pn2_tickets
        t_b = runner.net.backbone_net.sa1.mlp_module._pn2_tickets
        assert t_a.data_ptr() != t_b.data_ptr() and int(t_a.abs().sum()) == 0 and int(t_b.abs().sum()) == 0
        assert abs(l_e - l_g) <= 1e-5 * max(1.0, abs(l_e)), (l_e, l_g)
        # (two Adam steps of lr 2e-3, as test_semi_graph_replay_matches_eager; corrupted statistics
        #  put the gradient orders of magnitude apart)
        assert float((s_e - s_g).norm() / s_e.norm()) < 1.2e-2
        assert float((t_e - t_g).norm() / t_e.norm()) < 1.2e-2


@pytest.mark.parametrize("use_gpu", [pytest.param(False, id="cpu-hostlogic"),
                                     pytest.param(True, id="gpu-hip", marks=pytest.mark.gpu)])
def test_jitter_noise_from_the_caller(use_gpu, oracle_omp, monkeypatch):
    """forward_with_pred_jitter with inputs['jitter_noise'] = the two standard-normal tensors the
    pass would have drawn itself (votenet_iou_branch.py:161-162, in that order) is the same pass:
    what lets the semi-supervised runner draw the noise of its two forward graphs ahead of both."""
    V, dev = _setup(use_gpu, oracle_omp, monkeypatch)
    cfg = V.scannet_config()
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    net = step_mod.build_detector(cfg, num_proposal=K, seed=4).to(dev).train()
    pc = data.make_batch(2, N, cfg, seed=3, num_objects=5)["point_clouds"].to(dev)

    def seed():
        torch.manual_seed(11)
        if use_gpu:
            torch.cuda.manual_seed_all(11)

    seed()
    with torch.no_grad():
        drawn = net({"point_clouds": pc}, mode="jitter")
    seed()
    noise = (torch.randn((2, K, 3), device=dev), torch.randn((2, K, 3), device=dev))
    with torch.no_grad():
        given = net({"point_clouds": pc, "jitter_noise": noise}, mode="jitter")
    for key in ("jitter_center", "jitter_size", "iou_scores", "iou_scores_jitter"):
        # (two passes of the GPU forward differ by the order of their fp32 atomics)
        assert torch.allclose(drawn[key], given[key], rtol=1e-3, atol=1e-4), key
    # (other noise is another pass: the check above is not vacuous)
    other = (noise[1], noise[0])
    with torch.no_grad():
        swapped = net({"point_clouds": pc, "jitter_noise": other}, mode="jitter")
    assert not torch.allclose(drawn["jitter_center"], swapped["jitter_center"], rtol=1e-3, atol=1e-4)
    if use_gpu:
        with pytest.raises(ValueError):
            net({"point_clouds": pc, "jitter_noise": (noise[0][:1], noise[1])}, mode="jitter")
