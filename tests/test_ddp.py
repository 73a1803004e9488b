"""Data-parallel train step on the CPU: world_size 2, gloo, oracle stand-ins for the extensions
(so the N>1 path of 3dioumatch_amd/votenet/step.py is covered without GPUs).

Checks: (1) after backward every rank holds the SAME gradient and it equals the mean of the
per-rank gradients computed without DDP; (2) parameters stay bit-identical across ranks after
the Adam step; (3) BatchNorm buffers are NOT synchronised (per-replica BN, like the reference);
(4) the EMA teacher update matches train.py:285-289 and needs no communication.
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg

B, N, K = 1, 2600, 32


def _install_standins():
    load_pkg()
    from oracle import standin
    from oracle.oracle import Oracle
    o = Oracle(omp=False)
    utils = importlib.import_module("pointnet2.pointnet2_utils")
    losses = importlib.import_module("3dioumatch_amd.votenet.losses")
    utils._ext = standin.make(o)
    losses.boxes_iou3d_gpu = lambda a, b: torch.from_numpy(
        o.boxes_iou3d(a.detach().numpy(), b.detach().numpy()))
    unl = importlib.import_module("3dioumatch_amd.votenet.losses_unlabeled")
    from test_unlabeled_loss import _oracle_nms
    unl._lhs_nms = _oracle_nms(o)


def _batch(V, cfg, seed):
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    return data.make_batch(B, N, cfg, seed=seed, num_objects=5)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_standins()
    V = importlib.import_module("3dioumatch_amd.votenet")
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    cfg = V.scannet_config()
    runner = V.SupervisedStep(cfg, torch.device("cpu"), world_size=world, num_proposal=K)
    torch.manual_seed(100 + rank)  # jitter noise differs per rank, like the data
    loss, _ = runner(_batch(V, cfg, seed=200 + rank))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             grad=step_mod.flat_grads(runner.net).numpy(),
             params=step_mod.flat_params(runner.net).numpy(),
             bn_mean=runner.net.backbone_net.sa1.mlp_module.layer0.bn.bn.running_mean.numpy(),
             loss=float(loss))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_gloo(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # (1) identical, all-reduced gradients on both ranks
    assert np.array_equal(r0["grad"], r1["grad"])
    # (2) identical parameters after the optimizer step
    assert np.array_equal(r0["params"], r1["params"])
    # (3) per-replica BatchNorm statistics (different data -> different running means)
    assert not np.array_equal(r0["bn_mean"], r1["bn_mean"])
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"]) and r0["loss"] != r1["loss"]

    # the all-reduced gradient is the MEAN of the two single-process gradients
    _install_standins()
    V = importlib.import_module("3dioumatch_amd.votenet")
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    cfg = V.scannet_config()
    singles = []
    for rank in range(world):
        runner = V.SupervisedStep(cfg, torch.device("cpu"), world_size=1, num_proposal=K)
        torch.manual_seed(100 + rank)
        runner.optimizer.zero_grad(set_to_none=True)
        batch = _batch(V, cfg, seed=200 + rank)
        ep = runner.model(batch, mode="jitter")
        ep.update(batch)
        loss, _ = V.get_labeled_loss(ep, cfg, {"dataset_config": cfg})
        loss.backward()
        singles.append(step_mod.flat_grads(runner.net).numpy())
    want = (singles[0] + singles[1]) / 2
    err = np.linalg.norm(r0["grad"] - want) / np.linalg.norm(want)
    assert err < 1e-5, err


def test_ema_update_and_schedules():
    load_pkg()
    V = importlib.import_module("3dioumatch_amd.votenet")
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    cfg = V.scannet_config()
    student = step_mod.build_detector(cfg, num_proposal=16, seed=1)
    teacher = step_mod.build_detector(cfg, num_proposal=16, seed=2)
    ref = [p.detach().clone() for p in teacher.parameters()]
    for step, alpha in ((0, 0.999), (5, 0.999), (5000, 0.999)):
        a = min(1 - 1 / (step + 1), alpha)
        ref = [r * a + (1 - a) * s.detach() for r, s in zip(ref, student.parameters())]
        V.update_ema_variables(student, teacher, alpha, step)
        for r, t in zip(ref, teacher.parameters()):
            assert torch.allclose(r, t, atol=1e-7)
    # buffers (BatchNorm statistics) are not touched by the EMA (parameters() only)
    assert V.lr_at(0) == 1e-3 and abs(V.lr_at(450) - 1e-4) < 1e-12 and abs(V.lr_at(900) - 1e-6) < 1e-15
    assert V.bn_momentum_at(0) == 0.5 and V.bn_momentum_at(20) == 0.25
    assert V.bn_momentum_at(10000) == 0.001


def test_adam_state_round_trips_through_the_reference_layout():
    """The flat Adam state <-> the reference's per-parameter `optimizer_state_dict`
    (pretrain.py:196,374): exported state loads into a torch.optim.Adam over net.parameters(),
    continues identically there, and loads back."""
    _install_standins()
    V = importlib.import_module("3dioumatch_amd.votenet")
    cfg = V.scannet_config()
    runner = V.SupervisedStep(cfg, torch.device("cpu"), num_proposal=K)
    torch.manual_seed(1)
    runner(_batch(V, cfg, seed=7))
    sd = runner.optimizer_state_dict()
    assert len(sd["state"]) == len(list(runner.net.parameters())) == 96
    clone = V.SupervisedStep(cfg, torch.device("cpu"), num_proposal=K)
    clone.net.load_state_dict(runner.net.state_dict())
    ref_opt = torch.optim.Adam(clone.net.parameters(), lr=1e-3)
    ref_opt.load_state_dict(sd)  # the reference's optimizer accepts it as is
    # one more identical step on both: flat Adam vs per-parameter Adam
    batch = _batch(V, cfg, seed=8)
    torch.manual_seed(2)
    runner(dict(batch))
    torch.manual_seed(2)
    for p in clone.net.parameters():
        p.grad = None
    ep = clone.net(dict(batch), mode="jitter")
    ep.update(batch)
    loss, _ = V.get_labeled_loss(ep, cfg, {"dataset_config": cfg})
    loss.backward()
    ref_opt.step()
    a = torch.cat([p.detach().reshape(-1) for p in runner.net.parameters()])
    b = torch.cat([p.detach().reshape(-1) for p in clone.net.parameters()])
    assert float((a - b).abs().max()) <= 1e-6
    fresh = V.SupervisedStep(cfg, torch.device("cpu"), num_proposal=K)
    fresh.load_optimizer_state_dict(ref_opt.state_dict())
    back = fresh.optimizer_state_dict()
    want = ref_opt.state_dict()
    for i in range(96):
        assert torch.allclose(back["state"][i]["exp_avg"], want["state"][i]["exp_avg"])
        assert float(back["state"][i]["step"]) == float(want["state"][i]["step"]) == 2.0


def _semi_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _install_standins()
    V = importlib.import_module("3dioumatch_amd.votenet")
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    unl = importlib.import_module("3dioumatch_amd.votenet.losses_unlabeled")
    cfg = V.scannet_config()
    filt = unl.default_config_dict(cfg, unlabeled_batch_size=2)
    filt.update(obj_threshold=0.3, cls_threshold=0.03, iou_threshold=0.2)
    runner = V.SemiSupervisedStep(cfg, torch.device("cpu"), world_size=world, num_proposal=K,
                                  config_dict=filt)
    torch.manual_seed(100 + rank)
    loss, ep = runner(V.make_semi_batch(1, 2, N, cfg, seed=300 + rank, num_objects=5))
    np.savez(os.path.join(out_dir, "semi%d.npz" % rank),
             grad=step_mod.flat_grads(runner.net).numpy(),
             params=step_mod.flat_params(runner.net).numpy(),
             teacher=step_mod.flat_params(runner.teacher).numpy(),
             loss=float(loss), pseudo=int(ep["unlabeled_box_label_mask"].sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_semi_supervised_ddp_two_ranks_gloo(tmp_path):
    """Stage-2 step data-parallel over 2 ranks: after the single gradient all-reduce both ranks
    hold the same gradient, hence identical student AND teacher (EMA) weights, from different data."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_semi_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "semi0.npz")
    r1 = np.load(tmp_path / "semi1.npz")
    assert np.array_equal(r0["grad"], r1["grad"])
    assert np.array_equal(r0["params"], r1["params"])
    assert np.array_equal(r0["teacher"], r1["teacher"])
    assert np.isfinite(r0["loss"]) and np.isfinite(r1["loss"]) and r0["loss"] != r1["loss"]
    assert r0["pseudo"] > 0 and r1["pseudo"] > 0


# ------------------------------------------------------------ N > 1 WITH HIP graphs, on a GPU
def _gpu_graph_worker(rank, world, port, out_dir, semi):
    """Two processes share cuda:0 (backend gloo moves the flat gradient through the host): the
    captured path of step.py -- G1 replay -> eager all-reduce -> G2 replay -- with world_size 2."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    load_pkg()
    V = importlib.import_module("3dioumatch_amd.votenet")
    step_mod = importlib.import_module("3dioumatch_amd.votenet.step")
    cfg = V.scannet_config()

    def batch(step):
        if semi:
            return V.make_semi_batch(1, 2, 4096, cfg, seed=300 + rank + 10 * step, num_objects=5,
                                     device=dev)
        data = importlib.import_module("3dioumatch_amd.votenet.data")
        return data.make_batch(2, 4096, cfg, seed=200 + rank + 10 * step, num_objects=5, device=dev)

    def make(graphs, w):
        if semi:
            unl = importlib.import_module("3dioumatch_amd.votenet.losses_unlabeled")
            filt = unl.default_config_dict(cfg, unlabeled_batch_size=2)
            filt.update(obj_threshold=0.3, cls_threshold=0.03, iou_threshold=0.2)
            runner = V.SemiSupervisedStep(cfg, dev, world_size=w, num_proposal=K, graphs=graphs,
                                          config_dict=filt)
        else:
            runner = V.SupervisedStep(cfg, dev, world_size=w, num_proposal=K, graphs=graphs)
        # the parameters whose gradient is mathematically zero stay put (both arms): Adam would turn
        # their round-off into +-lr moves of either sign and the multi-step comparison into noise
        step_mod.freeze_shift_invariant_parameters(runner.net)
        return runner

    # Both arms step side by side and, after steps 0 and 1, the eager arm takes over the graph arm's
    # weights: the step is a discontinuous function of its weights at round-off scale (a max-pool
    # winner, a nearest pseudo label; tests/test_train_step.py::test_graph_replay_matches_eager), so
    # the arms are compared from a COMMON point at every step, never along two trajectories.
    out = {}
    runners = {"graph": make(True, world), "eager": make(False, world)}
    for s in range(3):
        for name, runner in runners.items():
            torch.manual_seed(100 + rank + 1000 * s)
            torch.cuda.manual_seed_all(100 + rank + 1000 * s)
            loss, _ = runner(batch(s))
            out[name + "_loss"] = float(loss)
            if s == 0:
                out[name + "_grad"] = runner.flat_grad.detach().cpu().numpy().copy()
        torch.cuda.synchronize(dev)
        if s < 2:
            runners["eager"].flat_params.data.copy_(runners["graph"].flat_params.data)
            if semi:
                runners["eager"].flat_teacher.data.copy_(runners["graph"].flat_teacher.data)
    for name, runner in runners.items():
        assert bool(runner.graphs) == (name == "graph"), "graph capture fell back to eager"
        out[name + "_params"] = runner.flat_params.detach().cpu().numpy().copy()
        if semi:
            out[name + "_teacher"] = runner.flat_teacher.detach().cpu().numpy().copy()
    # this rank's own gradient of step 0, computed without any exchange
    single = make(False, 1)
    torch.manual_seed(100 + rank)
    single(batch(0))
    out["single_grad"] = single.flat_grad.detach().cpu().numpy().copy()
    out["sizes"] = np.array([p.numel() for p in single._params], dtype=np.int64)
    np.savez(os.path.join(out_dir, "g%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("semi", [False, True], ids=["supervised", "semi"])
def test_graph_step_two_ranks_share_one_gpu(tmp_path, semi):
    """The HIP-graph N > 1 path: 3 steps, parameters bit-identical across
    ranks, equal to the eager N > 1 path stepping from the same weights, and the exchanged gradient
    of step 0 equal to the mean of the two single-process gradients."""
    world = 2
    port = 33500 + (os.getpid() % 2000) + (7 if semi else 0)
    mp.spawn(_gpu_graph_worker, args=(world, port, str(tmp_path), semi), nprocs=world, join=True)
    r = [np.load(tmp_path / ("g%d.npz" % i)) for i in range(world)]
    for key in ("graph_params", "eager_params") + (("graph_teacher",) if semi else ()):
        assert np.array_equal(r[0][key], r[1][key]), key
    assert np.array_equal(r[0]["graph_grad"], r[1]["graph_grad"])
    rel = lambda a, b: np.linalg.norm(a - b) / max(1e-12, np.linalg.norm(b))  # noqa: E731
    # graph replay == eager launches: the exchanged gradient of step 0 (same weights) agrees to
    # the reordering of atomic sums; with the zero-gradient parameters frozen
    # (step.freeze_shift_invariant_parameters: the last BatchNorm bias of every pooling module,
    # profiles/r4_step_repeatability.txt) the weights after three Adam steps agree too
    # (semi-supervised: the runner draws its box-jitter noise on the main stream AHEAD of its two
    # forward graphs -- inputs["jitter_noise"] -- so the concurrent replays no longer share torch's
    # one device-side (seed, offset) pair per generator; tools/semi_step_branches.py in two
    # processes, profiles/r6_semi_step_branches.txt: every run on one branch.  Same bounds as the
    # supervised arm: a wrong 1 / world or a missing tensor is >= 1e-2)
    g_tol = 1e-4
    assert rel(r[0]["graph_grad"], r[0]["eager_grad"]) < g_tol
    lr = 2e-3 if semi else 1e-3
    assert np.abs(r[0]["graph_params"] - r[0]["eager_params"]).max() <= 3 * 3 * lr
    print("params after 3 steps, graph vs eager: rel %.2e" % rel(r[0]["graph_params"], r[0]["eager_params"]))
    assert rel(r[0]["graph_params"], r[0]["eager_params"]) < 1e-2
    want = (r[0]["single_grad"] + r[1]["single_grad"]) / 2
    assert rel(r[0]["graph_grad"], want) < g_tol, rel(r[0]["graph_grad"], want)
    # ... and for EVERY parameter tensor, not only in the global norm: a wrong 1/world (or a tensor
    # left out of the exchange) on a small tensor would be a 100 % error there and invisible above
    errs, off, floor = [], 0, 1e-4 * np.linalg.norm(want)
    for n_el in r[0]["sizes"]:
        g_t, w_t = r[0]["graph_grad"][off:off + n_el], want[off:off + n_el]
        errs.append(np.linalg.norm(g_t - w_t) / (np.linalg.norm(w_t) + floor))
        off += int(n_el)
    errs = np.sort(np.array(errs))[::-1]
    assert off == want.size
    assert errs[0] < 2e-2, errs[:6]
    assert np.isfinite(r[0]["graph_loss"]) and r[0]["graph_loss"] != r[1]["graph_loss"]


# ------------------------------------------------------------ RCCL itself, on one GPU
def _rccl_worker(rank, world, port, out_dir):
    """A ONE-rank `nccl` (= RCCL) process group on cuda:0: init with device_id, the flat-gradient
    all-reduce on RCCL's stream between the replays of the backward graph and the update graph."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    load_pkg()
    V = importlib.import_module("3dioumatch_amd.votenet")
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    cfg = V.scannet_config()
    batch = lambda s: data.make_batch(2, 4096, cfg, seed=400 + s, num_objects=5, device=dev)  # noqa: E731
    out = {}
    for name, with_group in (("plain", False), ("rccl", True), ("rccl_in_graph", True)):
        if with_group and not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        runner = V.SupervisedStep(cfg, dev, world_size=1, num_proposal=K, graphs=True)
        importlib.import_module("3dioumatch_amd.votenet.step").freeze_shift_invariant_parameters(runner.net)
        runner.exchange_always = with_group
        runner.time_exchange = name == "rccl"
        runner.capture_exchange = name == "rccl_in_graph"  # the all-reduce as the first node of G2
        torch.manual_seed(7)
        losses = []
        for s in range(5):
            loss, _ = runner(batch(s))
            losses.append(float(loss))
        assert bool(runner.graphs), "graph capture fell back to eager"
        torch.cuda.synchronize(dev)
        out[name + "_params"] = runner.flat_params.detach().cpu().numpy().copy()
        out[name + "_losses"] = np.array(losses)
        if with_group:
            rep = runner.exchange_report()
            assert rep["backend"] == "nccl" and rep["world_size"] == 1
            out[name + "_in_graph"] = np.array([int(rep["in_graph"])])
            out[name + "_us"] = np.array([-1.0 if rep["allreduce_us"] is None else rep["allreduce_us"]])
    out["backend"] = np.array([ord(c) for c in dist.get_backend()])
    np.savez(os.path.join(out_dir, "rccl.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_graph_step_with_a_one_rank_rccl_group(tmp_path):
    """RCCL on this stack before an 8-GPU node ever sees it: process-group init, the gradient
    all-reduce between G1 and G2 (five replays), finite losses, and -- the collective over one
    rank being the identity -- parameters equal to the run without a process group (up to the
    order of atomic sums: a few Adam steps of sign noise, as in the two-rank test)."""
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rccl.npz")
    assert "".join(chr(c) for c in r["backend"]) == "nccl"
    assert np.all(np.isfinite(r["rccl_losses"])) and np.all(np.isfinite(r["plain_losses"]))
    assert np.abs(r["rccl_losses"][0] - r["plain_losses"][0]) <= 1e-4 * abs(r["plain_losses"][0])
    assert np.abs(r["rccl_params"] - r["plain_params"]).max() <= 5 * 3 * 1e-3
    rel = np.linalg.norm(r["rccl_params"] - r["plain_params"]) / np.linalg.norm(r["plain_params"])
    print("params after 5 steps, one-rank rccl vs no group: rel %.2e" % rel)
    assert rel < 1e-2, rel  # (the zero-gradient parameters are frozen in both arms)
    # the eager all-reduce was timed (events on the launching stream) ...
    assert int(r["rccl_in_graph"][0]) == 0 and float(r["rccl_us"][0]) > 0
    # ... and the opt-in form with the collective captured at the head of G2 runs the same step
    # (when RCCL refuses the capture the runner falls back to the eager launch and says so)
    assert np.all(np.isfinite(r["rccl_in_graph_losses"]))
    assert np.abs(r["rccl_in_graph_params"] - r["plain_params"]).max() <= 5 * 3 * 1e-3
    print("all-reduce captured in the update graph:", bool(int(r["rccl_in_graph_in_graph"][0])),
          "| eager all-reduce of one rank: %.1f us" % float(r["rccl_us"][0]))
