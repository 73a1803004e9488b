#!/usr/bin/env python
"""bench.py -- scenes/sec of the VoteNet-IoU supervised train step (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One STEP = one pass of the hot path over one synthetic batch, i.e. the reference's
pretrain.py:train_one_epoch body (pretrain.py:310-347) at BASELINE config 2:
    forward_with_pred_jitter (backbone SA1-4 + FP1-2, voting, vote aggregation, IoU branch)
    -> get_labeled_loss (incl. two (B*K)x(B*64) 3-D IoU matrices) -> backward -> Adam,
B=8 scenes of 40 000 points (xyz + height), 256 proposals, fp32, random-init weights,
synthetic scenes (no dataset, no checkpoint).  Inputs are resident in HBM before the timed
region.  N > 1: one process per GPU, same per-GPU batch (weak scaling); the data-parallel
exchange is ONE all_reduce of the flat 4.26 MB gradient buffer per step (RCCL by default,
--backend gloo for a CPU-mediated run), between the backward graph and the Adam graph.

The JSON line also carries
  roofline     -- the north-star kernel pair ball_query + group_points(xyz) + group_points(feat)
                  at B=8, N=40000, m=2048, nsample=64: algorithmic bytes (38 516 736 B, SURVEY
                  section 8d) / measured duration (events on the launch stream), vs 8 TB/s HBM;
  cpu_baseline -- the same train step on the host cores with the oracle (OpenMP build) behind
                  the same Python modules, on a bounded sample (B=1), rank 0 / N=1 only;
  kernels      -- per-op device times (us) at the config-2 shapes, for the record.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (~6.3 TB/s achievable)
B, NPTS, KPROP = 8, 40000, 256
PAIR_BYTES = 38516736  # ball_query 8 230 912 + group(xyz) 20 617 216 + group(feat) 9 668 608


SEMI_LABELED, SEMI_UNLABELED = 4, 8  # train.py default batch_size "4,8"
WORKLOADS = {
    "pretrain": "ScanNet pretrain step (BASELINE configs[1]): VoteNet-IoU "
                "forward_with_pred_jitter + labeled loss + backward + Adam",
    "sunrgbd": "SUN RGB-D pretrain step (BASELINE configs[2]): 20000 pts, batch 16, 12 heading "
               "bins (oriented-box IoU), forward_with_pred_jitter + labeled loss + backward + Adam",
    "semi": "ScanNet semi-supervised step (BASELINE configs[3]): EMA teacher + student "
            "forward_with_pred_jitter on 4 labeled + 8 unlabeled scenes, labeled loss + "
            "pseudo-label consistency loss (device-side filter + LHS-NMS), backward, Adam, EMA",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--workload", choices=("pretrain", "semi", "sunrgbd"), default="pretrain",
                    help="pretrain = BASELINE configs[1] (the headline metric); semi = configs[3]: "
                         "stage-2 step, 4 labeled + 8 unlabeled scenes per GPU, EMA teacher; "
                         "sunrgbd = configs[2]: SUN RGB-D pretrain, 20000 pts, batch 16, oriented boxes")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="compute the FPS chain inline instead of one step ahead on a side stream")
    return ap.parse_args()


def build_step(V, cfg, device, world, local_rank, workload="pretrain"):
    if workload == "semi":
        runner = V.SemiSupervisedStep(cfg, device, world_size=world, num_proposal=KPROP, lr=2e-3,
                                      graphs_fallback=True)
    else:
        runner = V.SupervisedStep(cfg, device, world_size=world, num_proposal=KPROP, lr=1e-3,
                                  graphs_fallback=True)  # reported as config.hip_graphs

    def step(batch):
        return runner(batch)[0]

    step.prefetch = runner.prefetch_geometry
    step.runner = runner
    return step


def time_op(fn, iters=20, warm=3):
    """Average device time of fn() in us.  The calls are captured into one HIP graph and the
    replay is timed with events on the launch stream, so host launch overhead (ctypes + torch
    allocations, ~10-20 us per call) does not leak into kernels that only run for a few us;
    dependent kernel boundaries inside the op are of course included."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        s.record()
        graph.replay()
        e.record()
    except Exception:  # capture not possible: time the eager loop instead
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us


def kernel_table(device):
    """Device time of each hot-path operator at the config-2 shapes (us per call)."""
    ext = importlib.import_module("pointnet2._ext")
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    synth = importlib.import_module("3dioumatch_amd.synth")
    t = {}
    xyz = torch.from_numpy(synth.cloud_uniform(B, NPTS, synth.cube_side(NPTS, 0.2, 64), seed=1)).to(device)
    flipped = xyz.transpose(1, 2).contiguous()
    feat = torch.rand(B, 1, NPTS, device=device)
    t["fps_40000_2048"] = time_op(lambda: ext.furthest_point_sampling(xyz, 2048), iters=3, warm=1)
    inds = ext.furthest_point_sampling(xyz, 2048)
    new_xyz = ext.gather_points(flipped, inds).transpose(1, 2).contiguous()
    t["gather_3x2048"] = time_op(lambda: ext.gather_points(flipped, inds))
    t["ball_query_sa1"] = time_op(lambda: ext.ball_query(new_xyz, xyz, 0.2, 64))
    idx = ext.ball_query(new_xyz, xyz, 0.2, 64)
    t["group_xyz_sa1"] = time_op(lambda: ext.group_points(flipped, idx))
    t["group_feat_sa1"] = time_op(lambda: ext.group_points(feat, idx))
    t["query_and_group_sa1_fused"] = time_op(
        lambda: ext.query_and_group(new_xyz, xyz, feat, 0.2, 64, True))
    g4 = torch.rand(B, 4, 2048, 64, device=device)
    t["group_grad_sa1_c4"] = time_op(lambda: ext.group_points_grad(g4, idx, NPTS))
    # SA2-scale gather / scatter (the large-C case)
    xyz2 = new_xyz
    inds2 = ext.furthest_point_sampling(xyz2, 1024)
    t["fps_2048_1024"] = time_op(lambda: ext.furthest_point_sampling(xyz2, 1024), iters=5)
    new2 = ext.gather_points(xyz2.transpose(1, 2).contiguous(), inds2).transpose(1, 2).contiguous()
    idx2 = ext.ball_query(new2, xyz2, 0.4, 32)
    t["ball_query_sa2"] = time_op(lambda: ext.ball_query(new2, xyz2, 0.4, 32))
    f128 = torch.rand(B, 128, 2048, device=device)
    t["group_feat_sa2_c128"] = time_op(lambda: ext.group_points(f128, idx2))
    g128 = torch.rand(B, 128, 1024, 32, device=device)
    t["group_grad_sa2_c128"] = time_op(lambda: ext.group_points_grad(g128, idx2, 2048))
    grid = torch.rand(B, 32768, 3, device=device) * 3
    seeds = torch.rand(B, 1024, 3, device=device) * 3
    t["three_nn_gridconv"] = time_op(lambda: ext.three_nn(grid, seeds))
    d2, nidx = ext.three_nn(grid, seeds)
    w = torch.rand(B, 32768, 3, device=device)
    f256 = torch.rand(B, 256, 1024, device=device)
    t["three_interpolate_gridconv"] = time_op(lambda: ext.three_interpolate(f256, nidx, w))
    a, b = synth.boxes_pair(2048, seed=3)
    a_d, b_d = torch.from_numpy(a).to(device), torch.from_numpy(b[:512]).to(device)
    t["iou3d_2048x512"] = time_op(lambda: ut.boxes_iou3d_gpu(a_d, b_d))
    t["iou3d_256x256"] = time_op(lambda: ut.boxes_iou3d_gpu(a_d[:256], b_d[:256]))
    pair_us = t["ball_query_sa1"] + t["group_xyz_sa1"] + t["group_feat_sa1"]
    return {k: round(v, 2) for k, v in t.items()}, pair_us


def cpu_baseline(V, cfg, steps=5):
    """The same supervised step on the host cores: oracle (OpenMP) behind the same modules."""
    from oracle.oracle import Oracle
    from oracle import standin
    o = Oracle(omp=True)
    utils = importlib.import_module("pointnet2.pointnet2_utils")
    losses = importlib.import_module("3dioumatch_amd.votenet.losses")
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    real_ext, real_iou = utils._ext, losses.boxes_iou3d_gpu
    utils._ext = standin.make(o)
    losses.boxes_iou3d_gpu = lambda a, b: torch.from_numpy(
        o.boxes_iou3d(a.detach().numpy(), b.detach().numpy()))
    try:
        cpu = torch.device("cpu")
        step = build_step(V, cfg, cpu, 1, 0)
        batch = data.make_batch(1, NPTS, cfg, seed=7, device=cpu)
        batch = dict(batch)
        step(batch)  # warm-up
        t0 = time.perf_counter()
        for _ in range(steps):
            step(batch)
        dt = (time.perf_counter() - t0) / steps
    finally:
        utils._ext, losses.boxes_iou3d_gpu = real_ext, real_iou
    return {"value": round(1.0 / dt, 4), "unit": "scenes/s", "cores": o.cores, "kind": "port",
            "sample": "same train step at B=1 (1 scene x 40000 pts, 256 proposals), %d timed "
                      "steps after 1 warm-up; custom ops = oracle (OpenMP), MLPs = torch CPU "
                      "(%d threads)" % (steps, torch.get_num_threads())}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group("gloo")
    importlib.import_module("3dioumatch_amd")
    V = importlib.import_module("3dioumatch_amd.votenet")
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    cfg = V.sunrgbd_config() if args.workload == "sunrgbd" else V.scannet_config()
    npts = 20000 if args.workload == "sunrgbd" else NPTS

    step = build_step(V, cfg, device, world, local_rank, args.workload)
    if args.workload == "semi":
        scenes = SEMI_LABELED + SEMI_UNLABELED
        batch = data.make_semi_batch(SEMI_LABELED, SEMI_UNLABELED, NPTS, cfg, seed=100 + rank,
                                     device=device)
    elif args.workload == "sunrgbd":
        scenes = 16
        batch = data.make_batch(scenes, npts, cfg, seed=100 + rank, device=device)
    else:
        scenes = B
        batch = data.make_batch(B, NPTS, cfg, seed=100 + rank, device=device)  # resident in HBM

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # Software-pipelined loop, as a training loop with a prefetching loader would run it: the
    # coordinate-only index computations (FPS chain) of batch i+1 are launched on a side stream
    # right before step i, so they overlap step i's dense kernels.  Every step still computes
    # exactly one set of FPS indices (the first one is computed before the timed region, the
    # one prefetched during the last timed step is consumed after it).
    pipelined = not args.no_prefetch
    views = [dict(batch), dict(batch)]  # two views of the resident batch: current / next
    if pipelined:
        step.prefetch(views[0])
    history = torch.zeros(args.warmup + args.steps, device=device)  # loss per step (diagnostics)
    for i in range(args.warmup):
        if pipelined:
            step.prefetch(views[(i + 1) % 2])
        history[i].copy_(step(views[i % 2]).detach())
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        if pipelined:
            step.prefetch(views[(i + 1) % 2])
        loss = step(views[i % 2])
        history[i].copy_(loss.detach())
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(history).all().item(), "loss diverged: %s" % history.tolist()

    if rank == 0:
        ms = elapsed * 1e3 / args.steps
        out = {
            "metric": "scenes/sec train-step (ScanNet 40k pts, 256 proposals)" if args.workload != "sunrgbd"
            else "scenes/sec train-step (SUN RGB-D 20k pts, 256 proposals)",
            "value": round(scenes * world * args.steps / elapsed, 3), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload],
                       "per_gpu_batch": scenes, "global_batch": scenes * world, "num_points": npts,
                       "num_proposals": KPROP, "parallelism": "dp%d" % world,
                       "fps_prefetch_one_step_ahead": pipelined,
                       "hip_graphs": bool(step.runner.graphs)},
        }
        if not args.no_kernels:
            table, pair_us = kernel_table(device)
            achieved = PAIR_BYTES / (pair_us * 1e-6) / 1e9
            # HBM bytes per pair from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
            # WRITE_SIZE in separate runs, gfx950 correction applied; profiles/r1_pair_pmc_v3.json)
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "r1_pair_pmc_v3.json")
            if os.path.exists(pmc):
                traffic = json.load(open(pmc)).get("traffic_bytes_per_pair")
            out["roofline"] = {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "kernel": "ball_query + group_points(xyz,C=3) + group_points(feat,C=1) @ B=8 "
                          "N=40000 m=2048 ns=64", "algorithmic_bytes": PAIR_BYTES,
                "duration_us": round(pair_us, 2)}
            out["kernels_us"] = table
        if world == 1 and not args.no_cpu_baseline and args.workload == "pretrain":
            out["cpu_baseline"] = cpu_baseline(V, cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()  # rank 0 is still timing the per-operator table
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
