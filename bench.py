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
  roofline     -- the north-star pair ball_query + group_points(xyz) + group_points(feat) at B=8,
                  N=40000, m=2048, nsample=64: algorithmic bytes (38 516 736 B)
                  / measured duration (events on the launch stream around HIP-graph replays) vs
                  8 TB/s HBM.  `frac` is the DOMINANT KERNEL: the fused query + gather kernel on
                  the cell lists the layer's sampling kernel leaves behind (how SA1 runs); the
                  self-contained operator (cell-list build included) and the three calls of the
                  reference's operator surface are reported next to it (`forms`);
  kernels      -- every hot-path kernel at the config-2 shapes: us, algorithmic bytes or flops,
                  the bound, the fraction of that roofline;
  ms_per_step_no_prefetch -- the same step with the coordinate-only index chain run inline;
  cpu_baseline -- the same train step on the host cores with the oracle (OpenMP build) behind
                  the same Python modules, on a bounded sample (B=1), rank 0 / N=1 only.
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
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
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
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the short passes of the two other BASELINE workloads (N = 1 default run)")
    ap.add_argument("--workload", choices=("pretrain", "semi", "sunrgbd"), default="pretrain",
                    help="pretrain = BASELINE configs[1] (the headline metric); semi = configs[3]: "
                         "stage-2 step, 4 labeled + 8 unlabeled scenes per GPU, EMA teacher; "
                         "sunrgbd = configs[2]: SUN RGB-D pretrain, 20000 pts, batch 16, oriented boxes")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="compute the FPS chain inline instead of one step ahead on a side stream")
    ap.add_argument("--rotate", type=int, default=4,
                    help="K distinct batches (seeds 100 .. 100+K-1) kept in pinned host memory; batch "
                         "i+2 is copied to the device on a copy stream while step i runs, INSIDE the "
                         "timed region (what a training loop with a DataLoader does, "
                         "pretrain.py:310-347).  `value` is measured on this loop; 0 = only the "
                         "resident-batch loop of rounds 1-4 (then `value` = `value_resident`)")
    return ap.parse_args()


def newest_profile(suffix):
    """profiles/r<N>_<suffix> of the latest round that committed one (None if there is none)."""
    for rnd in range(9, 0, -1):
        path = os.path.join(ROOT, "profiles", "r%d_%s" % (rnd, suffix))
        if os.path.exists(path):
            return path
    return None


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


TIME_OP_EAGER = []  # names of operators whose graph capture failed (timed as an eager loop)
COPY_STREAM = {}  # the feeding loops' host -> device stream (one per process)


def time_op(fn, iters=20, warm=3, name=None):
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
    except Exception as exc:  # capture not possible: time the eager loop instead, and say so
        TIME_OP_EAGER.append("%s: %s" % (name or getattr(fn, "__name__", "op"), type(exc).__name__))
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us


def pair_cloud(kind, seed=1):
    """(B, NPTS, 3) cloud of the north-star pair: 'U' = SURVEY 8(d) cloud U(L) (uniform, expected
    in-ball count = nsample), 'R' = cloud R (room shell), 'step' = the xyz of the batch the timed
    train step runs on (votenet/data.py:make_batch, seed 100: half the points inside <= 12 object
    boxes -- dense balls)."""
    synth = importlib.import_module("3dioumatch_amd.synth")
    if kind == "U":
        return torch.from_numpy(synth.cloud_uniform(B, NPTS, synth.cube_side(NPTS, 0.2, 64), seed=seed))
    if kind == "R":
        return torch.from_numpy(synth.cloud_room(B, NPTS, seed=seed))
    assert kind == "step", kind
    V = importlib.import_module("3dioumatch_amd.votenet")
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    batch = data.make_batch(B, NPTS, V.scannet_config(), seed=100)
    return batch["point_clouds"][:, :, :3].contiguous()


def kernel_table(device):
    """Every hot-path operator at the config-2 shapes: device time (us per call, events around
    HIP-graph replays), algorithmic bytes / flops / tests per SURVEY section 8(d), the bound and the
    achieved fraction of that roofline.  Returns (table, forms of the north-star pair)."""
    ext = importlib.import_module("pointnet2._ext")
    ut = importlib.import_module("pcdet.ops.iou3d_nms.iou3d_nms_utils")
    synth = importlib.import_module("3dioumatch_amd.synth")
    t = {}

    def hbm(name, us, nbytes):
        t[name] = {"us": round(us, 2), "bytes": int(nbytes), "bound": "hbm",
                   "GBps": round(nbytes / us / 1e3, 1),
                   "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}

    def valu(name, us, work, unit, note):
        t[name] = {"us": round(us, 2), "bound": note, unit: int(work),
                   "G%s_per_s" % unit: round(work / us / 1e3, 2)}

    m1, ns1, m2, ns2 = 2048, 64, 1024, 32
    xyz = torch.from_numpy(synth.cloud_uniform(B, NPTS, synth.cube_side(NPTS, 0.2, 64), seed=1)).to(device)
    flipped = xyz.transpose(1, 2).contiguous()
    feat = torch.rand(B, 1, NPTS, device=device)
    us = time_op(lambda: ext.furthest_point_sampling(xyz, m1), iters=3, warm=1)
    valu("fps_40000_2048", us, B * (m1 - 1) * NPTS, "dist_updates",
         "latency of 2047 dependent rounds (8 workgroups); brute-force-equivalent updates")
    inds, lists = ext.furthest_point_sampling_with_grid(xyz, m1, 0.2)
    us = time_op(lambda: ext.furthest_point_sampling_with_grid(xyz, m1, 0.2), iters=3, warm=1)
    valu("fps_40000_2048_with_cell_lists", us, B * (m1 - 1) * NPTS, "dist_updates",
         "the same kernel also leaving SA1's cell lists behind")
    new_xyz = ext.gather_points(flipped, inds).transpose(1, 2).contiguous()
    # the layer's own centroids (pointnet2_modules.py:236-245): the sampling kernel's query plans apply
    lists.mark_centroids(new_xyz, inds)
    hbm("gather_3x2048", time_op(lambda: ext.gather_points(flipped, inds)), 4 * B * m1 + 8 * B * 3 * m1)
    bq_bytes = 12 * B * NPTS + 12 * B * m1 + 4 * B * m1 * ns1
    gx_bytes = 4 * B * 3 * NPTS + 4 * B * m1 * ns1 + 4 * B * 3 * m1 * ns1
    gf_bytes = 4 * B * 1 * NPTS + 4 * B * m1 * ns1 + 4 * B * 1 * m1 * ns1
    assert bq_bytes + gx_bytes + gf_bytes == PAIR_BYTES
    idx = ext.ball_query(new_xyz, xyz, 0.2, ns1)
    hbm("ball_query_sa1", time_op(lambda: ext.ball_query(new_xyz, xyz, 0.2, ns1)), bq_bytes)
    hbm("ball_query_sa1_prebuilt_lists", time_op(
        lambda: ext.ball_query_prebuilt(new_xyz, xyz, 0.2, ns1, lists)), bq_bytes)
    hbm("group_xyz_sa1", time_op(lambda: ext.group_points(flipped, idx)), gx_bytes)
    hbm("group_feat_sa1", time_op(lambda: ext.group_points(feat, idx)), gf_bytes)
    hbm("cell_list_build_sa1", time_op(lambda: ext.build_grid(xyz, 0.2)), 12 * B * NPTS + 16 * B * NPTS)

    def api():
        i = ext.ball_query(new_xyz, xyz, 0.2, ns1)
        ext.group_points(flipped, i)
        ext.group_points(feat, i)

    forms = {
        "layer": time_op(lambda: ext.query_and_group(new_xyz, xyz, feat, 0.2, ns1, True, None, lists)),
        # the same kernel without the sampling kernel's query plans (centroids not known to be its picks)
        "layer_no_plan": time_op(lambda: ext.query_and_group(new_xyz.clone(), xyz, feat, 0.2, ns1, True, None, lists)),
        "self_contained": time_op(lambda: ext.query_and_group(new_xyz, xyz, feat, 0.2, ns1, True)),
        "reference_api_3_calls": time_op(api),
    }
    # the reference surface with _ext's per-cloud list cache: the lists the sampling call (or the
    # first query) of this cloud left behind are found again -- what a drop-in user's
    # furthest_point_sample -> ball_query -> group -> group sequence pays after the first step
    # (the capture only sees constant inputs, so cache hits inside it are sound)
    with ext.lists_cached_during_capture():
        forms["reference_api_3_calls_cached_lists"] = time_op(api)
    hbm("query_and_group_sa1_fused_kernel", forms["layer"], PAIR_BYTES)
    # the same kernel, `layer` form, on the two other clouds of SURVEY 8(d) / the timed step:
    # cloud R (room shell) and the xyz of the batch the timed step itself runs on (dense objects)
    for kind in ("R", "step"):
        xk = pair_cloud(kind).to(device)
        ik, lk = ext.furthest_point_sampling_with_grid(xk, m1, 0.2)
        nk = ext.gather_points(xk.transpose(1, 2).contiguous(), ik).transpose(1, 2).contiguous()
        lk.mark_centroids(nk, ik)
        forms["layer_cloud_" + kind] = time_op(
            lambda: ext.query_and_group(nk, xk, feat, 0.2, ns1, True, None, lk))
    # the same kernel with its inputs and outputs rotating through 12 distinct sets (12 x 26 MB >
    # the 256 MB Infinity Cache): back-to-back replays on ONE set find their 7 MB of reads in
    # L2 / MALL, which flatters a fraction quoted against HBM
    sets = []
    for q in range(12):
        xq = torch.from_numpy(synth.cloud_uniform(B, NPTS, synth.cube_side(NPTS, 0.2, 64), seed=10 + q)).to(device)
        iq, lq = ext.furthest_point_sampling_with_grid(xq, m1, 0.2)
        nq = ext.gather_points(xq.transpose(1, 2).contiguous(), iq).transpose(1, 2).contiguous()
        lq.mark_centroids(nq, iq)
        sets.append((nq, xq, torch.rand(B, 1, NPTS, device=device), lq))
    keep = []

    def rotating():
        for nq, xq, fq, lq in sets:
            keep.append(ext.query_and_group(nq, xq, fq, 0.2, ns1, True, None, lq))

    forms["layer_rotating_inputs"] = time_op(rotating, iters=2, warm=1, name="pair_rotating") / len(sets)
    del keep[:]
    g4 = torch.rand(B, 4, m1, ns1, device=device)
    hbm("group_grad_sa1_c4", time_op(lambda: ext.group_points_grad(g4, idx, NPTS)),
        4 * B * 4 * m1 * ns1 + 4 * B * m1 * ns1 + 4 * B * 4 * NPTS)
    # SA2-scale (the large-C case)
    xyz2 = new_xyz
    us = time_op(lambda: ext.furthest_point_sampling(xyz2, m2), iters=5)
    valu("fps_2048_1024", us, B * (m2 - 1) * m1, "dist_updates", "latency of 1023 dependent rounds")
    inds2 = ext.furthest_point_sampling(xyz2, m2)
    new2 = ext.gather_points(xyz2.transpose(1, 2).contiguous(), inds2).transpose(1, 2).contiguous()
    idx2 = ext.ball_query(new2, xyz2, 0.4, ns2)
    us = time_op(lambda: ext.ball_query(new2, xyz2, 0.4, ns2))
    valu("ball_query_sa2", us, B * m2 * m1, "tests", "VALU (brute-force tier: B*m*N distance tests)")
    f128 = torch.rand(B, 128, m1, device=device)
    c128 = 4 * B * 128 * m1 + 4 * B * m2 * ns2 + 4 * B * 128 * m2 * ns2
    hbm("group_feat_sa2_c128", time_op(lambda: ext.group_points(f128, idx2)), c128)
    g128 = torch.rand(B, 128, m2, ns2, device=device)
    hbm("group_grad_sa2_c128_atomic", time_op(lambda: ext.group_points_grad(g128, idx2, m1)), c128)
    inv2 = ext.group_inverse(idx2, m1)  # built once per batch in the prefetched index chain
    hbm("group_grad_sa2_c128", time_op(lambda: ext.group_points_grad_sorted(g128, inv2, m1)), c128)
    hbm("group_inverse_sa2", time_op(lambda: ext.group_inverse(idx2, m1)), 8 * B * m2 * ns2)
    ng, ms = 32768, 1024
    grid = torch.rand(B, ng, 3, device=device) * 3
    seeds = torch.rand(B, ms, 3, device=device) * 3
    us = time_op(lambda: ext.three_nn(grid, seeds))
    valu("three_nn_gridconv", us, B * ng * ms, "tests", "VALU (n*m distance tests, 3-slot insertion)")
    d2, nidx = ext.three_nn(grid, seeds)
    w = torch.rand(B, ng, 3, device=device)
    f256 = torch.rand(B, 256, ms, device=device)
    hbm("three_interpolate_gridconv", time_op(lambda: ext.three_interpolate(f256, nidx, w)),
        4 * B * 256 * ms + 24 * B * ng + 4 * B * 256 * ng)
    a, b = synth.boxes_pair(2048, seed=3)
    a_d, b_d = torch.from_numpy(a).to(device), torch.from_numpy(b[:512]).to(device)
    us = time_op(lambda: ut.boxes_iou3d_gpu(a_d, b_d))
    valu("iou3d_2048x512", us, 2048 * 512, "pairs", "VALU + transcendentals (~1 kFLOP per pair)")
    us = time_op(lambda: ut.boxes_iou3d_gpu(a_d[:256], b_d[:256]))
    valu("iou3d_256x256", us, 256 * 256, "pairs", "VALU + transcendentals; launch latency at this size")
    # shared-MLP GEMMs (forward, BN+ReLU folded into the operand load) at two network shapes
    K = importlib.import_module("pointnet2._mlp_ext")
    for name, r, mm, kk in (("mlp_fwd_sa2_128x128", 32768, 128, 128), ("mlp_fwd_sa1_128x64", 131072, 128, 64)):
        wgt = torch.randn(mm, kk, device=device) / kk ** 0.5
        x = torch.randn(B, kk, r, device=device)
        coeff = (torch.rand(kk, device=device) + 0.5, torch.rand(kk, device=device))
        us = time_op(lambda: K.gemm_forward(wgt, x, coeff), iters=5, warm=2)
        flops = 2.0 * B * mm * kk * r
        nbytes = 4.0 * B * r * (mm + kk)  # the operand read once, the result written once
        # since round 4 the fp32 products run as six bf16 MFMAs (exact three-term split): the
        # layer is bound by its compulsory HBM traffic, `frac` is of the 8 TB/s roofline;
        # `frac_of_fp32_mfma_peak` keeps the earlier rounds' figure (flops / 157.3 TFLOP/s)
        t[name] = {"us": round(us, 2), "flops": int(flops), "bytes": int(nbytes),
                   "bound": "hbm (fp32 products on the bf16 matrix pipe: 6 x mfma_f32_32x32x16_bf16 per 16 k)",
                   "TFLOPs": round(flops / us * 1e-6, 1),
                   "frac_of_fp32_mfma_peak": round(flops / (us * 1e-6) / 1e12 / F32_PEAK_TFLOPS, 4),
                   "hbm_GBps": round(nbytes / us / 1e3, 1),
                   "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
    # SA1's shared MLP as the step runs it since round 5: layers 2 + 3 chained in registers
    # (csrc/mlp_chain.hip: statistics pass + full pass, y3 not stored) and the last layer's backward
    # from the Gram matrix of its input (csrc/mlp_pool_gram.hip)
    x4 = torch.randn(B, 4, m1, ns1, device=device)
    w0 = torch.randn(64, 4, device=device) * 0.7
    w1c = torch.randn(64, 64, device=device) / 8
    w2c = torch.randn(128, 64, device=device) / 8
    bnp = lambda c: [torch.rand(c, device=device) + 0.5, torch.randn(c, device=device) * 0.3,  # noqa: E731
                     torch.zeros(c, device=device), torch.ones(c, device=device)]
    p0, p1, p2 = bnp(64), bnp(64), bnp(128)
    if K.chain_lin4_supported(w0, w1c, w2c, x4, ns1):
        mom = K.first4_moments(x4)
        c0 = K.first4_bn(mom, x4.numel() // 4, w0, p0[0], p0[1], p0[2], p0[3], 0.1, 1e-5)

        def chain():
            return K.chain_lin4_forward(x4, w0, (c0[2], c0[3]), (w1c, *p1, 0.1, 1e-5), (w2c, *p2, 0.1, 1e-5),
                                        store_last=False)

        us = time_op(chain, iters=5, warm=2)
        cols = B * m1 * ns1
        flops = 2.0 * cols * (64 * 64 * 2 + 128 * 64)  # layer 2 twice (statistics pass + full pass)
        nbytes = 4.0 * cols * (4 * 2 + 64)              # x4 read twice, y2 written once
        t["mlp_chain_fwd_sa1"] = {
            "us": round(us, 2), "flops": int(flops), "bytes": int(nbytes),
            "bound": "vector-instruction issue (4 cycles per wave instruction, ~1200 per 32-column tile "
                     "against 144 MFMAs; profiles/r5_chain_pmc.json)",
            "TFLOPs": round(flops / us * 1e-6, 1), "hbm_GBps": round(nbytes / us / 1e3, 1),
            "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "replaces mlp_fwd 64x64 (lin4) + 128x64 with their statistics / pool epilogues: "
                    "y3 (537 MB) is never written"}
        y1c, c1c, _, c2c, extc = chain()
        _, amax, ymax_ = K.pool_from_extrema(extc, c2c[2], c2c[3])
        dpool = torch.randn(B, 128, m1, device=device)
        _, _, coef3 = K.bn_relu_pool_backward_stats(None, dpool, amax, ymax_, p2[0], c2c[2], c2c[3], c2c[0],
                                                    c2c[1], True, ns=ns1)
        if K.pool_gram_supported(w2c, y1c, ns1):
            us = time_op(lambda: K.pool_gram_backward(w2c, y1c, c1c, p1[0], coef3, c2c, dpool, amax, ymax_,
                                                      ns1, True), iters=5, warm=2)
            nbytes = 4.0 * cols * (64 + 64)  # y2 read once, d relu(bn(y2)) written once
            t["mlp_gram_bwd_sa1_128x64"] = {
                "us": round(us, 2), "bytes": int(nbytes), "bound": "hbm / per-chunk latency (one workgroup per CU)",
                "hbm_GBps": round(nbytes / us / 1e3, 1),
                "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "backward of the pooled 64 -> 128 layer without y3: Gram matrix + sparse image"}
    # SA2's pooled 128 -> 256 layer as the step runs it since round 6: forward with statistics + extrema
    # and NO store, backward from the Gram matrix of its input (csrc/mlp_pool_gram256.hip: two passes)
    gm, gns = 1024, 32
    gy2 = torch.randn(B, 128, gm, gns, device=device) * 1.3 + 0.2
    gw3 = torch.randn(256, 128, device=device) / 11
    gq2, gq3 = bnp(128), bnp(256)
    gc2 = K.bn_coefficients(gy2, gq2[0], gq2[1], gq2[2], gq2[3], 0.1, 1e-5, True)
    if K.pool_gram_supported(gw3, gy2, gns) and K.forward_pool_supported(gw3, gy2, (gc2[2], gc2[3])):
        gfwd = lambda: K.gemm_forward_bn(gw3, gy2, (gc2[2], gc2[3]), gq3[0], gq3[1], gq3[2], gq3[3], 0.1,  # noqa: E731
                                         1e-5, pool=True, store=False)
        us = time_op(gfwd, iters=5, warm=2)
        gcols = B * gm * gns
        t["mlp_fwd_sa2_256x128_no_store"] = {
            "us": round(us, 2), "flops": int(2.0 * gcols * 256 * 128), "bytes": int(4.0 * gcols * 128),
            "bound": "mfma + the vector instructions between them (csrc/mlp_pool_fwd256.hip: 48 MFMAs and ~230 "
                     "vector instructions per wave and 32-column chunk; a wave's vector instructions do not "
                     "overlap its MFMAs: profiles/r6_mfma_bf16_peak.json) -- not its 134 MB",
            "TFLOPs": round(2.0 * gcols * 256 * 128 / us * 1e-6, 1),
            "frac_of_mfma_floor": round(6.0 * 2.0 * gcols * 256 * 128 / 2.3e15 / (us * 1e-6), 3),
            "note": "persistent T-form kernel + the two launches that finish the statistics: BatchNorm pairs "
                    "+ pooled extrema from in-lane scans, y3 (268 MB) not written; floor = six bf16 MFMAs per "
                    "product at the measured 2.3 PFLOP/s"}
        _, gmean3, ginv3, gsc3, gsh3, gext = gfwd()
        _, gamax, gymax = K.pool_from_extrema(gext, gsc3, gsh3)
        gdp = torch.randn(B, 256, gm, device=device)
        _, _, gcoef3 = K.bn_relu_pool_backward_stats(None, gdp, gamax, gymax, gq3[0], gsc3, gsh3, gmean3, ginv3,
                                                     True, ns=gns)
        us = time_op(lambda: K.pool_gram_backward(gw3, gy2, gc2, gq2[0], gcoef3, (gmean3, ginv3, gsc3, gsh3), gdp,
                                                  gamax, gymax, gns, True), iters=5, warm=2)
        gflops = 2.0 * gcols * (128 * 128 + 128 * 256 + 256 * 128 + 10.0 / 16 * 128 * 128)
        t["mlp_gram_bwd_sa2_256x128"] = {
            "us": round(us, 2), "flops": int(gflops), "bytes": int(4.0 * gcols * 128 * 3),
            "bound": "mfma (192 steps x 6 per 32-column chunk over two passes; floor 134 us at 2.1 GHz)",
            "TFLOPs": round(gflops / us * 1e-6, 1),
            "frac_of_mfma_floor": round(134.0 / us, 3),
            "note": "both passes + prep / pack / reduce / dw; replaces the stored-y3 backward (308 us) and "
                    "the layer below's 47-us sums pass"}
        del gy2, gext
    # SA2's first layer applied before the gather (csrc/mlp_pregather.hip): gather of the small
    # GEMM's output with the BatchNorm moments / its backward (BN+ReLU backward on the fly, scatter
    # through the inverse index), at N = 2048 -> m = 1024 x ns = 32, 128 channels
    n2, m2_, ns2_, c2 = 2048, 1024, 32, 128
    z_ext = torch.randn(B, c2, n2 + m2_, device=device)
    pidx = torch.randint(0, n2, (B, m2_, ns2_), dtype=torch.int32, device=device)
    pinv = ext.group_inverse(pidx, n2)
    gamma = torch.rand(c2, device=device) + 0.5
    beta, rmean, rvar = torch.zeros(c2, device=device), torch.zeros(c2, device=device), torch.ones(c2, device=device)
    us = time_op(lambda: K.pregather_forward(z_ext, pidx, n2, (gamma, beta, rmean, rvar, 0.1, 1e-5)))
    hbm("pregather_fwd_sa2", us, 4 * B * (c2 * m2_ * ns2_ + c2 * (n2 + m2_) + m2_ * ns2_))
    y1 = torch.randn(B, c2, m2_, ns2_, device=device)
    dz1 = torch.randn(B, c2, m2_, ns2_, device=device)
    mean, invstd, scale, shift = K.bn_coefficients(y1, gamma, beta, rmean, rvar, 0.1, 1e-5, True)
    _, _, coef = K.bn_relu_backward_stats(y1, dz1, gamma, scale, shift, mean, invstd, True)
    fly = (y1, dz1, scale, shift, mean, invstd, coef)
    us = time_op(lambda: K.pregather_backward(fly, pinv, n2))
    hbm("pregather_bwd_sa2", us, 4 * B * (2 * c2 * m2_ * ns2_ + c2 * (n2 + m2_)) + 4 * pinv.numel())
    return t, forms


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
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local_rank = 0  # every rank on cuda:0: exercises the N > 1 code path on a one-GPU box
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group("gloo")
        # the driver computes scaling from --gpus: the process group must be exactly that wide
        assert torch.distributed.get_world_size() == args.gpus == world, \
            "process group of %d ranks, --gpus %d, WORLD_SIZE %d" % (
                torch.distributed.get_world_size(), args.gpus, world)
    importlib.import_module("3dioumatch_amd")
    V = importlib.import_module("3dioumatch_amd.votenet")
    data = importlib.import_module("3dioumatch_amd.votenet.data")
    cfg = V.sunrgbd_config() if args.workload == "sunrgbd" else V.scannet_config()
    npts = 20000 if args.workload == "sunrgbd" else NPTS

    step = build_step(V, cfg, device, world, local_rank, args.workload)
    step.runner.time_exchange = world > 1  # events around every eager gradient all-reduce
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

    def timed_loop(step, batch, steps, warmup):
        views = [dict(batch), dict(batch)]  # two views of the resident batch: current / next
        if pipelined:
            step.prefetch(views[0])
        history = torch.zeros(warmup + steps, device=device)  # loss per step (diagnostics)
        for i in range(warmup):
            if pipelined:
                step.prefetch(views[(i + 1) % 2])
            history[i].copy_(step(views[i % 2]).detach())
        fence()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            if pipelined:
                step.prefetch(views[(i + 1) % 2])
            loss = step(views[i % 2])
            history[i].copy_(loss.detach())
        fence()
        elapsed = time.perf_counter() - t0
        if not bool(torch.isfinite(history).all()):
            raise RuntimeError("non-finite loss during the timed steps")
        return elapsed, views

    def rotating_loop(step, host_batches, steps, warmup):
        """The timed loop of a training run that FEEDS data: K distinct batches in pinned host
        memory, three device-side input sets; iteration i copies batch i+2 host -> device on a
        copy stream (its set was last read by step i-1), prefetches the index chain of batch i+1
        and runs step i.  Also returns the host time per iteration measured in a second pass with
        the device idle-waited between iterations (Python + launch time only)."""
        k = len(host_batches)
        # ONE copy stream per process: HIP maps streams onto a few hardware queues, and a later workload's
        # fresh stream was seen to share a queue with the step's own streams (the semi-supervised
        # workload fed 0.4 ms slower as the third runner of the default run than alone)
        if "stream" not in COPY_STREAM:
            COPY_STREAM["stream"] = torch.cuda.Stream(device=device)
        copy_stream = COPY_STREAM["stream"]
        main = torch.cuda.current_stream(device)
        # every batch is ONE pinned byte buffer (what a collate function that writes into a
        # preallocated pinned arena hands over) and every device-side set one byte buffer with typed
        # views into it: one host -> device copy per step instead of one per tensor
        layout, off = [], 0
        for key, v in host_batches[0].items():
            nbytes = v.numel() * v.element_size()
            layout.append((key, off, nbytes, v.dtype, tuple(v.shape)))
            off = (off + nbytes + 255) // 256 * 256
        h2d_bytes = off
        packed = []
        for hb in host_batches:
            flat = torch.empty(off, dtype=torch.uint8).pin_memory()
            for key, o, nbytes, dtype, shape in layout:
                flat[o:o + nbytes].view(dtype).view(shape).copy_(hb[key])
            packed.append(flat)
        flats = [torch.empty(off, dtype=torch.uint8, device=device) for _ in range(3)]
        sets = [{key: f[o:o + nbytes].view(dtype).view(shape) for key, o, nbytes, dtype, shape in layout}
                for f in flats]
        filled = [torch.cuda.Event() for _ in range(3)]
        consumed = [torch.cuda.Event() for _ in range(3)]

        def upload(i):  # batch i -> set i % 3, after that set's last reader
            slot = i % 3
            copy_stream.wait_event(consumed[slot])
            with torch.cuda.stream(copy_stream):
                flats[slot].copy_(packed[i % k], non_blocking=True)
                filled[slot].record(copy_stream)

        def one(i, history):
            upload(i + 2)
            if pipelined:
                main.wait_event(filled[(i + 1) % 3])
                step.prefetch(views[(i + 1) % 3])
            main.wait_event(filled[i % 3])
            loss = step(views[i % 3])
            consumed[i % 3].record(main)
            history[i].copy_(loss.detach())

        for ev in consumed:
            ev.record(main)
        views = [dict(sets[j]) for j in range(3)]
        if "supervised_mask" in host_batches[0]:
            # which samples are labeled is host-side control flow baked into the graphs: hand the
            # runner a host copy so it never reads the (rewritten) device tensor back
            masks = [tuple(int(v) for v in hb["supervised_mask"].tolist()) for hb in host_batches]
            assert all(m_ == masks[0] for m_ in masks), "the rotating batches must share one layout"
            for v in views:
                v["supervised_mask_host"] = masks[0]
        upload(0)
        upload(1)
        main.wait_event(filled[0])
        if pipelined:
            step.prefetch(views[0])
        history = torch.zeros(warmup + 2 * steps, device=device)
        for i in range(warmup):
            one(i, history)
        fence()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            one(i, history)
        fence()
        elapsed = time.perf_counter() - t0
        # host time per iteration: the same calls with the device drained between iterations, so the
        # host never waits on a full queue -- what one rank's Python + launches cost per step
        host = 0.0
        for i in range(warmup + steps, warmup + 2 * steps):
            t1 = time.perf_counter()
            one(i, history)
            host += time.perf_counter() - t1
            torch.cuda.synchronize()
        fence()
        if not bool(torch.isfinite(history).all()):
            raise RuntimeError("non-finite loss during the timed steps")
        # leave the runner as the resident loop expects it: no prefetched slot pending
        if pipelined:
            step(views[(warmup + 2 * steps) % 3])
            fence()
        return elapsed, host / steps, h2d_bytes

    rotate = max(0, args.rotate)
    elapsed_rot = host_ms = h2d_bytes = None
    if rotate > 0:
        def make(seed):
            if args.workload == "semi":
                return data.make_semi_batch(SEMI_LABELED, SEMI_UNLABELED, NPTS, cfg, seed=seed)
            return data.make_batch(scenes, npts, cfg, seed=seed)
        host_batches = [{key: v.pin_memory() for key, v in make(100 + rank * rotate + j).items()
                         if torch.is_tensor(v)} for j in range(rotate)]
        elapsed_rot, host_s, h2d_bytes = rotating_loop(step, host_batches, args.steps, args.warmup)
        host_ms = host_s * 1e3
    elapsed, views = timed_loop(step, batch, args.steps, args.warmup)
    rccl = None
    if world > 1:
        # MAX over ranks of both timed loops (and of the host time per step)
        t = torch.tensor([elapsed, elapsed_rot if elapsed_rot is not None else 0.0, host_ms or 0.0],
                         device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if elapsed_rot is not None:
            elapsed_rot, host_ms = float(t[1].item()), float(t[2].item())
        # what the N > 1 line needs to be checked from outside: the collective's backend, the
        # width of the process group, which device every rank drove, the gradient all-reduce's
        # device time (median over the warm-up + timed steps of rank 0)
        mine = torch.tensor([torch.cuda.current_device()], dtype=torch.int64,
                            device=device if args.backend == "nccl" else "cpu")
        ids = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(ids, mine)
        rccl = step.runner.exchange_report()
        rccl["ranks_device_ids"] = [int(i.item()) for i in ids]
        rccl["ranks_share_one_gpu"] = os.environ.get("BENCH_SHARE_GPU") == "1"

    # the same step with the index chain inline (not part of `value`): what the one-step-ahead
    # prefetch of the coordinate-only chain hides
    ms_inline = None
    if pipelined and world == 1:
        extra = max(2, args.steps // 2)
        step(views[(args.warmup + args.steps) % 2])  # consumes the last prefetched slot
        fence()
        t1 = time.perf_counter()
        for i in range(extra):
            step(views[i % 2])
        fence()
        ms_inline = (time.perf_counter() - t1) * 1e3 / extra

    if rank == 0:
        ms_res = elapsed * 1e3 / args.steps
        timed = elapsed_rot if elapsed_rot is not None else elapsed  # the loop `value` is quoted on
        ms = timed * 1e3 / args.steps
        out = {
            "metric": "scenes/sec train-step (ScanNet 40k pts, 256 proposals)" if args.workload != "sunrgbd"
            else "scenes/sec train-step (SUN RGB-D 20k pts, 256 proposals)",
            "value": round(scenes * world * args.steps / timed, 3), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            # the loop of rounds 1-4: ONE batch resident in HBM, replayed (no host -> device copy)
            "value_resident": round(scenes * world * args.steps / elapsed, 3),
            "ms_per_step_resident": round(ms_res, 3),
            "rotate": rotate,
            "host_ms_per_step": None if host_ms is None else round(host_ms, 3),
            "h2d_bytes_per_step": h2d_bytes,
            "timed_loop": ("%d distinct batches (seeds %d..%d) in pinned host memory, batch i+2 copied "
                           "host -> device on a copy stream inside the timed region, index chain of "
                           "batch i+1 prefetched, step i" % (rotate, 100 + rank * rotate,
                                                             100 + rank * rotate + rotate - 1))
            if rotate > 0 else "one batch resident in HBM, replayed",
            "ms_per_step_no_prefetch": None if ms_inline is None else round(ms_inline, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "products": "3xbf16 split: every fp32 product of the shared-MLP GEMMs = 6 x "
                        "v_mfma_f32_32x32x16_bf16 on an exact three-term split, fp32 accumulate",
            "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload],
                       "per_gpu_batch": scenes, "global_batch": scenes * world, "num_points": npts,
                       "num_proposals": KPROP, "parallelism": "dp%d" % world,
                       "fps_prefetch_one_step_ahead": pipelined,
                       "hip_graphs": bool(step.runner.graphs)},
        }
        if rccl is not None:
            out["rccl"] = rccl
        if not args.no_kernels:
            table, forms = kernel_table(device)
            # headline = the SLOWEST of the three clouds the kernel is quoted on (cloud U(L), cloud R,
            # the timed step's own batch); the per-cloud durations are all in `forms`
            per_cloud = {"U(L)": forms["layer"], "R": forms["layer_cloud_R"],
                         "timed step's batch": forms["layer_cloud_step"]}
            worst = max(per_cloud, key=per_cloud.get)
            layer_us = per_cloud[worst]
            achieved = PAIR_BYTES / (layer_us * 1e-6) / 1e9
            # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
            # WRITE_SIZE in separate runs, gfx950 correction of the guide applied): measured by
            # tools/pair_bench.py --plain under the profiler, NOT by this run
            traffic, source = None, None
            pmc = newest_profile("pair_pmc.json")
            if pmc:
                traffic = json.load(open(pmc)).get("traffic_bytes_fused_kernel")
                source = "profiles/%s (separate rocprofv3 --pmc passes)" % os.path.basename(pmc)
            # VALU utilisation of the VALU-bound operators, from the committed
            # counter pass of tools/op_bench.py (tools/valu_util.py), not from this run
            vu = newest_profile("ops_valu_util.json")
            if vu:
                busy = {k: v["valu_busy"] for k, v in json.load(open(vu))["kernels"].items()}
                for op_name, kern in (("iou3d_2048x512", "pair_matrix_kernel<2>"),
                                      ("three_nn_gridconv", "three_nn_kernel"),
                                      ("ball_query_sa2", "ball_query_bf_kernel<2>")):
                    if op_name in table and kern in busy:
                        table[op_name]["valu_busy"] = busy[kern]
                        table[op_name]["valu_busy_source"] = "profiles/%s (%s)" % (os.path.basename(vu), kern)
            # in-step duration of the kernels behind the table's entries, next to their standalone
            # `us` (committed rocprofv3 summary of the timed steps, tools/step_breakdown.py): a
            # kernel that behaves differently on the step's own data or beside the side stream
            # shows up here (round 3: the query kernel, 18 us standalone / 86 us in the step)
            ss = newest_profile("train_step_timed_summary.csv")
            if ss:
                import csv
                rows = {r["kernel"]: r for r in csv.DictReader(open(ss))}
                for op_name, kern in (
                        ("fps_40000_2048_with_cell_lists", "fps_bucket_rounds_kernel<8, 2, 2>"),
                        ("query_and_group_sa1_fused_kernel", "grid_query_kernel<192, 1, true, true, 1>"),
                        ("group_grad_sa2_c128", "group_points_grad_sorted_kernel<32>"),
                        ("group_inverse_sa2", "group_inverse_kernel"),
                        ("three_interpolate_gridconv", "three_interpolate_lds_kernel<8>"),
                        ("mlp_chain_fwd_sa1", "chain_lin4_kernel<4, 64, true>"),
                        ("mlp_gram_bwd_sa1_128x64", "pool_gram_bwd_kernel"),
                        ("pregather_bwd_sa2", "pregather_backward_kernel<32>")):
                    r_ = rows.get(kern)
                    if op_name in table and r_ is not None and float(r_["launches_per_step"]) > 0:
                        n_l = float(r_["launches_per_step"])  # 0.9: 18 prefetches in 20 timed steps
                        table[op_name]["in_step_us_per_launch"] = round(float(r_["us_per_step"]) / n_l, 2)
                        table[op_name]["in_step_kernel"] = "%s (%s launches per step; profiles/%s)" % (
                            kern, r_["launches_per_step"], os.path.basename(ss))
            # the same kernel inside the timed step (beside the main stream's GEMMs), from the
            # committed rocprofv3 summary of the timed steps -- not from this run
            in_step = table.get("query_and_group_sa1_fused_kernel", {}).get("in_step_us_per_launch")
            out["roofline"] = {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": source,
                "kernel": "grid_query_kernel<192,1,true,true,1>: ball_query + group_points(xyz,C=3) + "
                          "group_points(feat,C=1) in ONE launch @ B=8 N=40000 m=2048 ns=64, on the "
                          "cell lists and query plans the layer's furthest-point-sampling kernel "
                          "leaves behind; duration = the slowest of cloud U(L), cloud R and the "
                          "timed step's own batch (here: %s)" % worst,
                "algorithmic_bytes": PAIR_BYTES, "duration_us": round(layer_us, 2),
                "in_step_us": in_step,
                "in_step_frac": None if not in_step else round(PAIR_BYTES / (in_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "in_step_source": None if not ss else "profiles/%s (rocprofv3 --kernel-trace of the timed steps: "
                                  "the kernel runs on the prefetch stream beside the main stream's GEMMs)"
                                  % os.path.basename(ss),
                # how far this decomposition (one wave per centroid on cell lists) can go: a skeleton
                # of the nine row loads + the stores of the pair alone (no tests, no ranking)
                "floor": {"us": 13.0, "frac": round(PAIR_BYTES / 13.0e-6 / 1e9 / HBM_PEAK_GBS, 4),
                          "source": "tools/micro/td_rate.py (profiles/r3_vector_memory_microbench.json: "
                                    "12.8-13.3 us for 16 384 waves x 9 row loads + the pair's stores); "
                                    "round 6: 186 vector + 250 scalar instructions per centroid "
                                    "(profiles/r6_pair_sq_counters.csv; rounds 4-5: 345 + 288), texture "
                                    "data path ~70 % busy"},
                "forms": {k: {"us": round(v, 2),
                              "frac": round(PAIR_BYTES / (v * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                          for k, v in forms.items()},
                "forms_note": "layer = the kernel above, as the timed step runs SA1 (in the prefetched "
                              "index chain, votenet/backbone.py compute_geometry); self_contained = "
                              "pn2_query_and_group (two-kernel cell-list build + the kernel above); "
                              "reference_api_3_calls = ball_query (build + query), group_points, "
                              "group_points; ..._cached_lists = the same three calls when _ext "
                              "finds the cloud's lists in its cache (left by the sampling call)",
                "definition_note": "frac is the `layer` form since round 2 (round 1: the three-call "
                                   "form) and, since round 4, its minimum over the three clouds "
                                   "(forms.layer = U(L), layer_cloud_R, layer_cloud_step); "
                                   "PAIR_BYTES always counts the unfused 38.5 MB"}
            out["kernels"] = table
            out["time_op_eager_fallbacks"] = TIME_OP_EAGER
        if world == 1 and args.workload == "pretrain" and not args.no_workloads:
            # the two other single-GPU workloads of BASELINE.json, short passes OUTSIDE the headline
            # timed region (configs[2]: SUN RGB-D pretrain; configs[3]: the semi-supervised step)
            del step
            torch.cuda.empty_cache()
            out["workloads"] = {}
            for name in ("sunrgbd", "semi"):
                wcfg = V.sunrgbd_config() if name == "sunrgbd" else V.scannet_config()
                wstep = build_step(V, wcfg, device, 1, local_rank, name)
                if name == "semi":
                    wscenes = SEMI_LABELED + SEMI_UNLABELED
                    wbatch = data.make_semi_batch(SEMI_LABELED, SEMI_UNLABELED, NPTS, wcfg, seed=100,
                                                  device=device)
                else:
                    wscenes = 16
                    wbatch = data.make_batch(wscenes, 20000, wcfg, seed=100, device=device)
                wsteps = 10
                # the loop that feeds data first (as the headline's `value`), then the resident loop
                wrot = wh2d = whost = None
                if rotate > 0:
                    def wmake(seed, name=name, wcfg=wcfg, wscenes=wscenes):
                        if name == "semi":
                            return data.make_semi_batch(SEMI_LABELED, SEMI_UNLABELED, NPTS, wcfg, seed=seed)
                        return data.make_batch(wscenes, 20000, wcfg, seed=seed)
                    whb = [{key: v.pin_memory() for key, v in wmake(100 + j).items() if torch.is_tensor(v)}
                           for j in range(rotate)]
                    wrot, whost, wh2d = rotating_loop(wstep, whb, wsteps, 3)
                    del whb
                wel, _ = timed_loop(wstep, wbatch, wsteps, 3)
                wtimed = wrot if wrot is not None else wel
                out["workloads"][name] = {
                    "workload": WORKLOADS[name], "steps": wsteps, "warmup": 3,
                    "per_gpu_batch": wscenes, "ms_per_step": round(wtimed * 1e3 / wsteps, 3),
                    "value": round(wscenes * wsteps / wtimed, 3), "unit": "scenes/s",
                    "value_resident": round(wscenes * wsteps / wel, 3),
                    "ms_per_step_resident": round(wel * 1e3 / wsteps, 3),
                    "host_ms_per_step": None if whost is None else round(whost * 1e3, 3),
                    "h2d_bytes_per_step": wh2d,
                    "hip_graphs": bool(wstep.runner.graphs)}
                probe_ms = getattr(wstep.runner, "_teacher_probe_ms", None)
                if probe_ms:  # which stream replays the teacher's graph (votenet/step.py)
                    out["workloads"][name]["teacher_replay_stream_probe"] = {
                        "ms_per_pair_of_forward_graphs": [round(t, 3) for t in probe_ms],
                        "picked": wstep.runner._teacher_probe_pick}
                del wstep, wbatch
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline and args.workload == "pretrain":
            out["cpu_baseline"] = cpu_baseline(V, cfg)
        if "workloads" in out:  # (kept behind the contract's keys in the printed line)
            out["workloads"] = out.pop("workloads")
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()  # rank 0 is still timing the per-operator table
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
