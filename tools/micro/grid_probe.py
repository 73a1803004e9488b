"""Per-centroid clocks of the cell-list query kernel (csrc/pn2_ball_grid.hip built with
-DGRID_PROBE: s_memtime at the start / end of every centroid's wave, the path it took, the
number of 64-record chunks and sweeps of the general path, its hit count) on the three clouds the
north-star pair is quoted on.

    python tools/micro/grid_probe.py [out.json]          # on the GPU box
"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import importlib  # noqa: E402

CSRC = os.path.join(ROOT, "3dioumatch_amd", "csrc")
# GRID_PROBE_FLAGS="-DGRID_CPW=2 ...": the probed build of a kernel variant (GRID_PROBE_TAG names its library)
so = os.path.join(HERE, "libgrid_probe%s.so" % os.environ.get("GRID_PROBE_TAG", ""))
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(CSRC, "pn2_ball_grid.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared",
                           "-fPIC", "-ffp-contract=off", "-DGRID_PROBE", "-I", CSRC,
                           *os.environ.get("GRID_PROBE_FLAGS", "").split(),
                           os.path.join(HERE, "grid_probe.hip"), "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.grid_probe_ws.restype = ctypes.c_size_t
lib.grid_probe_ws.argtypes = [ctypes.c_int] * 4
lib.grid_probe_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 6 + \
    [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
lib.grid_probe_read.argtypes = [ctypes.c_void_p]
importlib.import_module("3dioumatch_amd")
ext = importlib.import_module("pointnet2._ext")
dev = torch.device("cuda:0")
B, N, M, NS, R = 8, 40000, 2048, 64, 0.2
res = {}
for kind in ("U", "R", "step"):
    xyz = bench.pair_cloud(kind).to(dev)
    # --layer: on the cell lists, launch order and query plans the layer's own sampling call leaves
    layer = "--layer" in sys.argv
    if layer:
        inds, lists = ext.furthest_point_sampling_with_grid(xyz, M, R)
    else:
        inds = ext.furthest_point_sampling(xyz, M)
    new_xyz = ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    feat = torch.rand(B, 1, N, device=dev)
    idx = torch.zeros(B, M, NS, dtype=torch.int32, device=dev)
    out = torch.zeros(B, 4, M, NS, device=dev)
    nbytes = lib.grid_probe_ws(B, N, M, NS)
    ws = lists.buf if layer else torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    assert ws.numel() >= nbytes
    for _ in range(3):
        rc = lib.grid_probe_run(B, N, M, 1, R, NS, new_xyz.data_ptr(), xyz.data_ptr(), feat.data_ptr(),
                                idx.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), None,
                                2 if layer else 0)
        assert rc == 0, rc
        torch.cuda.synchronize()
    t = np.zeros((16384, 8), np.uint64)
    assert lib.grid_probe_read(t.ctypes.data) == 0
    if '--raw' in sys.argv:
        os.makedirs(os.path.join(ROOT, 'gpurun_out', 'r4a'), exist_ok=True)
        np.save(os.path.join(ROOT, 'gpurun_out', 'r4a', 'grid_probe_raw_%s.npy' % kind), t)
    t = t.astype(np.int64)
    CLK = 2300.0  # s_memtime ticks per us (profiles/r3_instruction_costs.json); the counter's base
    #               differs from CU to CU, so only differences inside one wave / one CU are used
    sweeps, chunks, hits = t[:, 2] >> 32, t[:, 2] & 0xffffff, t[:, 3] & 0xffff
    path = (t[:, 2] >> 24) & 0xff
    d = lambda x, y: (t[:, x] - t[:, y]) / CLK  # noqa: E731
    total, s_start, s_sweep, s_rank, s_out = d(1, 0), d(4, 0), d(5, 4), d(6, 5), d(1, 6)
    r = {"classes": {}}
    for name, msk in (("plan_one_pass", path == 1), ("plan_two_passes", path == 2),
                      ("computed_single_load", path == 3), ("computed_seam", path == 4),
                      ("index_order_scan", path == 5),
                      ("fast", sweeps == 0), ("general_1_8_chunks", (sweeps > 0) & (chunks <= 8)),
                      ("general_9_16_chunks", (sweeps > 0) & (chunks > 8) & (chunks <= 16)),
                      ("general_17_24_chunks", (sweeps > 0) & (chunks > 16) & (chunks <= 24)),
                      ("general_over_24_chunks", (sweeps > 0) & (chunks > 24))):
        if msk.any():
            r["classes"][name] = {
                "centroids": int(msk.sum()), "wave_us_mean": round(float(total[msk].mean()), 2),
                "wave_us_max": round(float(total[msk].max()), 2),
                "centre_and_row_starts_us": round(float(s_start[msk].mean()), 2),
                "sweeps_us": round(float(s_sweep[msk].mean()), 2),
                "ranking_us": round(float(s_rank[msk].mean()), 2),
                "gather_and_stores_us": round(float(s_out[msk].mean()), 2),
                "hits_mean": round(float(hits[msk].mean()), 1),
                "more_than_one_sweep": int((sweeps[msk] > 1).sum())}
    r["stage_note"] = ("centre_and_row_starts_us of the fast class also holds its nine row loads and "
                       "tests; sweeps_us is the general path's loads + tests + list")
    # activity span per time base (one CU, sometimes a few with the same base): first wave start
    # to last wave end
    order = np.argsort(t[:, 0])
    gaps = np.nonzero(np.diff(t[order, 0]) > 3 * CLK)[0]
    spans = [(t[g, 1].max() - t[g, 0].min()) / CLK for g in np.split(order, gaps + 1) if 32 <= len(g) <= 135]  # one or two CUs
    r["activity_span_us_per_cu_burst"] = {"bursts": len(spans), "median": round(float(np.median(spans)), 2),
                                          "p90": round(float(np.percentile(spans, 90)), 2),
                                          "max": round(float(np.max(spans)), 2)}
    res[kind] = r
print(json.dumps(res))
if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
