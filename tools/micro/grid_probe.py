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
so = os.path.join(HERE, "libgrid_probe.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(CSRC, "pn2_ball_grid.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared",
                           "-fPIC", "-ffp-contract=off", "-DGRID_PROBE", "-I", CSRC,
                           os.path.join(HERE, "grid_probe.hip"), "-o", so])
if "--build-only" in sys.argv:
    sys.exit(0)
lib = ctypes.CDLL(so)
lib.grid_probe_ws.restype = ctypes.c_size_t
lib.grid_probe_ws.argtypes = [ctypes.c_int] * 4
lib.grid_probe_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 6 + \
    [ctypes.c_size_t, ctypes.c_void_p]
lib.grid_probe_read.argtypes = [ctypes.c_void_p]
importlib.import_module("3dioumatch_amd")
ext = importlib.import_module("pointnet2._ext")
dev = torch.device("cuda:0")
B, N, M, NS, R = 8, 40000, 2048, 64, 0.2
res = {}
for kind in ("U", "R", "step"):
    xyz = bench.pair_cloud(kind).to(dev)
    inds = ext.furthest_point_sampling(xyz, M)
    new_xyz = ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    feat = torch.rand(B, 1, N, device=dev)
    idx = torch.zeros(B, M, NS, dtype=torch.int32, device=dev)
    out = torch.zeros(B, 4, M, NS, device=dev)
    nbytes = lib.grid_probe_ws(B, N, M, NS)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    for _ in range(3):
        rc = lib.grid_probe_run(B, N, M, 1, R, NS, new_xyz.data_ptr(), xyz.data_ptr(), feat.data_ptr(),
                                idx.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    t = np.zeros((16384, 8), np.uint64)
    assert lib.grid_probe_read(t.ctypes.data) == 0
    if '--raw' in sys.argv:
        np.save(os.path.join(ROOT, 'gpurun_out', 'r4a', 'grid_probe_raw_%s.npy' % kind), t)
    t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
    sweeps, chunks, hits = (t[:, 2] >> np.uint64(32)).astype(np.int64), (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64), t[:, 3].astype(np.int64)
    CLK = 2300.0  # s_memtime ticks per us as measured in profiles/r3_instruction_costs.json (per XCD, unsynchronised)
    dur = (t1 - t0) / CLK
    base = np.repeat(np.array([t0[c * M:(c + 1) * M].min() for c in range(B)]), M)  # one cloud = one XCD
    start = (t0 - base) / CLK
    end = (t1 - base) / CLK
    gen = sweeps > 0
    r = {"span_us_per_cloud": [round(float(end[c * M:(c + 1) * M].max()), 2) for c in range(B)],
         "general_per_cloud": [int(gen[c * M:(c + 1) * M].sum()) for c in range(B)],
         "last_start_us_per_cloud": [round(float(start[c * M:(c + 1) * M].max()), 2) for c in range(B)],
         "mean_concurrency_per_cloud": [round(float(dur[c * M:(c + 1) * M].sum() / end[c * M:(c + 1) * M].max()), 1) for c in range(B)],
         "fast": {"count": int((~gen).sum()), "dur_us_mean": round(float(dur[~gen].mean()), 2),
                  "dur_us_p99": round(float(np.percentile(dur[~gen], 99)), 2)}}
    if gen.any():
        r["general"] = {"count": int(gen.sum()), "dur_us_mean": round(float(dur[gen].mean()), 2),
                        "dur_us_p99": round(float(np.percentile(dur[gen], 99)), 2),
                        "dur_us_max": round(float(dur[gen].max()), 2),
                        "chunks_mean": round(float(chunks[gen].mean()), 1), "chunks_max": int(chunks.max()),
                        "two_or_more_sweeps": int((sweeps > 1).sum()),
                        "dur_us_multi_sweep_mean": round(float(dur[sweeps > 1].mean()), 2) if (sweeps > 1).any() else None}
    worst = int(np.argmax(r["span_us_per_cloud"]))
    sl = slice(worst * M, (worst + 1) * M)
    order = np.argsort(end[sl])[-6:] + worst * M
    r["last_finishers_of_slowest_cloud"] = [
        {"start_us": round(float(start[i]), 2), "end_us": round(float(end[i]), 2), "general": bool(gen[i]),
         "chunks": int(chunks[i]), "sweeps": int(sweeps[i]), "hits": int(hits[i])} for i in order]
    # waves in flight over time for the slowest cloud (1 us bins)
    tl = np.arange(0, end[sl].max() + 1, 1.0)
    r["in_flight_slowest_cloud_per_us"] = [int(((start[sl] <= x) & (end[sl] > x)).sum()) for x in tl]
    res[kind] = r
print(json.dumps(res))
if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
