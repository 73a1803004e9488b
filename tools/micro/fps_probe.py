"""Where a round of the bucketed FPS kernel spends its time: build pn2_fps_bucket.hip with
-DFPS_PROBE (per-round s_memtime stamps and visited-bucket counts of cloud 0), run it on the
bench scene and on a uniform cloud, print a summary.

    python tools/micro/fps_probe.py          # on the GPU box
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

CSRC = os.path.join(ROOT, "3dioumatch_amd", "csrc")
src = os.path.join(HERE, "fps_probe.hip")


CONFIGS = ((8, 2), (4, 2), (4, 4))


def load(waves, group=4):
    so = os.path.join(HERE, "libfps_probe_w%d_g%d.so" % (waves, group))
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(CSRC, "pn2_fps_bucket.hip")):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared",
                               "-fPIC", "-ffp-contract=off", "-DFPS_PROBE", "-DFPS_BUCKET_WAVES=%d" % waves, "-DFPS_BUCKET_GROUP=%d" % group,
                               "-I", CSRC, src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.fps_probe_run.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    lib.fps_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.fps_probe_scratch.restype = ctypes.c_size_t
    lib.fps_probe_scratch.argtypes = [ctypes.c_int, ctypes.c_int]
    return lib


if "--build-only" in sys.argv:
    for wv, gg in CONFIGS:
        load(wv, gg)
    sys.exit(0)

dev = torch.device("cuda:0")
import importlib
amd = importlib.import_module("3dioumatch_amd")
from importlib import import_module
data_mod = import_module("3dioumatch_amd.votenet.data")
cfg_mod = import_module("3dioumatch_amd.votenet.config")
B, N, M = 8, 40000, 2048
cfg = cfg_mod.scannet_config()
scene = data_mod.make_batch(B, N, cfg, seed=100)["point_clouds"][:, :, :3].contiguous().to(dev)
torch.manual_seed(0)
uniform = (torch.rand(B, N, 3, device=dev) * 6.0).contiguous()
res = {}
for WV, GG, tag, xyz in [(wv, gg, "scene", scene) for wv, gg in CONFIGS] + [(8, 2, "uniform", uniform)]:
    lib = load(WV, GG)
    nbytes = lib.fps_probe_scratch(B, N)
    scratch = torch.empty(nbytes // 4 + 16, device=dev)
    idx = torch.empty(B, M, dtype=torch.int32, device=dev)

    def run():
        rc = lib.fps_probe_run(B, N, M, int(np.floor(np.log2(N))) if N < 512 else 9, xyz.data_ptr(),
                               scratch.data_ptr(), idx.data_ptr(), nbytes,
                               torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    us = bench.time_op(run, iters=3, warm=1)
    t = np.zeros(2048 * 16 * 16, np.uint64)
    v = np.zeros(2048 * 16, np.uint8)
    torch.cuda.synchronize()
    lib.fps_probe_read(t.ctypes.data, v.ctypes.data)
    t = t.reshape(2048, 16, 16).astype(np.int64)[:, :WV]
    v = v.reshape(2048, 16).astype(np.int64)[:, :WV]
    r0, r1 = 64, M - 1          # steady rounds
    T = t[r0:r1]
    nxt = t[r0 + 1:r1 + 1, :, 0]
    vis = v[r0:r1]
    has = vis > 0
    out = {"us_per_call": round(us, 1), "us_per_round": round(us / (M - 1), 3),
           "clk_per_round": round(float((nxt - T[:, :, 0]).mean()), 1),
           "visits_per_wave_mean": round(float(vis.mean()), 3),
           "visits_max_over_waves_mean": round(float(vis.max(axis=1).mean()), 3)}
    seg = {"stamp_cost(0-1)": (0, 1), "box_tests(1-2)": (1, 2), "visits_all(2-7)": (2, 7),
           "sets+wave_max(7-8)": (7, 8), "select_bucket(8-9)": (8, 9), "slot+barrier(9-10)": (9, 10),
           "collect(10-11)": (10, 11), "far_pt_read+barrier(11-12)": (11, 12)}
    out["clk"] = {k: round(float((T[:, :, b] - T[:, :, a]).mean()), 1) for k, (a, b) in seg.items()}
    out["clk"]["loop_back(12-next0)"] = round(float((nxt - T[:, :, 12]).mean()), 1)
    out["clk"]["barrier_min_over_waves(9-10)"] = round(float((T[:, :, 10] - T[:, :, 9]).min(axis=1).mean()), 1)
    first = {"walk+issue_loads(2-3)": (2, 3), "wait_load+dist+store(3-4)": (3, 4), "maxima+ties(4-5)": (4, 5),
             "far_pt+bval(5-6)": (5, 6)}
    out["first_visit_clk"] = {k: round(float((T[:, :, b] - T[:, :, a])[has].mean()), 1) for k, (a, b) in first.items()}
    by = {}
    for k in range(0, 10):
        sel = vis == k
        if sel.sum() > 20:
            by[str(k)] = round(float((T[:, :, 7] - T[:, :, 2])[sel].mean()), 1)
    out["visits_clk_by_count"] = by
    early = (t[2:33, :, 0][1:] - t[2:33, :, 0][:-1]).mean()
    out["clk_per_round_first_32"] = round(float(early), 1)
    out["indices_checksum"] = int(idx.long().sum().item())
    res["%s_w%d_g%d" % (tag, WV, GG)] = out
print(json.dumps(res))
