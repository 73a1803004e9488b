"""Sustained fp32 MFMA rate of this box (tools/micro/mfma_peak.hip): MFMA-only waves at 1, 2 and 4
waves per SIMD, short (GEMM-launch sized, ~100 us) and long (thermal steady state) runs.

    python tools/micro/mfma_peak.py
"""
import ctypes
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

so = os.path.join(HERE, "libmfma_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "mfma_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_peak_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
inp = torch.rand(8192, device=dev)
out = torch.zeros(8192, device=dev)
NAMES = {0: "acc4", 1: "acc4_2_lds_reads", 2: "acc4_4_lds_reads", 3: "acc8", 4: "acc2", 5: "acc1"}
ACC = {0: 4, 1: 4, 2: 4, 3: 8, 4: 2, 5: 1}
res = {}
for variant in (0, 1, 2, 3, 4, 5):
    for wps in (1, 2, 4):          # waves per SIMD: blocks of 4 waves, wps blocks per CU
        blocks = 256 * wps
        for iters in (2000, 40000):
            n = iters // wps

            def run():
                rc = lib.mfma_peak_launch(variant, blocks, n, out.data_ptr(), inp.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
                assert rc == 0
            us = bench.time_op(run, iters=10 if iters < 10000 else 3, warm=2)
            flops = blocks * 4 * n * ACC[variant] * 4096.0
            res["%s_wps%d_%s" % (NAMES[variant], wps, "short" if iters < 10000 else "long")] = {
                "us": round(us, 1), "TFs": round(flops / us / 1e6, 1)}
print(json.dumps(res))
