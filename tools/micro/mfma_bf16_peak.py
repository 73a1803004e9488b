"""Sustained rate of v_mfma_f32_32x32x16_bf16 on this box (tools/micro/mfma_bf16_peak.hip): one to four
independent accumulators per wave, 1 / 2 waves per SIMD, and with plain vector instructions between the
MFMAs (how much vector issue fits under an MFMA stream).

    python tools/micro/mfma_bf16_peak.py [out.json]
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

so = os.path.join(HERE, "libmfma_bf16_peak.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "mfma_bf16_peak.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_bf16_peak_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
inp = torch.rand(8192, device=dev)
out = torch.zeros(8192, device=dev)
VAR = {0: (1, 0), 1: (2, 0), 2: (4, 0), 3: (1, 2), 4: (1, 4), 5: (1, 6), 6: (1, 8), 7: (2, 4), 8: (2, 6), 9: (4, 6)}
res = {}
for variant, (acc, vper) in VAR.items():
    for wps in (1, 2):          # waves per SIMD: blocks of 4 waves, wps blocks per CU
        blocks = 256 * wps
        n = 8000 // wps

        def run():
            rc = lib.mfma_bf16_peak_launch(variant, blocks, n, out.data_ptr(), inp.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        us = bench.time_op(run, iters=5, warm=2)
        flops = blocks * 4 * n * acc * 32768.0
        # cycles per MFMA and SIMD at 2.4 GHz would be 32 at the 2.5 PFLOP/s peak
        res["acc%d_valu%d_wps%d" % (acc, vper, wps)] = {"us": round(us, 1), "TFs": round(flops / us / 1e6, 1)}
for variant, name in ((10, "x6_chains_of_six"), (11, "x6_two_accumulators_alternating")):
    for wps in (1, 2):
        blocks = 256 * wps
        n = 2000 // wps

        def run():
            rc = lib.mfma_bf16_peak_launch(variant, blocks, n, out.data_ptr(), inp.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        us = bench.time_op(run, iters=5, warm=2)
        res["%s_wps%d" % (name, wps)] = {"us": round(us, 1), "TFs": round(blocks * 4 * n * 24 * 32768.0 / us / 1e6, 1)}
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
