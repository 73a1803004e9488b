"""Build and run tools/micro/td_rate.hip: time of 16 384 waves x 9 (and 18, 36) 64-lane row loads
of 16 / 12 / 8 / 4 bytes per lane from L2-resident arrays (one 640 KB array per XCD-sized cloud).

    python tools/micro/td_rate.py            # on the GPU box (compiles with hipcc there or here)
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

so = os.path.join(HERE, "libtd_rate.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "td_rate.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "td_rate.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.td_rate_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, M = 8, 2048
cloud_bytes = 40000 * 16
data = torch.rand(B * cloud_bytes // 4 + 4096, device=dev)
out = torch.zeros(B * M * 64 * 5, device=dev)
res = {}
for rows in (9, 18, 36):
    for nbytes in (16, 12, 8, 4):
        def run():
            rc = lib.td_rate_launch(nbytes, data.data_ptr(), cloud_bytes, B, M, rows, out.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        us = bench.time_op(run, iters=20, warm=3)
        res["rows%d_bytes%d_us" % (rows, nbytes)] = round(us, 2)
for tag, code in (("five_dword_stores", 165), ("one_dwordx4_plus_one_dword_store", 162),
                  ("five_dword_stores_4_waves_per_workgroup", 1654),
                  ("five_dword_stores_16_waves_per_workgroup", 16516)):
    def run():
        rc = lib.td_rate_launch(code, data.data_ptr(), cloud_bytes, B, M, 9, out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    res["rows9_bytes16_%s_us" % tag] = round(bench.time_op(run, iters=20, warm=3), 2)
    if code >= 1654:  # the fixed part alone: no row loads
        def run0():
            rc = lib.td_rate_launch(code, data.data_ptr(), cloud_bytes, B, M, 0, out.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        res["rows0_%s_us" % tag] = round(bench.time_op(run0, iters=20, warm=3), 2)
for tag, code in (("rows9_bytes16_lds_direct", 1600), ("rows9_bytes16_lds_direct_five_dword_stores", 1605)):
    def run():
        rc = lib.td_rate_launch(code, data.data_ptr(), cloud_bytes, B, M, 9, out.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    res["%s_us" % tag] = round(bench.time_op(run, iters=20, warm=3), 2)


def run00():
    rc = lib.td_rate_launch(165, data.data_ptr(), cloud_bytes, B, M, 0, out.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0
res["rows0_five_dword_stores_us"] = round(bench.time_op(run00, iters=20, warm=3), 2)
print(json.dumps(res))
