"""LDS read throughput per CU by instruction (tools/micro/lds_read_rate.hip): the transposing 8-byte
fragment read against plain 8- and 16-byte reads.    python tools/micro/lds_read_rate.py"""
import ctypes, json, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import bench  # noqa: E402
so = os.path.join(HERE, "liblds_read_rate.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "lds_read_rate.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.lds_read_rate_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
out = torch.zeros(8192, device="cuda:0")
res = {}
for kind, name, width in ((0, "ds_read_b64_tr_b16", 8), (1, "ds_read_b64", 8), (2, "ds_read_b128", 16)):
    iters = 4000
    def run():
        assert lib.lds_read_rate_launch(kind, 256, iters, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    us = bench.time_op(run, iters=5, warm=2)
    per_cu = 512 * iters * 8 * width / (us * 1e-6)  # bytes per second per CU
    res[name] = {"us": round(us, 1), "GBps_per_CU": round(per_cu / 1e9, 1), "bytes_per_clk_at_2.1GHz": round(per_cu / 2.1e9, 1)}
print(json.dumps(res, indent=1))
