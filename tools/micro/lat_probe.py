"""Instruction / construct costs in shader clocks on this box (tools/micro/lat_probe.hip), for one
workgroup of 4 / 8 / 16 waves (1 / 2 / 4 per SIMD).

    python tools/micro/lat_probe.py
"""
import ctypes
import json
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "liblat_probe.so")
src = os.path.join(HERE, "lat_probe.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so])
if __name__ == "__main__":
    lib = ctypes.CDLL(so)
    lib.lat_probe_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4
    dev = torch.device("cuda:0")
    inp = torch.rand(4096, device=dev) + 1.0
    # pointer chase: a cycle through 64 slots 4 KiB apart (1024 ints)
    chase = torch.zeros(64 * 1024 + 64, dtype=torch.int32, device=dev)
    order = torch.randperm(64)
    for i in range(64):
        chase[int(order[i]) * 1024] = int(order[(i + 1) % 64]) * 1024
    # the chain starts at slot 0
    NAMES = ["stamp_pair", "v_add_f32_dependent", "v_add_f32_4_streams", "v_max_f32_dpp_dependent",
             "v_max_f32_dpp_4_streams", "v_permlane32_swap_dependent", "v_permlane16_swap_dependent",
             "v_readlane_const", "valu_sgpr_valu_roundtrip", "cmp_ff1_readlane_add_chain", "branch_taken",
             "branch_not_taken", "s_ff1_bitset0_chain", "lds_write_read_roundtrip", "s_barrier",
             "global_load_chain_L2", "global_load_chain_again", "v_cndmask_dependent", "v_mul_add_dependent_per_instr",
             "cmp_scalar_test_branch", "global_store_issue", "s_memtime_wait", "gpr_idx_mov", "lds_read_latency"]
    res = {}
    for waves in (1, 4, 8, 16):
        out = torch.zeros(4096, device=dev)
        for _ in range(3):
            rc = lib.lat_probe_launch(waves, out.data_ptr(), inp.data_ptr(), chase.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
        v = out[:len(NAMES)].cpu().tolist()
        res["waves_%d" % waves] = {n: round(x, 1) for n, x in zip(NAMES, v)}
    print(json.dumps(res, indent=1))
