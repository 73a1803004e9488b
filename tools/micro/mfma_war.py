"""tools/micro/mfma_war.hip: is a write to an MFMA's SrcB one instruction after its issue safe on
gfx950, also when the MFMA queues behind 1 / 3 / 7 others?    python tools/micro/mfma_war.py"""
import ctypes, json, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libmfma_war.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "mfma_war.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "mfma_war.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.mfma_war_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
# bf16 bit patterns of small integers -4..4 (exact products and sums)
vals = torch.randint(-4, 5, (1024,), generator=g).float()
bits = (vals.view(torch.int32) >> 16).to(torch.int32).to(dev)
res = {}
for nq in (1, 2, 4, 8):
    ref = None
    for dist in (3, 2, 1):   # 3 = the write 16+ cycles behind the MFMA: the reference
        out = torch.zeros(256 * 64, device=dev)
        rc = lib.mfma_war_launch(nq, dist, bits.data_ptr(), out.data_ptr(), 0x7fc07fc0, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        res["queued_%d_dist_%d" % (nq - 1, dist)] = {"equal_to_late_write": bool(torch.equal(out, ref)),
                                                      "mismatching_lanes": int((out != ref).sum())}
print(json.dumps(res, indent=1))
