"""Static look at the gfx950 code of one HIP source: registers / LDS / occupancy per kernel
(hipcc -Rpass-analysis=kernel-resource-usage) and the instruction mix of each kernel's ISA
(VALU / SALU / VMEM / SMEM / LDS / MFMA counts from the -save-temps assembly).  Runs on a machine
without a GPU.

    python tools/isa_stats.py 3dioumatch_amd/csrc/pn2_ball_grid.hip [name-filter]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics"]


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_branch", "s_cbranch")):
        return "ctl"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    src = os.path.abspath(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        out = subprocess.run([HIPCC, *FLAGS, "-c", src, "-o", os.path.join(tmp, "a.o"),
                              "-save-temps", "-Rpass-analysis=kernel-resource-usage"],
                             cwd=tmp, stderr=subprocess.PIPE, text=True)
        if out.returncode != 0:
            sys.stderr.write(out.stderr)
            sys.exit(1)
        usage = collections.OrderedDict()
        cur = None
        for line in out.stderr.splitlines():
            m = re.search(r"remark: [^:]*:\d+:\d+: +(\S.*?) \[-Rpass", line) or \
                re.search(r"remark: +(\S.*?) \[-Rpass", line)
            body = line.split("remark:")[-1].strip() if "remark:" in line else ""
            body = re.sub(r"^\S+:\d+:\d+:\s*", "", body)
            body = body.replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
            if body.startswith("Function Name:") or body.startswith("Name:"):
                cur = body.split(":", 1)[1].strip()
                usage[cur] = {}
            elif cur and ":" in body:
                k, v = body.split(":", 1)
                usage[cur][k.strip()] = v.strip()
        asm = [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f]
        mix = {}
        if asm:
            name = None
            for line in open(os.path.join(tmp, asm[0])):
                m = re.match(r"^(\w+):\s*(;.*)?$", line)
                if m and m.group(1).startswith("_Z"):
                    name = m.group(1)
                    mix[name] = collections.Counter()
                    continue
                if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                    name = None
                if name:
                    t = line.strip().split()
                    if t and re.match(r"^[a-z_0-9]+$", t[0]) and not t[0].startswith("."):
                        mix[name][classify(t[0])] += 1
        for k, u in usage.items():
            if flt and flt not in k:
                continue
            dem = subprocess.run(["c++filt", k], stdout=subprocess.PIPE, text=True).stdout.strip()
            dem = re.sub(r"\(anonymous namespace\)::", "", dem)
            if os.environ.get("ISA_SHORT"):
                print("%-60s V %3s A %3s S %3s LDS %6s occ %s scr %s" % (
                    dem.split("(")[0][-60:], u.get("VGPRs", "?"), u.get("AGPRs", "?"),
                    u.get("TotalSGPRs", u.get("SGPRs", "?")), u.get("LDS Size [bytes/block]", "?"),
                    u.get("Occupancy [waves/SIMD]", "?"), u.get("ScratchSize [bytes/lane]", "?")))
                continue
            print(dem[:150])
            print("   VGPR %s AGPR %s SGPR %s  LDS %s  occupancy %s  scratch %s" % (
                u.get("VGPRs", "?"), u.get("AGPRs", "?"), u.get("TotalSGPRs", u.get("SGPRs", "?")),
                u.get("LDS Size [bytes/block]", "?"), u.get("Occupancy [waves/SIMD]", "?"),
                u.get("ScratchSize [bytes/lane]", "?")))
            if k in mix:
                print("   static ISA mix:", dict(mix[k]))


if __name__ == "__main__":
    main()
