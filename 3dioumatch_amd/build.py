"""Build the gfx950 shared library (C ABI, include/*.h) with hipcc -- in-tree, no torch.

    python 3dioumatch_amd/build.py [--force]

Produces 3dioumatch_amd/lib3dioumatch_hip.so.  hipcc cross-compiles for gfx950 without a GPU.
-ffp-contract=off is part of the parity contract (fp32, source order, no fused a*b+c);
-munsafe-fp-atomics selects the hardware global_atomic_add_f32 for the scatter-add kernels.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# PN2_BUILD_SUFFIX=_x: a second library lib3dioumatch_hip_x.so (objects in build_x/) next to the
# product, loaded instead of it when PN2_LIB_SUFFIX=_x (A/B of a compile-time switch inside ONE
# gpurun call: both libraries travel with the snapshot)
SUFFIX = os.environ.get("PN2_BUILD_SUFFIX", "")
OUT = os.path.join(HERE, "lib3dioumatch_hip%s.so" % SUFFIX)
OBJ = os.path.join(HERE, "build" + SUFFIX)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("PN2_EXTRA_HIPCC_FLAGS", "").split()  # A/B experiments (tools/micro/README.md)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers_mtime():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC)
               if f.endswith(".h"))


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hm = headers_mtime()
    jobs = []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        stale = force or not os.path.exists(obj) or \
            os.path.getmtime(obj) < max(os.path.getmtime(src), hm)
        jobs.append((src, obj, stale))

    def compile_one(job):
        src, obj, stale = job
        if stale:
            subprocess.check_call([HIPCC, *FLAGS, "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    relink = force or not os.path.exists(OUT) or any(j[2] for j in jobs) or \
        any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs)
    if relink:
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
        if verbose:
            print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
