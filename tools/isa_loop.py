"""Instruction mix of every kernel of one HIP source, whole body and innermost hot loop (the
largest backward-branch region): MFMA / VALU / LDS / VMEM / SALU / waits.  No GPU needed.
    python tools/isa_loop.py 3dioumatch_amd/csrc/mlp_chain.hip [name-filter]"""
import collections, os, re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/isa_loop.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                       "-ffp-contract=off", "-munsafe-fp-atomics", "-fvisibility=hidden",
                       "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src],
                      stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if not m: i += 1; continue
    name = m.group(1); j = i + 1
    while j < len(lines) and "s_endpgm" not in lines[j]: j += 1
    body = lines[i:j]
    if flt in name:
        labels = {}
        for k, l in enumerate(body):
            mm = re.match(r"^(\.LBB\w+):", l)
            if mm: labels[mm.group(1)] = k
        best = (0, 0, 0)
        for k, l in enumerate(body):
            mm = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", l) or re.search(r"s_branch\s+(\.LBB\w+)", l)
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                if k - labels[mm.group(1)] > best[0]: best = (k - labels[mm.group(1)], labels[mm.group(1)], k)
        def mix(seg):
            c = collections.Counter(); ops = collections.Counter()
            for l in seg:
                l = l.strip()
                if not l or l.startswith((".", ";")) or l.endswith(":"): continue
                op = l.split()[0]; c[cls(op)] += 1
                if cls(op) in ("valu", "lds", "vmem"): ops[op] += 1
            return dict(c), ops.most_common(16)
        print(name[:90])
        print("  whole:", mix(body)[0])
        if best[0]:
            c, ops = mix(body[best[1]:best[2] + 1])
            print("  loop :", c); print("        ", ops)
    i = j
