"""Instruction-class sequence of one kernel's gfx950 ISA (M mfma, r/w LDS read/write, G/S global
load/store, W s_waitcnt, B barrier, v/s other vector/scalar):  python tools/isa_seq.py file.hip filter"""
import os, re, subprocess, sys, tempfile
src, flt = os.path.abspath(sys.argv[1]), sys.argv[2]
with tempfile.TemporaryDirectory() as tmp:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-ffp-contract=off", "-munsafe-fp-atomics", "-c", src, "-o", "a.o", "-save-temps"],
                          cwd=tmp, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp, [f for f in os.listdir(tmp) if f.endswith(".s") and "gfx950" in f][0])).read()
names = [n for n in re.findall(r"^(_Z\S+):", asm, re.M) if flt in n]
for name in names:
    i = asm.index(name + ":"); j = asm.index("s_endpgm", i)
    ops = [l.strip().split()[0] for l in asm[i:j].split("\n")
           if l.strip() and not l.strip().startswith((".", ";")) and not l.strip().endswith(":")]
    def cls(o):
        for p, c in (("v_mfma", "M"), ("ds_read", "r"), ("ds_load", "r"), ("ds_write", "w"), ("ds_store", "w"),
                     ("s_waitcnt", "W"), ("global_load", "G"), ("buffer_load", "G"), ("global_store", "S"),
                     ("s_barrier", "B"), ("v_", "v"), ("s_", "s")):
            if o.startswith(p): return c
        return "?"
    print(name[:90]); print("".join(cls(o) for o in ops))
