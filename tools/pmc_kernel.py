"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV with the derived occupancies of
the SQ counters (MI355X_MICROARCH.md: SQ_BUSY_CYCLES summed over 32 shader engines,
SQ_VALU_MFMA_BUSY_CYCLES cycles summed over 1024 SIMDs, SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES
quad-cycles summed over waves).
    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
        SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY \
        --output-format csv -d out -o p -- python tools/chain_bench.py
    python tools/pmc_kernel.py out/p_counter_collection.csv [name-filter] [out.json]"""
import csv, json, re, sys
from collections import defaultdict
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if flt not in k: continue
    short = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))
    a = agg[short][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, d in agg.items():
    m = {c: v[1] / v[0] for c, v in d.items()}
    busy = m.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    e = {"launches": int(max(v[0] for v in d.values())), "busy_cycles": round(busy)}
    if busy:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: e["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / busy, 3)
        if "SQ_ACTIVE_INST_VALU" in m: e["valu_busy"] = round(m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / busy, 3)
        if "SQ_INSTS_VALU" in m: e["valu_insts_per_simd"] = round(m["SQ_INSTS_VALU"] / 1024)
        if "SQ_INSTS_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m["SQ_INSTS_VALU"]:
            e["cycles_per_valu_inst"] = round(m["SQ_ACTIVE_INST_VALU"] * 4 / m["SQ_INSTS_VALU"], 2)
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS"):
            if c in m: e[c.lower() + "_of_wave_cycles"] = round(m[c] / wc, 3)
    for c in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        if c in m: e[c.lower()] = round(m[c])
    out[k] = e
print(json.dumps(out, indent=1))
if len(sys.argv) > 3: json.dump(out, open(sys.argv[3], "w"), indent=1)
