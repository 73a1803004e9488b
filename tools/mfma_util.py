"""MFMA utilisation per GEMM kernel from a rocprofv3 --pmc counter_collection CSV.

    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
        SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
        --output-format csv -d out -o g -- python tools/gemm_bench.py
    python tools/mfma_util.py out/g_counter_collection.csv profiles/r1_gemm_mfma_util.csv

SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs:
utilisation = (MFMA_BUSY / 1024) / (BUSY / 32).
"""
import csv
import re
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "gemm_" not in k:
        continue
    mt = re.search(r"gemm_\w+<[^>]*>", k)
    short = mt.group(0) if mt else k[:60]
    a = agg[short][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
rows = []
for k, d in agg.items():
    m = {c: v[1] / v[0] for c, v in d.items()}
    busy = m.get("SQ_BUSY_CYCLES", 0) / 32
    util = (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024) / busy if busy else 0
    wait = m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else 0
    rows.append((k, int(d["SQ_BUSY_CYCLES"][0]), busy, util, m.get("SQ_INSTS_VALU", 0),
                 m.get("SQ_INSTS_LDS", 0), m.get("SQ_LDS_BANK_CONFLICT", 0), wait))
rows.sort(key=lambda r: -r[2] * r[1])
with open(sys.argv[2], "w") as f:
    f.write("kernel,launches,busy_cycles_per_SE,mfma_utilisation,valu_insts,lds_insts,"
            "lds_bank_conflict_cycles,wait_inst_fraction_of_wave_cycles\n")
    for r in rows:
        f.write('"%s",%d,%.0f,%.3f,%.0f,%.0f,%.0f,%.3f\n' % r)
print(open(sys.argv[2]).read())
