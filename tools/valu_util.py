"""VALU utilisation of the VALU-bound operators (pairs/s AND VALU utilisation
for the IoU kernels, three_nn, the brute-force ball query) from a rocprofv3 --pmc pass:

    rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES \
        --output-format csv -d out -o ops -- python tools/op_bench.py 3
    python tools/valu_util.py out/ops_counter_collection.csv profiles/r3_ops_valu_util.json

SQ_ACTIVE_INST_VALU counts quad-cycles (4 clocks) summed over waves, SQ_BUSY_CYCLES clocks summed
over the 32 shader engines; a SIMD issues one VALU instruction at a time, so
    valu_busy = (SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32)
is the fraction of the kernel's busy time an average SIMD spends executing vector ALU work."""
import csv
import json
import re
import sys
from collections import defaultdict

WANT = ("pair_matrix_kernel", "three_nn_kernel", "ball_query_bf_kernel", "scene_max_kernel",
        "nms_mask_kernel", "grid_query_kernel", "fps_bucket_rounds_kernel", "fps_bucket_setup_kernel", "fps_reg_kernel")
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if not any(w in k for w in WANT):
        continue
    short = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))
    a = agg[short][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
out = {"method": __doc__.split("SQ_ACTIVE_INST_VALU counts")[1].strip().replace("\n", " "), "kernels": {}}
for k, d in agg.items():
    m = {c: v[1] / v[0] for c, v in d.items()}
    busy = m.get("SQ_BUSY_CYCLES", 0.0) / 32.0
    out["kernels"][k] = {
        "launches": int(max(v[0] for v in d.values())),
        "SQ_ACTIVE_INST_VALU": round(m.get("SQ_ACTIVE_INST_VALU", 0.0)),
        "SQ_BUSY_CYCLES": round(m.get("SQ_BUSY_CYCLES", 0.0)),
        "SQ_INSTS_VALU": round(m.get("SQ_INSTS_VALU", 0.0)),
        "raw_ratio_active_valu_over_busy": round(m.get("SQ_ACTIVE_INST_VALU", 0.0) / m["SQ_BUSY_CYCLES"], 4) if m.get("SQ_BUSY_CYCLES") else None,
        "valu_busy": round(m.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / busy, 4) if busy else None}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
