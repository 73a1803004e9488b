"""HBM traffic of the backward GEMM kernels from two rocprofv3 counter passes over
tools/bwd_bench.py (fused one-pass kernel vs the two separate GEMMs, per launch).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d F -o b -- python tools/bwd_bench.py sa2_l2 sa2_l3 sa1_l3
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d W -o b -- python tools/bwd_bench.py sa2_l2 sa2_l3 sa1_l3
    python tools/bwd_pmc.py F/b_counter_collection.csv W/b_counter_collection.csv out.json

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (counter unit KiB; gfx950 reports half of the bytes of
wide coalesced reads -- MI355X_MICROARCH.md, HBM section).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def means(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = re.sub(r"\(.*", "", k)
        if not k.startswith("gemm_"):
            continue
        key = (k, r["Grid_Size_X"] if "Grid_Size_X" in r else "", r.get("Grid_Size_Z", ""))
        a = agg[key]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: (v[1] / v[0], v[0]) for k, v in agg.items()}


f = means(sys.argv[1], "FETCH_SIZE")
w = means(sys.argv[2], "WRITE_SIZE")
out = []
for k in sorted(f, key=lambda k: -f[k][0]):
    fe, n = f[k]
    wr = w.get(k, (0.0, 0))[0]
    out.append({"kernel": k[0], "grid_x": k[1], "grid_z": k[2], "launches": n,
                "fetch_MB": round(2 * fe * 1024 / 1e6, 1), "write_MB": round(wr * 1024 / 1e6, 1),
                "hbm_MB": round((2 * fe + wr) * 1024 / 1e6, 1)})
json.dump(out, open(sys.argv[3], "w"), indent=1)
for o in out:
    print(o)
