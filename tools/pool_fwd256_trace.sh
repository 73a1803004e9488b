#!/bin/bash
# kernel durations of the no-store pooled forward per launch shape (run on the GPU box):
#     bash tools/pool_fwd256_trace.sh [lib suffix]
export TMPDIR=/tmp
P=/tmp/pf256trace$1; rm -rf $P
PN2_LIB_SUFFIX=$1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $P -o g -- python tools/pool_fwd256_time.py > /dev/null 2>&1
python - <<EOF
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$P/g_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "pool_fwd256" in n or "bn_pairs" in n:
        d[(n.split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print("lib[$1]", k, "n=%d median %.1f us min %.1f" % (len(v), v[len(v) // 2], v[0]))
EOF
