"""Per kernel of a rocprofv3 --kernel-trace CSV: launches, mean duration, workgroups per launch and
lanes per workgroup -- to spot launches that do not fill the chip (a few workgroups, long duration:
bound by the latency of their own dependent chains).    python tools/grid_table.py trace.csv [min_us]"""
import csv, sys
from collections import defaultdict
rows = defaultdict(lambda: [0, 0.0, 0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    name = name.split("(")[0][-90:]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    e = rows[(name, grid // max(wg, 1), wg)]
    e[0] += 1
    e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
out = []
for (name, wgs, wg), e in rows.items():
    mean = e[1] / e[0]
    if mean >= min_us and wgs <= 1024:
        out.append((mean * e[0], name, e[0], mean, wgs, wg))
for tot, name, n, mean, wgs, wg in sorted(out, reverse=True)[:60]:
    print("%9.1f us total  %5d x %7.1f us  %6d workgroups x %4d lanes  %s" % (tot, n, mean, wgs, wg, name))
