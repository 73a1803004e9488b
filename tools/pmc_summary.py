"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel, mean counter value per launch.

    python tools/pmc_summary.py <counter_collection.csv> <COUNTER> <out.csv>
"""
import csv
import sys
from collections import defaultdict

path, counter, out = sys.argv[1], sys.argv[2], sys.argv[3]
agg = defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    if r.get("Counter_Name") != counter:
        continue
    a = agg[r["Kernel_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
with open(out, "w") as f:
    f.write("kernel,launches,%s_mean_per_launch\n" % counter)
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%.1f\n' % (k[:150].replace('"', "'"), n, v / n))
print(open(out).read()[:3000])
