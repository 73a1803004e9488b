"""Per-kernel table of every counter in one or more rocprofv3 --pmc counter_collection CSVs
(mean per launch).

    python tools/pmc_table.py out.csv [--filter substr] pass1_counter_collection.csv [pass2 ...]
"""
import csv
import sys
from collections import OrderedDict, defaultdict

args = sys.argv[1:]
out = args.pop(0)
flt = None
if args and args[0] == "--filter":
    flt = args[1]
    args = args[2:]
vals = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
counters = OrderedDict()
for path in args:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        c = r["Counter_Name"]
        counters[c] = 1
        a = vals[k][c]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
with open(out, "w") as f:
    f.write("kernel,launches," + ",".join(counters) + "\n")
    for k, cs in vals.items():
        n = max(v[0] for v in cs.values())
        f.write('"%s",%d,' % (k[:120].replace('"', "'"), n) +
                ",".join("%.0f" % (cs[c][1] / cs[c][0]) if c in cs else "" for c in counters) + "\n")
print(open(out).read()[:6000])
