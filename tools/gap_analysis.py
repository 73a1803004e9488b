"""Busy time / gaps per queue in a rocprofv3 kernel trace, over the steady-state steps.

    python tools/gap_analysis.py <kernel_trace.csv> <marker substring> <skip>
"""
import csv
import sys
from collections import defaultdict

path, marker, skip = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
t0, t1 = starts[skip], starts[-1]
steps = len(starts) - 1 - skip
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
by_q = defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s < t1:
        by_q[r[qkey]].append((s, e, r["Kernel_Name"]))
print("window %.3f ms, %d steps -> %.3f ms/step" % ((t1 - t0) / 1e6, steps, (t1 - t0) / 1e6 / steps))
for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    pos = [g for g in gaps if g > 0]
    small = [(e - s) for s, e, _ in ks if e - s < 10000]
    print("queue %s: %d kernels/step, busy %.3f ms/step, idle gaps %.3f ms/step (median gap %.2f us), "
          "kernels <10us: %d/step totalling %.3f ms/step"
          % (q, len(ks) // steps, busy / 1e6 / steps, sum(pos) / 1e6 / steps,
             sorted(pos)[len(pos) // 2] / 1e3 if pos else 0, len(small) // steps,
             sum(small) / 1e6 / steps))
