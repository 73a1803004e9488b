"""Per-step kernel breakdown from a rocprofv3 --kernel-trace csv of bench.py: the timed steps are
delimited by the once-per-step optimizer kernel (adam_update_kernel; the loss' finalize kernel runs
twice per semi-supervised step since round 5); prints time per kernel per step, busiest first.

    python tools/step_breakdown.py <kernel_trace.csv> [steps=20] [top=40] [out.csv]
"""
import collections
import csv
import re
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out_csv = sys.argv[4] if len(sys.argv) > 4 else None
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    return n


names = [short(r["Kernel_Name"]) for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]
en = [int(r["End_Timestamp"]) for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith("adam_update_kernel")]
lo, hi = adam[-steps - 1] + 1, adam[-1] + 1
tot, cnt = collections.Counter(), collections.Counter()
for i in range(lo, hi):
    tot[names[i]] += en[i] - st[i]
    cnt[names[i]] += 1
span = (max(en[lo:hi]) - min(st[lo:hi])) / steps / 1e6
busy = sum(tot.values()) / steps / 1e6
print("span %.3f ms/step, kernel time summed over streams %.3f ms/step, %.0f kernels/step"
      % (span, busy, (hi - lo) / steps))
for n, t in tot.most_common(top):
    print("%8.1f us %5.1f%% x%6.1f  %s" % (t / steps / 1e3, 100 * t / sum(tot.values()), cnt[n] / steps, n[:120]))

if out_csv:
    with open(out_csv, "w") as f:
        f.write("kernel,launches_per_step,us_per_step,percent_of_kernel_time\n")
        f.write('"(all kernels, both streams; profiler serialises them)",%.1f,%.1f,100.0\n'
                % ((hi - lo) / steps, busy * 1e3))
        for n, t in tot.most_common():
            f.write('"%s",%.1f,%.1f,%.2f\n' % (n.replace('"', "'"), cnt[n] / steps, t / steps / 1e3,
                                              100 * t / sum(tot.values())))
