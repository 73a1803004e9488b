"""Aggregate a rocprofv3 --kernel-trace CSV over the TIMED steps of bench.py only.

    python tools/prof_summary.py <kernel_trace.csv> <marker kernel substring> <skip> <out.csv>

The warm-up steps contain MIOpen's solver search (hundreds of candidate kernels), which would
swamp a whole-run --stats table.  Steps are delimited by the marker kernel (one launch per step,
e.g. the SA1 furthest-point-sampling kernel); the first <skip> steps are dropped.
"""
import csv
import sys
from collections import defaultdict


def main():
    path, marker, skip, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
    if len(starts) <= skip:
        raise SystemExit("marker seen %d times, cannot skip %d" % (len(starts), skip))
    t0 = starts[skip]
    nsteps = len(starts) - skip
    agg = defaultdict(lambda: [0, 0])
    last_end = t0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < t0:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += e - s
        last_end = max(last_end, e)
    total = sum(v[1] for v in agg.values())
    wall = last_end - t0
    with open(out, "w") as f:
        f.write("# steps=%d wall_ms_per_step=%.3f kernel_busy_ms_per_step=%.3f\n"
                % (nsteps, wall / nsteps / 1e6, total / nsteps / 1e6))
        f.write("kernel,calls_per_step,avg_us,ms_per_step,percent_of_busy\n")
        for name, (calls, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%.2f,%.2f,%.4f,%.2f\n' % (name[:160].replace('"', "'"), calls / nsteps,
                                                   dur / calls / 1e3, dur / nsteps / 1e6,
                                                   100.0 * dur / total))
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()
