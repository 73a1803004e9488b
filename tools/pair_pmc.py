"""HBM traffic of the north-star pair kernels from two rocprofv3 counter passes.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d F -o pair -- python tools/pair_bench.py 5 --plain
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d W -o pair -- python tools/pair_bench.py 5 --plain
    rocprofv3 --kernel-trace --stats    --output-format csv -d S -o pair -- python tools/pair_bench.py 10 --plain
    python tools/pair_pmc.py F/pair_counter_collection.csv W/pair_counter_collection.csv \
        S/pair_kernel_stats.csv profiles/r3_pair_pmc.json

Counters are collected in their own passes (MI355X_MICROARCH.md, HBM section).  Units: the counter
values are KiB; on gfx950 FETCH_SIZE reports HALF of the bytes of wide coalesced reads, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- an upper estimate where a kernel's reads are
narrow gathers (uncalibrated, per the guide).
"""
import csv
import json
import sys
from collections import defaultdict

PAIR_BYTES = 38516736
KERNELS = {  # substring -> role
    "grid_split_kernel": "cell-list build, pass 1 (split by z-layer)",
    "grid_bin_kernel": "cell-list build, pass 2 (CSR rows of a layer)",
    "grid_query_kernel<192, 1, true, true, 1>": "fused ball query + gathers from the sampling kernel's query plans (the dominant kernel: the `layer` form)",
    "grid_query_kernel<192, 1, true, false, 1>": "fused ball query + gathers, rows computed by the wave (self-contained form)",
    "grid_query_kernel<192, 1, false, false, 1>": "ball query only (reference operator surface)",
    "group_points_lds_kernel": "group_points (reference operator surface), two launches per pair",
}


def means(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in agg.items()}


def main():
    fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    dur = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[3]))}
    out = {"what": "north-star pair @ B=8 N=40000 m=2048 ns=64: HBM traffic per launch",
           "method": __doc__.split("Counters are")[1].strip().replace("\n", " "),
           "algorithmic_bytes": PAIR_BYTES, "kernels": []}
    for key, role in KERNELS.items():
        name = next((k for k in fetch if key in k), None)
        if name is None:
            continue
        f, w = fetch[name], write.get(name, 0.0)
        d = next((v for k, v in dur.items() if key in k), None)
        out["kernels"].append({"kernel": key, "role": role,
                               "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1),
                               "hbm_bytes_corrected": int((2 * f + w) * 1024),
                               "avg_us_under_profiler": None if d is None else round(d, 2)})
    fused = next((k for k in out["kernels"] if "1, true, true, 1>" in k["kernel"]), None) or \
        next((k for k in out["kernels"] if "1, true, false, 1>" in k["kernel"]), None)
    if fused:
        out["traffic_bytes_fused_kernel"] = fused["hbm_bytes_corrected"]
        out["traffic_over_algorithmic"] = round(fused["hbm_bytes_corrected"] / PAIR_BYTES, 3)
    build = [k for k in out["kernels"] if "grid_split" in k["kernel"] or "grid_bin" in k["kernel"]]
    if fused and len(build) == 2:
        out["traffic_bytes_self_contained"] = fused["hbm_bytes_corrected"] + sum(
            k["hbm_bytes_corrected"] for k in build)
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
