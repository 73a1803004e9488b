"""Where and when the query kernel's waves ran (csrc/pn2_ball_grid.hip built with -DGRID_PROBE:
s_memtime at a wave's start / end, s_memrealtime at its start, HW_ID and XCC_ID): waves resident
per SIMD over the launch, dispatch rate, slot reuse gaps.

    python tools/micro/grid_probe.py --raw ; python tools/micro/grid_occupancy.py [out.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
res = {}
CLK = 2300.0  # guessed s_memtime ticks per us; re-derived below from the 100 MHz counter
for kind in ("U", "R", "step"):
    f = os.path.join(ROOT, "gpurun_out", "r4a", "grid_probe_raw_%s.npy" % kind)
    if not os.path.exists(f):
        continue
    t = np.load(f).astype(np.uint64)
    t0, t1 = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64)
    rt0 = (t[:, 3] >> np.uint64(16)).astype(np.int64)      # 10 ns ticks, chip-wide
    hw = (t[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
    xcc = ((t[:, 7] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    wave_slot, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    life = (t1 - t0)
    start_us = (rt0 - rt0.min()) / 100.0
    # ticks per us of s_memtime from waves of one CU: (t0 differences) / (rt0 differences)
    key_cu = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ratios = []
    for k in np.unique(key_cu)[:64]:
        m = key_cu == k
        if m.sum() > 8:
            a, b = t0[m], rt0[m]
            if b.max() > b.min():
                ratios.append((a.max() - a.min()) / ((b.max() - b.min()) / 100.0))
    clk = float(np.median(ratios)) if ratios else CLK
    life_us = life / clk
    end_us = start_us + life_us
    span = float(end_us.max())
    r = {"waves": int(len(t0)), "memtime_ticks_per_us": round(clk, 1), "launch_span_us": round(span, 2),
         "wave_life_us": {"mean": round(float(life_us.mean()), 2), "p10": round(float(np.percentile(life_us, 10)), 2),
                          "p90": round(float(np.percentile(life_us, 90)), 2)},
         "xcds": int(len(np.unique(xcc))), "cus": int(len(np.unique(key_cu))),
         "simds": int(len(np.unique(key_cu * 4 + simd)))}
    # chip-wide: waves started per us, waves resident at time x
    grid = np.arange(0.0, span, 0.25)
    started = np.array([(start_us < x).sum() for x in grid])
    resident = np.array([((start_us <= x) & (end_us > x)).sum() for x in grid])
    r["timeline_every_quarter_us"] = {"started": started.tolist(), "resident": resident.tolist()}
    r["dispatch_waves_per_us_first_2us"] = round(float((start_us < 2.0).sum() / 2.0), 1)
    r["peak_resident_per_simd"] = round(float(resident.max() / max(1, r["simds"])), 2)
    r["mean_resident_per_simd"] = round(float(life_us.sum() / span / max(1, r["simds"])), 2)
    # per SIMD: waves, distinct slots used, gap between a slot's waves
    key_simd = key_cu * 4 + simd
    per = [int((key_simd == k).sum()) for k in np.unique(key_simd)]
    r["waves_per_simd"] = {"min": int(min(per)), "median": float(np.median(per)), "max": int(max(per))}
    gaps = []
    for k in np.unique(key_simd)[::8]:
        m = key_simd == k
        for sl in np.unique(wave_slot[m]):
            mm = m & (wave_slot == sl)
            o = np.argsort(start_us[mm])
            s_, e_ = start_us[mm][o], end_us[mm][o]
            gaps += list(s_[1:] - e_[:-1])
    if gaps:
        r["slot_reuse_gap_us"] = {"median": round(float(np.median(gaps)), 2), "p90": round(float(np.percentile(gaps, 90)), 2)}
    r["slots_used_per_simd_max"] = int(max(len(np.unique(wave_slot[key_simd == k])) for k in np.unique(key_simd)[::8]))
    res[kind] = r
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
