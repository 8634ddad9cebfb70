#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (one directory per pass) over `python3 bench.py ...`
into the JSON kept under profiles/: per-counter mean over the scan_kernel dispatches of the
dominant launch size, and the HBM bytes per launch as MI355X_MICROARCH.md prescribes
(FETCH_SIZE and WRITE_SIZE in separate passes, in KiB... x1024; gfx950 reports half of the
bytes of wide 16 B/lane streaming reads -> FETCH_SIZE x 2).

    python tools/summarize_pmc.py profiles/r1_v9_pmc.json gpurun_out/pmc_a gpurun_out/pmc_b ...
"""
import collections
import csv
import glob
import json
import os
import sys

dst, srcs = sys.argv[1], sys.argv[2:]
vals = collections.defaultdict(list)
grid = collections.Counter()
rows_all = []
for src in srcs:
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if "scan_kernel" in r["Kernel_Name"]:
                    rows_all.append(r)
                    grid[int(r["Grid_Size"])] += 1
big = max(grid)  # dominant launch = the largest grid
for r in rows_all:
    if int(r["Grid_Size"]) == big:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"what": "rocprofv3 --pmc passes over `python3 bench.py --steps 5 --warmup 1 --preroll-ms 0 --no-cpu-baseline` (MI355X, ROCm 7.2), "
               "adsb::scan_kernel dispatches of the dominant launch size only; one pass per counter group as "
               "MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE cannot share a pass)",
       "grid_size": big, "counters": {}}
for k, v in sorted(vals.items()):
    out["counters"][k] = {"n": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    f = out["counters"]["FETCH_SIZE"]["mean"]
    w = out["counters"]["WRITE_SIZE"]["mean"]
    out["FETCH_SIZE_KB_mean"], out["WRITE_SIZE_KB_mean"] = f, w
    out["correction"] = "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads -> x2; WRITE_SIZE exact"
    out["hbm_bytes_per_launch"] = (2 * f + w) * 1024
if len(sys.argv) > 2 and os.environ.get("LAUNCH_OFFSETS"):
    out["launch_offsets"] = int(os.environ["LAUNCH_OFFSETS"])
    out["algorithmic_bytes_per_launch"] = 4 * out["launch_offsets"]
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "counters"}, indent=1))
for k, v in out["counters"].items():
    print(f"{k:28s} n={v['n']:3d} mean={v['mean']:.6g}")
