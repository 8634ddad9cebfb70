#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into the small summary
kept under profiles/: the stats rows of this library's kernels (plus the copy/fill
helpers) and one line per dispatch of them.

    python tools/summarize_prof.py gpurun_out/prof_xxx profiles/r1_xxx
"""
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
keep = ("adsb::", "__amd_rocclr")
stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
trace = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
with open(stats) as f, open(dst + "_kernel_stats.csv", "w", newline="") as g:
    r = csv.reader(f)
    w = csv.writer(g)
    w.writerow(next(r))
    for row in r:
        if row[0].startswith(("void adsb::", "adsb::", "__amd_rocclr")) or "adsb::" in row[0]:
            w.writerow(row)
with open(trace) as f, open(dst + "_dispatches.csv", "w", newline="") as g:
    r = csv.DictReader(f)
    w = csv.writer(g)
    w.writerow(["Kernel_Name", "Duration_ns", "Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count",
                "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size"])
    for row in r:
        if "adsb::" in row["Kernel_Name"]:
            w.writerow([row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                        row["Grid_Size_X"], row["Workgroup_Size_X"], row["LDS_Block_Size"], row["VGPR_Count"],
                        row["Accum_VGPR_Count"], row["SGPR_Count"], row["Scratch_Size"]])
print(open(dst + "_kernel_stats.csv").read())
