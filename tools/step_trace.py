#!/usr/bin/env python3
"""Where the microseconds of one adsb_decode_device call go, from a rocprofv3 trace of bench.py:

    cd /tmp && rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /tmp/steptrace -o run -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline
    python tools/step_trace.py /tmp/steptrace

For each large scan launch: the HIP calls of the calling thread between the previous launch's end and this kernel's start
(how long before the kernel starts was the launch call made, which calls lie in between), and what happens between the kernel's
end and the next launch call.  Prints medians over the steps.
"""
import csv
import glob
import statistics
import sys


def main():
    d = sys.argv[1]
    api, ker = [], []
    for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "?")))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "scan_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 600000:
                ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    api.sort()
    ker.sort()
    if len(ker) < 12:
        raise SystemExit(f"only {len(ker)} large scan launches in the trace")
    ker = ker[-20:]
    launches = [(a, b, t) for a, b, f, t in api if f == "hipLaunchKernel"]
    rows = []
    for i in range(1, len(ker)):
        k0, k1 = ker[i]
        pk1 = ker[i - 1][1]
        # the scan's own launch call: the last hipLaunchKernel that STARTED before the kernel did, minus the report kernel's
        mine = [x for x in launches if x[0] < k0 and x[0] > pk1]
        if not mine:
            continue
        first_call = min(a for a, b, f, t in api if pk1 < a < k0 and t == mine[0][2]) if mine else k0
        la, lb, tid = mine[0]
        between = [(f, (b - a) / 1e3) for a, b, f, t in api if t == tid and pk1 < a < k0]
        rows.append({"prev_kernel_end_to_first_call": (first_call - pk1) / 1e3, "first_call_to_launch_call": (la - first_call) / 1e3,
                     "launch_call": (lb - la) / 1e3, "launch_call_end_to_kernel_start": (k0 - lb) / 1e3, "kernel": (k1 - k0) / 1e3,
                     "n_calls_before_kernel": len(between), "period": (k0 - ker[i - 1][0]) / 1e3, "calls": between})
    for k in ("period", "kernel", "prev_kernel_end_to_first_call", "first_call_to_launch_call", "launch_call", "launch_call_end_to_kernel_start",
              "n_calls_before_kernel"):
        print(f"{k:36s} median {statistics.median(r[k] for r in rows):9.2f}")
    print("the calling thread's HIP calls between the previous kernel's end and this kernel's start (one step):")
    for f, us in rows[len(rows) // 2]["calls"]:
        print(f"   {us:8.2f} us  {f}")


if __name__ == "__main__":
    main()
