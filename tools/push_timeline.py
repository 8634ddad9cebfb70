#!/usr/bin/env python3
"""Timeline of adsb_push_async at the reference's 1 Mi-sample call size.
Run under `rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o run -- python3 tools/push_timeline.py`,
then `python3 tools/push_timeline.py --summarize DIR` prints copies and scan kernels on one time axis."""
import csv, glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    d = sys.argv[2]
    ev = []
    for path in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?")))
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "adsb::" in r["Kernel_Name"]:
                ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-24:]))
    ev.sort()
    # the last 40 % of the events: steady state
    tail = ev[int(len(ev) * 0.6):][:60]
    t0 = tail[0][0]
    for s, e, n in tail:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {n}")
    sys.exit(0)

import numpy as np, torch
from adsbdec_amd import capi
torch.cuda.set_device(0)
n, chunk = 64 << 20, 1 << 20
dec = capi.Decoder()
with capi.PinnedBuffers(1, n) as bufs:
    rng = np.random.default_rng(1)
    bufs[0][:] = (2048 + rng.normal(0, 8, n)).clip(0, 4095).astype(np.uint16)
    base = bufs[0].ctypes.data
    for rep in range(3):
        dec.reset()
        t0 = time.perf_counter()
        for i in range(0, n, chunk):
            dec.push_async((base + 2 * i, chunk))
            dec.take_raw()
        dec.finish()
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {n / dt / 1e9:.2f} GS/s, {dt / (n // chunk) * 1e6:.1f} us per push", file=sys.stderr)
