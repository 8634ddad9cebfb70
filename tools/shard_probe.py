"""Where does a decode of HBM-resident slices spend its time?  The same 2 Gi-sample stream through (1) the plain stream
path, (2) adsb_scan_shard_resolved_walk (copies out), (3) adsb_scan_shard_resolved_take (in place), (4) the multi-GPU
driver with one handle -- all on the calling thread except (4).  Median ms of N calls each."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from adsbdec_amd import capi, sharding
from bench import make_workload

torch.cuda.set_device(0)
total = int(sys.argv[1]) if len(sys.argv) > 1 else (2 << 30)
total -= total % 28
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x, _ = make_workload(torch, total, seed=9)
torch.cuda.synchronize()
L = capi.load()


def med(f):
    f(); f()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3), round(min(ts), 3)


d = capi.Decoder(df18=True)
print("stream path (adsb_decode_device):", med(lambda: d.decode_device_raw(x.data_ptr(), x.numel())), flush=True)
p = capi.plan_shards(total, 1)[0]
cap = (p["g_end"] - p["g_begin"]) // 39780 + 8
head, bases = capi.ShardHead(), (C.c_uint64 * cap)()
fr, hc = (capi.Frame * 400000)(), (capi.Candidate * 4096)()
d.reset()
print("adsb_scan_shard_resolved_walk (copies out):", med(lambda: L.adsb_scan_shard_resolved_walk(
    d._h, x.data_ptr(), 0, total, p["g_begin"], p["g_end"], total, C.byref(head), fr, 400000, hc, 4096, bases, cap)), head.n_frames, flush=True)
fp, cp = C.POINTER(capi.Frame)(), C.POINTER(capi.Candidate)()
print("adsb_scan_shard_resolved_take (in place):", med(lambda: L.adsb_scan_shard_resolved_take(
    d._h, x.data_ptr(), 0, total, p["g_begin"], p["g_end"], total, C.byref(head), C.byref(fp), C.byref(cp), bases, cap)), head.n_frames, flush=True)
d.close()
for k in (1, 2, 8):
    md = sharding.MultiDecoder(k, [0] * k, df18=True)
    plan = md.plan(total)
    ptrs = [x.data_ptr() + 2 * q["first_sample"] for q in plan]
    r = med(lambda: md.decode_device(total, ptrs))
    print(f"adsb_multi_decode_device, {k} handle(s):", r, md.info(), flush=True)
    md.close()

# ---- is it the thread?  the same in-place call from a second Python thread (handle created on that thread / on the main one)
import threading
for where in ("created on main, called on a thread", "created and called on a thread"):
    box = {}
    d2 = capi.Decoder(df18=True) if where.startswith("created on main") else None

    def body():
        dd = d2 or capi.Decoder(df18=True)
        box["r"] = med(lambda: L.adsb_scan_shard_resolved_take(dd._h, x.data_ptr(), 0, total, p["g_begin"], p["g_end"], total,
                                                                C.byref(head), C.byref(fp), C.byref(cp), bases, cap))
        prof = dd.profile()
        box["prof"] = {k: round(prof[k], 3) if isinstance(prof[k], float) else prof[k] for k in ("launches", "host_ms", "wait_ms")}
    t = threading.Thread(target=body)
    t.start(); t.join()
    print(f"take, {where}:", box["r"], box["prof"], flush=True)
half = total // 2 - (total // 2) % 28
md = sharding.MultiDecoder(1, [0], df18=True)
print("multi 1 handle, 1 Gi (4 launches):", med(lambda: md.decode_device(half, [x.data_ptr()])), flush=True)
md.close()
