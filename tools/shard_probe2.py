import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi, sharding
from bench import make_workload
torch.cuda.set_device(0)
total = (1 << 30); total -= total % 28
x, _ = make_workload(torch, total, seed=9)
torch.cuda.synchronize()
def prof_delta(a, b, n):
    return {k: round((b[k] - a[k]) / n, 4) for k in ("launches", "kernel_ms", "host_ms", "wait_ms", "candidates")}
for k in (1, 2):
    md = sharding.MultiDecoder(k, [0] * k, df18=True, profile=True)
    plan = md.plan(total)
    ptrs = [x.data_ptr() + 2 * q["first_sample"] for q in plan]
    for _ in range(3): md.decode_device(total, ptrs)
    p0 = [md.worker_profile(i) for i in range(k)]
    t0 = time.perf_counter()
    for _ in range(20): md.decode_device(total, ptrs)
    dt = (time.perf_counter() - t0) / 20 * 1e3
    p1 = [md.worker_profile(i) for i in range(k)]
    print(k, "handles:", round(dt, 3), "ms/step;", [prof_delta(a, b, 20) for a, b in zip(p0, p1)], {q: md.info()[q] for q in ("workers_ms", "serial_us", "workers_bound")}, flush=True)
    md.close()
d = capi.Decoder(df18=True, profile=True)
p = capi.plan_shards(total, 1)[0]
cap = (p["g_end"] - p["g_begin"]) // 39780 + 8
head, bases = capi.ShardHead(), (C.c_uint64 * cap)()
fp, cp = C.POINTER(capi.Frame)(), C.POINTER(capi.Candidate)()
L = capi.load()
f = lambda: L.adsb_scan_shard_resolved_take(d._h, x.data_ptr(), 0, total, p["g_begin"], p["g_end"], total, C.byref(head), C.byref(fp), C.byref(cp), bases, cap)
for _ in range(3): f()
a = d.profile(); t0 = time.perf_counter()
for _ in range(20): f()
dt = (time.perf_counter() - t0) / 20 * 1e3
print("direct:", round(dt, 3), prof_delta(a, d.profile(), 20))
print("cpus allowed:", len(os.sched_getaffinity(0)), "of", os.cpu_count())
