import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi
from bench import make_workload, bind_near_gpu
torch.cuda.set_device(0)
bind_near_gpu(torch, 0)
total = (2 << 30); total -= total % 28
x, _ = make_workload(torch, total, seed=9)
torch.cuda.synchronize()
L = capi.load()
def run(f, n=10):
    for _ in range(3): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3)
p = capi.plan_shards(total, 1)[0]
cap = (p["g_end"] - p["g_begin"]) // 39780 + 8
head, bases = capi.ShardHead(), (C.c_uint64 * cap)()
fp, cp = C.POINTER(capi.Frame)(), C.POINTER(capi.Candidate)()
for stats in (False, True):
    d = capi.Decoder(df18=True, collect_stats=stats, profile=True)
    take = lambda: L.adsb_scan_shard_resolved_take(d._h, x.data_ptr(), 0, total, p["g_begin"], p["g_end"], total, C.byref(head), C.byref(fp), C.byref(cp), bases, cap)
    a = d.profile()
    print("take alone, stats =", stats, run(take), {k: round((d.profile()[k] - a[k]) / 13, 3) for k in ("host_ms", "wait_ms", "kernel_ms")}, flush=True)
    if stats:
        def with_head():
            d.scan_shard(x.data_ptr(), 0, 2 * (17584 + 1196), 0, 17584)
            take()
        print("head window scan + take:", run(with_head), flush=True)
        def with_both():
            d.scan_shard(x.data_ptr(), 0, 2 * (17584 + 1196), 0, 17584)
            take()
            tf = p["g_end"] - 42181; tf -= tf % 28
            d.scan_shard(x.data_ptr() + 4 * (tf - 8), 2 * (tf - 8), total - 2 * (tf - 8), tf, p["g_end"])
        print("head + take + tail:", run(with_both), flush=True)
        t0 = time.perf_counter(); d.scan_shard(x.data_ptr(), 0, 2 * (17584 + 1196), 0, 17584); print("one window scan:", round((time.perf_counter() - t0) * 1e3, 3))
    d.close()
