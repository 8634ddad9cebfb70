"""Cost of the Try/Ok table on multi-launch streams: the plain stream path and the sharded path, with and without collect_stats."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi, sharding
from bench import make_workload, bind_near_gpu
torch.cuda.set_device(0)
print(bind_near_gpu(torch, 0))
total = int(sys.argv[1]) if len(sys.argv) > 1 else (2 << 30); total -= total % 28
x, _ = make_workload(torch, total, seed=9)
torch.cuda.synchronize()
def run(f, n=15):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return round((time.perf_counter() - t0) / n * 1e3, 3)
def pd(a, b, n=18):
    return {k: round((b[k] - a[k]) / n, 3) for k in ("launches", "relaunches", "kernel_ms", "host_ms", "wait_ms", "tries")}
for stats in (False, True):
    d = capi.Decoder(df18=True, collect_stats=stats, profile=True)
    a = d.profile()
    ms = run(lambda: d.decode_device_raw(x.data_ptr(), x.numel()))
    print("stream path, stats =", stats, ms, "ms", pd(a, d.profile()), flush=True)
    if stats:
        t0 = time.perf_counter(); d.stats(); print("   adsb_get_stats:", round((time.perf_counter() - t0) * 1e3, 3), "ms")
    d.close()
    for k in (1, 4):
        md = sharding.MultiDecoder(k, [0] * k, df18=True, collect_stats=stats, profile=True)
        plan = md.plan(total)
        ptrs = [x.data_ptr() + 2 * q["first_sample"] for q in plan]
        a = md.worker_profile(0)
        ms = run(lambda: md.decode_device(total, ptrs))
        print(f"sharded, {k} handle(s), stats =", stats, ms, "ms", pd(a, md.worker_profile(0)), {q: round(md.info()[q], 3) for q in ("workers_ms", "serial_us", "stitch_us")}, flush=True)
        md.close()
