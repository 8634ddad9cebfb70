import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi, sharding
from bench import make_workload, bind_near_gpu
torch.cuda.set_device(0)
bind_near_gpu(torch, 0)
total = (2 << 30); total -= total % 28
x, _ = make_workload(torch, total, seed=9)
torch.cuda.synchronize()
for k, stats in ((1, True), (1, False), (4, True)):
    md = sharding.MultiDecoder(k, [0] * k, df18=True, collect_stats=stats, profile=True)
    plan = md.plan(total)
    ptrs = [x.data_ptr() + 2 * q["first_sample"] for q in plan]
    rows = []
    for _ in range(16):
        t0 = time.perf_counter(); md.decode_device(total, ptrs); dt = (time.perf_counter() - t0) * 1e3
        i = md.info()
        rows.append((round(dt, 2), round(i["workers_ms"], 2), round(i["serial_us"]), round(i["total_ms"], 2)))
    print(k, stats, rows, flush=True)
    md.close()
