import ctypes as C, os, sys, time, threading
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
p = capi.plan_shards(total, 1)[0]
cap = (p["g_end"] - p["g_begin"]) // 39780 + 8
def body(tag, with_windows):
    head, bases = capi.ShardHead(), (C.c_uint64 * cap)()
    fp, cp = C.POINTER(capi.Frame)(), C.POINTER(capi.Candidate)()
    d = capi.Decoder(df18=True, collect_stats=True, profile=True)
    cands = (capi.Candidate * 4096)(); tries = (C.c_uint64 * 65536)(); nc, nt = C.c_size_t(0), C.c_size_t(0)
    tf = p["g_end"] - 42181; tf -= tf % 28
    def f():
        if with_windows:
            L.adsb_scan_shard(d._h, x.data_ptr(), 0, 2 * (17584 + 1196), 0, 17584, cands, 4096, C.byref(nc), tries, 65536, C.byref(nt))
        L.adsb_scan_shard_resolved_take(d._h, x.data_ptr(), 0, total, p["g_begin"], p["g_end"], total, C.byref(head), C.byref(fp), C.byref(cp), bases, cap)
        if with_windows:
            L.adsb_scan_shard(d._h, x.data_ptr() + 4 * (tf - 8), 2 * (tf - 8), total - 2 * (tf - 8), tf, p["g_end"], cands, 4096, C.byref(nc), tries, 65536, C.byref(nt))
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); f(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(tag, "windows" if with_windows else "no windows", ts, flush=True)
    d.close()
body("main thread", False); body("main thread", True)
for w in (False, True):
    t = threading.Thread(target=body, args=("python thread", w)); t.start(); t.join()
