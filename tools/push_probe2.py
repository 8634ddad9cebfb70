"""The first 16 pushes of a process, one by one (no warm-up): which of them pay a one-time cost, and is it a relaunch?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi
torch.cuda.set_device(0)
L = capi.load()
N, NB = 16 << 20, 16
rng = np.random.default_rng(5)
noise = (2048 + rng.normal(0, 20, N)).clip(0, 4095).astype(np.uint16)
bufs = []
for _ in range(NB):
    p = L.adsb_host_alloc(2 * N)
    a = np.frombuffer((C.c_uint16 * N).from_address(p), dtype=np.uint16)
    a[:] = noise
    bufs.append(p)
for stats in (True, False):
    for take in (True, False):
        d = capi.Decoder(df18=False, collect_stats=stats, profile=False)
        per = []
        for p in bufs:
            t1 = time.perf_counter()
            d.push_async((p, N))
            per.append((time.perf_counter() - t1) * 1e3)
            if take:
                d.take_raw()
        t1 = time.perf_counter(); d.finish(); fin = (time.perf_counter() - t1) * 1e3
        pr = d.profile()
        print(f"stats={int(stats)} take={int(take)}: " + " ".join(f"{x:.2f}" for x in per) + f" | finish {fin:.2f} | launches {pr['launches']} relaunches {pr['relaunches']}", flush=True)
        d.close()
