"""Where does a step on the dense captures go?  Profile counters per step: kernel, host (collect + resolve), waits, records."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi
from bench import make_dense10, make_gate_storm, make_dense, bind_near_gpu
torch.cuda.set_device(0)
bind_near_gpu(torch, 0)
n = (256 << 20); n -= n % 28
for name, make in (("noise", lambda: make_dense(torch, n, 100)), ("dense10", lambda: make_dense10(torch, n, 101)), ("gate_storm", lambda: make_gate_storm(torch, n, 102))):
    x = make()
    torch.cuda.synchronize()
    for kw in (dict(), dict(host_threads=1), dict(host_threads=2), dict(host_threads=3), dict(host_threads=4), dict(host_threads=5), dict(host_threads=6), dict(host_threads=8)):
        d = capi.Decoder(df18=True, profile=True, **kw)
        for _ in range(3): d.decode_device_raw(x.data_ptr(), x.numel())
        a = d.profile(); t0 = time.perf_counter()
        for _ in range(10): r = d.decode_device_raw(x.data_ptr(), x.numel())
        dt = (time.perf_counter() - t0) / 10 * 1e3
        b = d.profile()
        print(name, kw, "step", round(dt, 3), "ms; frames", r[1], {k: round((b[k] - a[k]) / 10, 3) for k in ("launches", "relaunches", "kernel_ms", "host_ms", "wait_ms", "candidates")}, flush=True)
        d.close()
    del x
