import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from adsbdec_amd import capi
from bench import make_workload
torch.cuda.set_device(0)
n = (256<<20); n -= n % 28
x, truth = make_workload(torch, n, seed=1)
torch.cuda.synchronize()
dec = capi.Decoder(profile=True, collect_stats=bool(os.environ.get("ADSB_STATS")), debug_try_cap=int(os.environ.get("ADSB_TRYCAP", "0")))
for it in range(int(os.environ.get("ADSB_STEPS", "8"))):
    t0=time.perf_counter(); dec.reset()
    t1=time.perf_counter(); dec.push_device_final(x.data_ptr(), x.numel())
    t2=time.perf_counter()
    t3=time.perf_counter(); raw = dec.take_raw()
    t4=time.perf_counter()
    p = dec.profile()
    q = prev if 'prev' in dir() and prev else {k: 0 for k in p}
    prev = p
    p = dict(p, kernel_ms=p['kernel_ms'] - q['kernel_ms'], launches=p['launches'] - q['launches'],
             candidates=p['candidates'] - q['candidates'], host_ms=p['host_ms'] - q['host_ms'], wait_ms=p['wait_ms'] - q['wait_ms'])
    print(f"reset {1e3*(t1-t0):.3f} push {1e3*(t2-t1):.3f} finish {1e3*(t3-t2):.3f} drain {1e3*(t4-t3):.3f} ms | kernel {p['kernel_ms']:.3f} ms launches {p['launches']} cands {p['candidates']} host {p['host_ms']:.3f} wait {p['wait_ms']:.3f}")
