"""Print the library's own per-launch kernel time (cfg.profile: in-kernel clock, latest tile end -
earliest tile start) for a few bench-sized steps; run it under `rocprofv3 --kernel-trace` and
compare with the trace's durations."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from adsbdec_amd import capi
from bench import make_workload
torch.cuda.set_device(0)
n = (256 << 20); n -= n % 28
x, _ = make_workload(torch, n, seed=1)
torch.cuda.synchronize()
dec = capi.Decoder(profile=True)
for it in range(16):
    dec.reset(); dec.push_device_final(x.data_ptr(), x.numel()); dec.drain_raw(reuse=True)
    print("event_us %.1f" % (1e3 * dec.profile()["last_kernel_ms"]))
