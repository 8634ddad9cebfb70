"""Kernel time of N back-to-back bench-sized steps (cfg.profile's in-kernel clock): shows the chip's
clock management -- a fast start, a slow stretch after ~2 ms of load, and the steady state."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from adsbdec_amd import capi
from bench import make_workload
torch.cuda.set_device(0)
n = (256 << 20); n -= n % 28
x, _ = make_workload(torch, n, seed=1, n_frames=int(os.environ["KTC_FRAMES"]) if "KTC_FRAMES" in os.environ else None)
torch.cuda.synchronize()
dec = capi.Decoder(profile=True)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ks = []
for it in range(steps):
    dec.reset(); dec.push_device_final(x.data_ptr(), x.numel()); dec.take_raw()
    ks.append(1e3 * dec.profile()["last_kernel_ms"])
for i in range(0, steps, 20):
    seg = ks[i:i + 20]
    print(f"steps {i:4d}..{i + len(seg) - 1:4d}: mean {sum(seg) / len(seg):6.1f} us  min {min(seg):6.1f}  max {max(seg):6.1f}")
