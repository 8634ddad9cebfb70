import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi
from bench import make_dense10, bind_near_gpu
torch.cuda.set_device(0); bind_near_gpu(torch, 0)
n = (256 << 20); n -= n % 28
x = make_dense10(torch, n, 101); torch.cuda.synchronize()
d = capi.Decoder(df18=True, profile=True)
for _ in range(3): d.decode_device_raw(x.data_ptr(), x.numel())
os.environ["ADSB_DEBUG_HOST"] = "1"
for _ in range(3): d.decode_device_raw(x.data_ptr(), x.numel())
