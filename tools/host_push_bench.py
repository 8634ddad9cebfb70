"""PCIe-inclusive rate of adsb_push() (host memory -> frames); never reported as bench `value`."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi
torch.cuda.set_device(0)
n = 128 << 20
rng = np.random.default_rng(1)
x = (2048 + rng.normal(0, 8, n)).clip(0, 4095).astype(np.uint16)
xp = torch.from_numpy(x.view(np.int16)).pin_memory().numpy().view(np.uint16)
dec = capi.Decoder()
for name, arr in (("pageable", x), ("pinned", xp)):
    for chunk in (1 << 20, 16 << 20, None):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); dec.decode(arr, chunk=chunk); best = min(best, time.perf_counter() - t0)
        print(f"{name:9s} chunk={chunk}: {n / best / 1e9:.2f} Gsamples/s ({2 * n / best / 1e9:.1f} GB/s)")
