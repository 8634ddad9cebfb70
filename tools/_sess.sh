ADSB_CLOCK_OUT=1 ADSB_LIB_PATH=adsbdec_amd/lib_var/clk3/libadsbdec_amd.so timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>&1 | grep "tile phases" | tail -2
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from adsbdec_amd import capi
from bench import make_workload
torch.cuda.set_device(0)
n = (256<<20); n -= n % 28
x, _ = make_workload(torch, n, seed=1)
torch.cuda.synchronize()
for rep in range(2):
  for prof in (True, False):
    dec = capi.Decoder(profile=prof)
    for it in range(1500):
        if it == 500: torch.cuda.synchronize(); t0 = time.perf_counter()
        dec.reset(); dec.push_device_final(x.data_ptr(), x.numel()); dec.take_raw()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"profile={prof}: step {dt/1000*1e3:.4f} ms", flush=True)
    dec.close()
PY
