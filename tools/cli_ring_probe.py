"""Wall time of the C host program on a 512 MiB tmpfs file for several read-ahead ring sizes (ADSB_CLI_RING_MB), and through -G 0."""
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from adsbdec_amd import capi
from tools import gen_signal as G
x, _ = G.dense_capture(256 << 20, seed=5, sigma=8.0, n_frames=13000, amp=(200, 1500)) if len(sys.argv) < 2 else (np.fromfile(sys.argv[1], np.uint16), None)
path = "/dev/shm/cli_ring_probe.u16"
x.tofile(path)
def run(args, env):
    ws = []
    for _ in range(5):
        t0 = time.perf_counter()
        p = subprocess.run([capi.CLI_PATH] + args + ["-f", path], capture_output=True, env=dict(os.environ, ADSB_CLI_TIMING="1", **env))
        ws.append((time.perf_counter() - t0) * 1e3)
        assert p.returncode == 0, p.stderr
    t = [ln for ln in p.stderr.decode().splitlines() if ln.startswith("timing")][0]
    return round(sorted(ws)[2], 1), p.stdout.count(b"\n"), t
for mb in (1024, 192):
    print("ring", mb, "MiB:", run([], {"ADSB_CLI_RING_MB": str(mb)}), flush=True)
print("no hipHostRegister (ADSB_CLI_REGISTER=0):", run([], {"ADSB_CLI_REGISTER": "0"}), flush=True)
print("-G 1:", run(["-G", "1"], {}), flush=True)
print("-G 0,0:", run(["-G", "0,0"], {}), flush=True)
os.unlink(path)
