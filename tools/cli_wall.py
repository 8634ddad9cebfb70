#!/usr/bin/env python3
"""Wall time of the C host program on a 512 MiB capture, process start to exit, A/B against the
round-1 program (read -> push in series after the runtime is up) when that binary is present."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gen_signal as G
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = "/tmp/cap512.u16"
if not os.path.exists(path):
    x, _ = G.sparse_capture(256 << 20, 12000, seed=3)
    x.tofile(path)
new = os.path.join(root, "adsbdec_amd", "lib", "adsbdec_amd_cli")
old = os.path.join(root, "adsbdec_amd", "lib_var", "r1cli", "adsbdec_amd_cli_r1")
progs = [("round 2 (reader thread + async pushes)", new, {}), ("round 2, pageable pushes", new, {"ADSB_CLI_REGISTER": "0"})]
if os.path.exists(old):
    progs.append(("round 1 (read, push in series)", old, {}))
open(path, "rb").read()  # page cache
res = {k: [] for k, _, _ in progs}
null = {k: [] for k, _, _ in progs}
outs = {}
reps = 7
for rep in range(reps):
    order = progs[rep % len(progs):] + progs[:rep % len(progs)]   # rotate: the first program of a round pays for cold caches
    for name, exe, env in order:
        e = {**os.environ, "ADSB_CLI_TIMING": "1", **env}
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-f", path], capture_output=True, env=e)
        dt = time.perf_counter() - t0
        res[name].append(dt)
        outs[name] = p.stdout
        t0 = time.perf_counter()   # the same program on a file that does not exist: loader + GPU runtime start, nothing else
        subprocess.run([exe, "-f", path + ".absent"], capture_output=True, env=e)
        null[name].append(time.perf_counter() - t0)
        tm = [l for l in p.stderr.decode().splitlines() if l.startswith("timing")]
        print(f"{name:42s} wall {dt * 1e3:6.1f} ms, without a file {null[name][-1] * 1e3:6.1f} ms  {tm[0] if tm else ''}", flush=True)
assert len(set(outs.values())) == 1, "outputs differ"
med = lambda v: sorted(v)[len(v) // 2]
for name, v in res.items():
    print(f"{name:42s} wall: min {min(v) * 1e3:.1f} ms, median {med(v) * 1e3:.1f} ms; loader + runtime start alone: median "
          f"{med(null[name]) * 1e3:.1f} ms; decode beyond that: {(med(v) - med(null[name])) * 1e3:.1f} ms  ({len(outs[name].splitlines())} frames)")
