#!/usr/bin/env python3
"""Diagnostic: decode DIFFERENT device-resident captures in rotation on one handle and compare every
step's frames with the oracle's for that capture; on a mismatch say what differs and whether the
foreign records belong to another capture of the rotation (stale hand-off bytes).

    python tools/rotate_check.py [--samples N] [--steps K] [--captures C]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adsbdec_amd import capi
from oracle import oracle as O
from bench import make_workload

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=256 << 20)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--captures", type=int, default=3)
ap.add_argument("--same-size", type=int, default=1)
args = ap.parse_args()
torch.cuda.set_device(0)
key = lambda fs: [(f["g"], f["ts"], f["pw"], f["frame"]) for f in fs]
caps = []
for j in range(args.captures):
    n = args.samples - (0 if args.same_size else 28 * 40_000 * j)
    n -= n % 28
    x, _ = make_workload(torch, n, seed=1 + 1000 * j)
    want = key(O.decode(x.cpu().numpy().view(np.uint16), df18=False)[0])
    caps.append((x, want))
    print(f"capture {j}: {n} samples, {len(want)} frames", flush=True)
dec = capi.Decoder(df18=False, profile=True)
bad = 0
for i in range(args.steps):
    j = i % args.captures
    x, want = caps[j]
    dec.reset()
    dec.push_device_final(x.data_ptr(), x.numel())
    p, n = dec.take_raw()
    got = key(capi._frames_to_dicts(p, n))
    if got != want:
        bad += 1
        sg, sw = set(got), set(want)
        extra, missing = sorted(sg - sw), sorted(sw - sg)
        others = {r: jj for jj, (_, w) in enumerate(caps) if jj != j for r in w}
        stale = sum(1 for r in extra if r in others)
        stale_g = sum(1 for r in extra if any(r[0] == o[0] for o in others))
        print(f"step {i} capture {j}: got {len(got)} want {len(want)}; extra {len(extra)} (stale-from-other-capture {stale}), "
              f"missing {len(missing)}; first extra {extra[:2]} first missing {missing[:2]}", flush=True)
        if bad > 8:
            break
print("mismatching steps:", bad, "of", i + 1)
