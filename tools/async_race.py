#!/usr/bin/env python3
"""Diagnosis: repeat an asynchronous small-staging decode and count runs whose frames / statistics differ
from a synchronous decode of the same capture.   python tools/async_race.py [iterations]"""
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: one HIP runtime)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adsbdec_amd import capi  # noqa: E402
from tools import gen_signal as G  # noqa: E402


def rec(fs):
    return [(f["g"], f["ts"], f["pw"], bytes(f["frame"])) for f in fs]


def decode_mixed(d, x, chunk):
    """adsb_push and adsb_push_async alternating, back to back, from page-locked buffers; frames taken at the end only."""
    d.reset()
    L = capi.load()
    with capi.PinnedBuffers(2, chunk) as bufs:
        for k, i in enumerate(range(0, x.size, chunk)):
            piece = x[i:i + chunk]
            b = bufs[k % 2][: piece.size]
            b[:] = piece
            if k % 2:
                d.push_async(b)
            elif L.adsb_push(d._h, b.ctypes.data, b.size) != 0:
                raise RuntimeError("adsb_push failed")
        d.finish()
    return d.drain()


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    x, _ = G.dense_capture((5 << 20) + 6, seed=61, sigma=30.0, n_frames=1200, amp=(150, 1800))
    ref = capi.Decoder(df18=True, collect_stats=True)
    want = rec(ref.decode(x))
    wstats = ref.stats()
    ref.close()
    print("frames", len(want), "stats", wstats, flush=True)
    configs = []
    for stats in (False, True):
        # ADSB_DEBUG_ASYNC is only read by a -DADSB_TUNING build (tools/build_variant.sh tuning -DADSB_TUNING; load it with
        # ADSB_LIB_PATH): 4 switches the ordering rule of the tail copy off -- the old race.  The shipped library ignores it.
        for env, kw in (({}, {}), ({}, {"debug_no_streaming": True}), ({"ADSB_DEBUG_ASYNC": "4"}, {})):
            configs.append((f"async stage=64Ki stats={int(stats)} {env} {kw}", dict(stage=1 << 16, stats=stats, mode="async", env=env, kw=kw)))
    configs += [
        ("sync  stage=64Ki stats=0 chunk 1Mi", dict(stage=1 << 16, stats=False, mode="sync", env={})),
        ("async stage=default stats=0", dict(stage=0, stats=False, mode="async", env={})),
        ("async stage=1Mi stats=0", dict(stage=1 << 20, stats=False, mode="async", env={})),
        ("async stage=64Ki stats=0 chunk 65546", dict(stage=1 << 16, stats=False, mode="async", env={}, chunk=65546)),
        ("async stage=default stats=1 chunk 65546", dict(stage=0, stats=True, mode="async", env={}, chunk=65546)),
        ("async stage=default stats=0 chunk 99998", dict(stage=0, stats=False, mode="async", env={}, chunk=99998)),
    ]
    # round 2's advisor finding: a SYNCHRONOUS push leaves its compaction's tail copy on the scan stream, and the next
    # adsb_push_async copies right behind that tail on a copy stream -- odd sizes put both into one cache line.  Alternate the
    # two calls back to back (no drain in between), and the one-buffer overlap mode (cfg.push_overlap) on its own.
    configs += [
        ("mixed sync/async stage=64Ki stats=0 chunk 65546", dict(stage=1 << 16, stats=False, mode="mixed", env={}, chunk=65546)),
        ("mixed sync/async stage=64Ki stats=1 chunk 33334", dict(stage=1 << 16, stats=True, mode="mixed", env={}, chunk=33334)),
        ("mixed sync/async stage=default stats=0 chunk 99998", dict(stage=0, stats=False, mode="mixed", env={}, chunk=99998)),
        ("overlap stage=64Ki stats=0 chunk 65546", dict(stage=1 << 16, stats=False, mode="overlap", env={}, chunk=65546, overlap=True)),
        ("overlap stage=default stats=1 chunk 1Mi", dict(stage=0, stats=True, mode="overlap", env={}, overlap=True)),
    ]
    only = os.environ.get("RACE_ONLY")
    for name, c in configs:
        if only and only not in name:
            continue
        os.environ.pop("ADSB_DEBUG_ASYNC", None)
        os.environ.update(c["env"])
        d = capi.Decoder(df18=True, collect_stats=c["stats"], stage_samples=c["stage"], push_overlap=c.get("overlap", False), **c.get("kw", {}))
        bad_f = bad_s = 0
        first = None
        for i in range(iters):
            if c["mode"] == "mixed":
                got = rec(decode_mixed(d, x, c.get("chunk", 1 << 20) + 2 * (i % 7)))
            else:
                got = rec(d.decode(x, chunk=c.get("chunk", 1 << 20), mode=c["mode"]))
            if got != want:
                bad_f += 1
                if first is None:
                    a = set(got) ^ set(want)
                    first = (i, len(got), sorted(a)[:4])
            elif c["stats"] and d.stats() != wstats:
                bad_s += 1
                if first is None:
                    first = (i, "stats", d.stats()["try"])
        d.close()
        print(f"{name}: {bad_f} frame mismatches, {bad_s} stats-only mismatches of {iters}; first: {first}", flush=True)


if __name__ == "__main__":
    main()
