#!/usr/bin/env python3
"""Leak soak: many decodes on long-lived handles (plain, with the Try/Ok table, the multi-GPU driver with four handles on this
device, a dense capture that brings the second host thread up); host RSS and free device memory before and after each leg.
    python tools/soak_probe.py [--seconds 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from adsbdec_amd import capi, sharding  # noqa: E402


def rss_mb():
    """Resident MiB of this process in mappings below 100 MiB -- the heap and the threads' malloc arenas, where a leak of this
    library's would show -- and, separately, the number of anonymous mappings of 100 MiB and more: the GPU runtime grows its
    own pools in chunks of 173 MiB (three exist before this library is loaded; a new one appears every few hundred calls of a
    multi-threaded host-fed run for a while), which is not this code's to free."""
    small, big, cur = 0, 0, 0
    with open("/proc/self/smaps") as f:
        for ln in f:
            p = ln.split()
            if p and "-" in p[0] and len(p) >= 5 and not p[0].endswith(":"):
                a, b = p[0].split("-")
                cur = (int(b, 16) - int(a, 16)) >> 20
                big += cur >= 100 and len(p) == 5
            elif ln.startswith("Rss:") and cur < 100:
                small += int(p[1])
    return small / 1024.0, big


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    n = (64 << 20) - (64 << 20) % 28
    sparse, _ = bench.make_workload(torch, n, seed=3)
    dense = bench.make_dense10(torch, n, 7)
    legs = []
    for name, t, kw in (("plain", sparse, {}), ("stats", sparse, dict(collect_stats=True)), ("dense10, stats", dense, dict(collect_stats=True, df18=True))):
        d = capi.Decoder(**kw)
        for _ in range(50):
            d.decode_device_raw(t.data_ptr(), t.numel())
        torch.cuda.synchronize()
        (r0, b0), f0, t0, it = rss_mb(), torch.cuda.mem_get_info()[0], time.time(), 0
        while time.time() - t0 < a.seconds:
            for _ in range(100):
                d.decode_device_raw(t.data_ptr(), t.numel())
            it += 100
        torch.cuda.synchronize()
        r1, b1 = rss_mb()
        legs.append((name, it, r1 - r0, (f0 - torch.cuda.mem_get_info()[0]) / 2 ** 20, b1 - b0))
        d.close()
    x = sparse.cpu().numpy().view("uint16")
    # host pushes at the reference's call size (1 Mi samples, air.c:218): from pageable memory (adsb_push), with cfg.push_overlap,
    # and from two page-locked buffers in turn (adsb_push_async)
    for name, kw, mode in (("adsb_push, pageable memory, 1 Mi samples per call", dict(collect_stats=True), "sync"),
                           ("adsb_push with push_overlap", dict(collect_stats=True, push_overlap=True), "overlap"),
                           ("adsb_push_async, two page-locked buffers", dict(collect_stats=True), "async")):
        d = capi.Decoder(**kw)
        part = x[: 16 << 20]
        for _ in range(3):
            d.decode(part, chunk=1 << 20, mode=mode)
        (r0, b0), f0, t0, it = rss_mb(), torch.cuda.mem_get_info()[0], time.time(), 0
        while time.time() - t0 < a.seconds:
            d.decode(part, chunk=1 << 20, mode=mode)
            it += 16
        r1, b1 = rss_mb()
        legs.append((name + " (calls)", it, r1 - r0, (f0 - torch.cuda.mem_get_info()[0]) / 2 ** 20, b1 - b0))
        d.close()
    md = sharding.MultiDecoder(4, [0, 0, 0, 0], collect_stats=True)
    with capi.PinnedBuffers(1, n) as bufs:
        bufs[0][:] = x
        for _ in range(10):
            md.decode_host(bufs[0])
        (r0, b0), f0, t0, it = rss_mb(), torch.cuda.mem_get_info()[0], time.time(), 0
        while time.time() - t0 < a.seconds:
            for _ in range(20):
                md.decode_host(bufs[0])
            it += 20
        r1, b1 = rss_mb()
        legs.append(("multi driver, 4 handles, host-fed, stats", it, r1 - r0, (f0 - torch.cuda.mem_get_info()[0]) / 2 ** 20, b1 - b0))
    md.close()
    for name, it, drss, dvram, dbig in legs:
        print(f"{name}: {it} decodes, resident host memory in mappings < 100 MiB {drss:+.1f} MiB, device memory in use {dvram:+.1f} MiB, "
              f"runtime pool chunks {dbig:+d}")
    bad = [l for l in legs if l[2] > 32 or l[3] > 64]
    print("soak ok" if not bad else f"GROWTH: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
