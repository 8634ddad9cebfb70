#!/usr/bin/env python3
"""A/B of several builds of the library IN ONE PROCESS, launches interleaved.

    python tools/ab_interleaved.py [--dense10] [--rounds 40] [--steps 10] name=path/to/libadsbdec_amd.so ...

Separate processes on the same box differ by 2-3 % from run to run (clock governor, which CUs a process gets first): more
than most of the changes round 6 tried.  Here every build is dlopen'ed side by side (they share the one HIP runtime), each
gets its own handle, and the rounds go A B C A B C ..: `--steps` decodes of the same device-resident capture per build and
round, timed by the kernel's own clock (adsb_profile.big_ms / big_launches).  Prints the median, the quartiles and the
per-round ratio to the first build.
"""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from adsbdec_amd import capi  # noqa: E402
from bench import make_dense, make_dense10, make_gate_storm, make_workload  # noqa: E402


def bind(path):
    L = C.CDLL(path)
    for name in ("adsb_config_init", "adsb_create", "adsb_destroy", "adsb_decode_device", "adsb_get_profile_sized", "adsb_last_error"):
        res, args = capi.SYMBOLS[name]
        try:
            fn = getattr(L, name)
        except AttributeError:  # a build of rounds 4-5: adsb_get_profile(handle, struct) -- it fills the members it knows
            fn = L.adsb_get_profile
            res, args = C.c_int, [C.c_void_p, C.POINTER(capi.Profile)]
            L.adsb_get_profile_sized = lambda h, p, n, fn=fn: fn(h, p)
        fn.restype, fn.argtypes = res, args
    return L


def main():
    args = sys.argv[1:]
    dense10 = "--dense10" in args
    storm = "--gate-storm" in args
    noise = "--dense" in args
    stats = "--stats" in args
    rounds = int(args[args.index("--rounds") + 1]) if "--rounds" in args else 40
    steps = int(args[args.index("--steps") + 1]) if "--steps" in args else 10
    builds = []  # name=path[@passes=N[,big_tiles=M]]: the same library under several adsb_debug_config settings
    for a in args:
        if "=" in a and not a.startswith("--"):
            name, rest = a.split("=", 1)
            path, _, knobs = rest.partition("@")
            builds.append((name, path, dict(kv.split("=") for kv in knobs.split(",") if kv)))
    n = int(args[args.index("--samples") + 1]) if "--samples" in args else 256 << 20
    n -= n % 28
    x = make_dense10(torch, n, 101) if dense10 else make_gate_storm(torch, n, 102) if storm else make_dense(torch, n, 100) if noise else make_workload(torch, n, seed=1)[0]
    torch.cuda.synchronize()
    hs = []
    keep = []
    for name, path, knobs in builds:
        L = bind(os.path.abspath(path))
        L.adsb_abi_version.restype = C.c_int
        old_abi = L.adsb_abi_version() < 5  # a build of rounds 4-5: its struct (no `abi` member, debug_* members inside)
        cfg = capi.ConfigV4() if old_abi else capi.Config()
        L.adsb_config_init(C.byref(cfg), C.sizeof(cfg))
        for k in [k for k in knobs if k.startswith("cfg.")]:  # cfg.host_threads=2: a member of adsb_config itself
            setattr(cfg, k[4:], int(knobs.pop(k)))
        if knobs and old_abi:
            for k, v in knobs.items():
                setattr(cfg, "debug_" + k, int(v))
        elif knobs:
            dbg = capi.DebugConfig()
            dbg.struct_size = C.sizeof(dbg)
            for k, v in knobs.items():
                setattr(dbg, k, int(v))
            keep.append(dbg)
            cfg.debug = C.cast(C.pointer(dbg), C.c_void_p)
        cfg.df18 = 1 if (dense10 or storm or noise) else 0
        cfg.collect_stats = 1 if stats else 0
        cfg.profile = 1
        h = L.adsb_create(C.byref(cfg))
        if not h:
            raise SystemExit(f"{name}: adsb_create failed")
        hs.append((name, L, h))
    out = C.POINTER(capi.Frame)()

    def kernel_ms(L, h):
        p = capi.Profile()
        L.adsb_get_profile_sized(h, C.byref(p), C.sizeof(p))
        return p.big_ms, p.big_launches

    frames = {}
    for name, L, h in hs:  # warm-up, and every build must find the same frames
        for _ in range(5):
            frames[name] = L.adsb_decode_device(h, x.data_ptr(), x.numel(), C.byref(out))
    if len(set(frames.values())) != 1:
        raise SystemExit(f"the builds disagree: {frames}")
    per = {name: [] for name, _, _ in hs}
    wall = {name: [] for name, _, _ in hs}  # the call as the caller sees it: adsb_decode_device back to back
    for r in range(rounds):
        order = hs[r % len(hs):] + hs[:r % len(hs)]
        for name, L, h in order:
            m0, l0 = kernel_ms(L, h)
            t0 = time.perf_counter()
            for _ in range(steps):
                L.adsb_decode_device(h, x.data_ptr(), x.numel(), C.byref(out))
            wall[name].append((time.perf_counter() - t0) * 1e3 / steps)
            m1, l1 = kernel_ms(L, h)
            per[name].append((m1 - m0) / max(1, l1 - l0))
    base = per[hs[0][0]]
    print(f"{n} samples; workload {'dense10' if dense10 else 'gate_storm' if storm else 'noise 7 %' if noise else 'sparse'}{' +stats' if stats else ''}, {frames[hs[0][0]]} frames, "
          f"{rounds} rounds x {steps} launches per build, kernel clock, ms per launch")
    for name, _, _ in hs:
        v = sorted(per[name])
        q = statistics.quantiles(v, n=4)
        ratio = statistics.median(a / b for a, b in zip(per[name], base))
        wr = statistics.median(a / b for a, b in zip(wall[name], wall[hs[0][0]]))
        print(f"  {name:10s} median {statistics.median(v):.5f}  quartiles {q[0]:.5f} .. {q[2]:.5f}  min {v[0]:.5f}  ratio to {hs[0][0]} (median of rounds) {ratio:.4f}"
              f"  | call {statistics.median(wall[name]):.5f} ms, ratio {wr:.4f}")
    for name, L, h in hs:
        L.adsb_destroy(h)


if __name__ == "__main__":
    main()
