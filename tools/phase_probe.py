"""Where does a TILE's life go?  Per-phase device-clock stamps of adsb::scan_kernel on the sparse headline capture and on
the three dense ones (noise 7 %, BASELINE configs[2] at 10 %, gate storm), plain and with the Try/Ok table.

Needs the stamps build of the library (never the shipped one):
    tools/build_variant.sh stamps -DADSB_PHASE_STAMPS
    ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_var/stamps/libadsbdec_amd.so python tools/phase_probe.py > gpurun_out/r6_phase_stamps.txt
Thread 0 of every tile adds the ticks of the device's 100 MHz clock between phase boundaries to one accumulator per phase
(scan_kernel.hip, ADSB_STAMP); this prints the mean per tile in microseconds.  The stamps cost a few per cent themselves:
the figures are shares of a tile's life, not the shipped kernel's absolute times (kernel_ms of the same build is printed).
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

from adsbdec_amd import capi  # noqa: E402
from bench import bind_near_gpu, make_dense, make_dense10, make_gate_storm, make_workload  # noqa: E402

PHASES = ["tiles", "stage_a", "gate+queue", "slicer+crc(+tries)", "filter+rank", "order+links", "finish(bytes,pw,reserve)",
          "records+store", "rounds", "survivors", "staged", "records", "kept", "marker", "longest_tile_ticks"]


def main():
    torch.cuda.set_device(0)
    bind_near_gpu(torch, 0)
    lib = capi.load()
    if not hasattr(lib, "adsb_debug_phase_read"):
        raise SystemExit("this library has no phase stamps: build with tools/build_variant.sh stamps -DADSB_PHASE_STAMPS and set ADSB_LIB_PATH")
    rd = lib.adsb_debug_phase_read
    rd.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
    n = 256 << 20
    n -= n % 28
    steps = 20
    for name, make, df18 in (("sparse (configs[1])", lambda: make_workload(torch, n, seed=1)[0], False),
                             ("noise 7 %", lambda: make_dense(torch, n, 100), True),
                             ("dense10 (configs[2])", lambda: make_dense10(torch, n, 101), True),
                             ("gate_storm", lambda: make_gate_storm(torch, n, 102), True)):
        if os.environ.get("PHASE_ONLY") and os.environ["PHASE_ONLY"] not in name:
            continue
        x = make()
        torch.cuda.synchronize()
        for stats in ((False,) if os.environ.get("PHASE_ONLY") else (False, True)):
            d = capi.Decoder(df18=df18, profile=True, collect_stats=stats)
            for _ in range(5):
                d.decode_device_raw(x.data_ptr(), x.numel())
            torch.cuda.synchronize()
            buf = (C.c_ulonglong * 16)()
            rd(buf, 16, 1)
            p0 = d.profile()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = d.decode_device_raw(x.data_ptr(), x.numel())
            dt = (time.perf_counter() - t0) / steps * 1e3
            torch.cuda.synchronize()
            p1 = d.profile()
            rd(buf, 16, 1)
            v = [int(buf[i]) for i in range(16)]
            tiles = max(1, v[0])
            print(f"== {name}, collect_stats={int(stats)}: step {dt:.4f} ms, kernel {(p1['kernel_ms'] - p0['kernel_ms']) / steps:.4f} ms, "
                  f"frames {r[1]}, tiles/launch {tiles // steps}")
            total = 0.0
            for i in (1, 2, 3, 4, 5, 6, 7, 13):
                us = v[i] / tiles / 100.0
                total += us
                print(f"   {PHASES[i]:28s} {us:8.2f} us per tile")
            print(f"   {'sum':28s} {total:8.2f} us per tile; longest tile {v[14] / 100.0:.1f} us")
            print(f"   per tile: rounds {v[8] / tiles:.2f}, survivors {v[9] / tiles:.1f}, staged CRC-valid {v[10] / tiles:.1f}, "
                  f"kept {v[12] / tiles:.1f}, records {v[11] / tiles:.1f}", flush=True)
            d.close()
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
