"""Why does the C host program push a 32 MiB buffer in 2.2 ms when bench.py's e2e_host_fed leg pushes one in 0.63?
adsb_push_async over 16 buffers of 16 Mi samples each, every buffer used ONCE (the host program's case) or twice, from
hipHostMalloc memory / registered huge-page memory / registered 4 KiB-page memory, with and without the Try/Ok table."""
import ctypes as C, mmap, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from adsbdec_amd import capi
from bench import bind_near_gpu
torch.cuda.set_device(0); bind_near_gpu(torch, 0)
L = capi.load()
N, NB = 16 << 20, 16
libc = C.CDLL(None)
libc.posix_memalign.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t]
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]

def buffers(kind):
    out = []
    for _ in range(NB):
        if kind == "hipHostMalloc":
            p = L.adsb_host_alloc(2 * N)
        else:
            q = C.c_void_p()
            assert libc.posix_memalign(C.byref(q), 2 << 20, 2 * N) == 0
            libc.madvise(q, 2 * N, 14 if kind == "registered, huge pages" else 15)   # MADV_HUGEPAGE / MADV_NOHUGEPAGE
            p = q.value
        a = np.frombuffer((C.c_uint16 * N).from_address(p), dtype=np.uint16)
        a[:] = 2048
        a[::5000] = 2300
        if kind != "hipHostMalloc":
            t0 = time.perf_counter()
            assert L.adsb_host_register(p, 2 * N) == 0
        out.append((p, a))
    return out

for kind in ("hipHostMalloc", "registered, huge pages", "registered, 4 KiB pages"):
    for stats in (False, True):
        bufs = buffers(kind)
        d = capi.Decoder(df18=False, collect_stats=stats)
        # (the first launch of a process loads the code object: take it out of the picture)
        d.push_async((bufs[0][0], 1 << 20)); d.finish(); d.drain(); d.reset()
        for rnd in range(2):
            t0 = time.perf_counter()
            per = []
            for p, _ in bufs:
                t1 = time.perf_counter()
                d.push_async((p, N))
                per.append((time.perf_counter() - t1) * 1e3)
            d.finish()
            dt = (time.perf_counter() - t0) * 1e3
            d.drain(); d.reset()
            print(f"{kind:26s} stats={int(stats)} round {rnd}: {NB} pushes of 32 MiB in {dt:6.1f} ms = {NB * N / dt / 1e6:5.1f} GS/s; per push "
                  f"{min(per):.2f} .. {max(per):.2f} ms, first {per[0]:.2f}", flush=True)
        d.close()
