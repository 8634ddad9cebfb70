"""ctypes caller of the library's multi-GPU driver (adsb_multi_*, csrc/multi.cpp): ONE process, a worker thread and a
decoder handle per device, shards of one capture (BASELINE configs[4]) or independent captures (configs[3]).

Nothing is orchestrated here: planning, the per-device copy / scan / resolve pipeline, the stitcher, the fallback and the
gather are C++ behind the C-ABI; this file only marshals arguments so that tests/ and bench.py exercise that driver.
(Rounds 2-3 had a Python twin of it over torch.distributed and a shared-memory board; it is gone.)

    md = MultiDecoder(n_devices=8, df18=True, collect_stats=True)
    frames, n = md.decode_host(ptr, n_samples)      # one page-locked capture -> frames in the reference's order
    md.stats(), md.info()
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class ShardError(capi.AdsbError):
    pass


class MultiDecoder:
    def __init__(self, n_devices: int = 1, devices=None, **cfg_kw):
        L = capi.load()
        self._L = L
        cfg = capi.make_config(**cfg_kw)
        self._fix = bool(cfg_kw.get("fix_1bit"))
        devs = None
        if devices is not None:
            assert len(devices) == n_devices
            devs = (C.c_int * n_devices)(*devices)
        self._h = L.adsb_multi_create(C.byref(cfg), n_devices, devs)
        if not self._h:
            raise ShardError("adsb_multi_create failed: " + (L.adsb_multi_last_error(None) or b"").decode())
        self.n_devices = n_devices

    def _err(self, what):
        return ShardError(f"{what} failed: " + (self._L.adsb_multi_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._L.adsb_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def plan(self, total_samples: int):
        arrs = [(C.c_uint64 * self.n_devices)() for _ in range(4)]
        n = self._L.adsb_multi_plan(self._h, total_samples, *arrs)
        if n < 0:
            raise self._err("adsb_multi_plan")
        return [dict(g_begin=int(arrs[0][i]), g_end=int(arrs[1][i]), first_sample=int(arrs[2][i]), n_samples=int(arrs[3][i]))
                for i in range(n)]

    # -- configs[4]: one capture, sharded.  All three return (pointer to Frame, count), valid until the next call.
    def decode_host(self, x, n: int | None = None):
        """x: a uint16 ndarray, or the address of n samples in (ideally page-locked) host memory."""
        if n is None:
            assert x.dtype == np.uint16 and x.flags["C_CONTIGUOUS"]
            x, n = x.ctypes.data, x.size
        p = C.POINTER(capi.Frame)()
        k = self._L.adsb_multi_decode_host(self._h, x, n, C.byref(p))
        if k < 0:
            raise self._err("adsb_multi_decode_host")
        return p, int(k)

    def decode_file(self, path: str):
        p = C.POINTER(capi.Frame)()
        k = self._L.adsb_multi_decode_file(self._h, path.encode(), C.byref(p))
        if k < 0:
            raise self._err("adsb_multi_decode_file")
        return p, int(k)

    def decode_device(self, total_samples: int, slice_ptrs):
        """slice_ptrs[i]: device address of the samples plan(total_samples)[i] names, resident on worker i's device."""
        arr = (C.c_void_p * len(slice_ptrs))(*slice_ptrs)
        p = C.POINTER(capi.Frame)()
        k = self._L.adsb_multi_decode_device(self._h, total_samples, arr, len(slice_ptrs), C.byref(p))
        if k < 0:
            raise self._err("adsb_multi_decode_device")
        return p, int(k)

    def stats(self):
        st = capi.Stats()
        if self._L.adsb_multi_get_stats(self._h, C.byref(st)) != 0:
            raise ShardError("no statistics: create the MultiDecoder with collect_stats=True and decode first")
        return capi._stats_to_dict(st, self._fix)

    def info(self):
        i = capi.MultiInfo()
        self._L.adsb_multi_get_info(self._h, C.byref(i))
        return {k: getattr(i, k) for k, _ in capi.MultiInfo._fields_}

    def host_alloc(self, total_samples: int):
        """adsb_multi_host_alloc: a page-locked uint16 array for ONE capture, every shard's part on the NUMA node of the device
        that will pull it.  Returns (ndarray view, address); release with host_free(address)."""
        addr = self._L.adsb_multi_host_alloc(self._h, total_samples)
        if not addr:
            raise self._err("adsb_multi_host_alloc")
        buf = (C.c_uint16 * total_samples).from_address(addr)
        return np.frombuffer(buf, dtype=np.uint16), addr

    def host_free(self, addr: int):
        self._L.adsb_host_free(addr)

    def placement(self, worker: int):
        """adsb_multi_worker_placement: where worker's slice of the last host-fed capture lives, and where its device is."""
        pl = capi.WorkerPlacement()
        if self._L.adsb_multi_worker_placement(self._h, worker, C.byref(pl)) != 0:
            raise self._err("adsb_multi_worker_placement")
        return {k: getattr(pl, k) for k, _ in capi.WorkerPlacement._fields_}

    def worker_profile(self, worker: int):
        p = capi.Profile()
        if self._L.adsb_multi_worker_profile_sized(self._h, worker, C.byref(p), C.sizeof(p)) != 0:
            raise self._err("adsb_multi_worker_profile")
        return {k: getattr(p, k) for k, _ in capi.Profile._fields_}

    # -- configs[3]: independent captures, one stream each
    def decode_streams_host(self, xs):
        """xs: uint16 ndarrays, or (address, n) pairs."""
        pairs = [(x.ctypes.data, x.size) if not isinstance(x, tuple) else x for x in xs]
        ptrs = (C.c_void_p * len(pairs))(*[p for p, _ in pairs])
        lens = (C.c_size_t * len(pairs))(*[n for _, n in pairs])
        if self._L.adsb_multi_decode_streams_host(self._h, len(pairs), ptrs, lens) != 0:
            raise self._err("adsb_multi_decode_streams_host")

    def decode_streams_file(self, paths):
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        if self._L.adsb_multi_decode_streams_file(self._h, len(paths), arr) != 0:
            raise self._err("adsb_multi_decode_streams_file")

    def stream_frames(self, s: int):
        p = C.POINTER(capi.Frame)()
        k = self._L.adsb_multi_stream_frames(self._h, s, C.byref(p))
        if k < 0:
            raise self._err("adsb_multi_stream_frames")
        return p, int(k)

    def stream_stats(self, s: int):
        st = capi.Stats()
        if self._L.adsb_multi_stream_stats(self._h, s, C.byref(st)) != 0:
            raise self._err("adsb_multi_stream_stats")
        return capi._stats_to_dict(st, self._fix)
