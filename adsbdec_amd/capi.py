"""ctypes binding of include/adsbdec_amd.h.

This is plumbing for tests/ and bench.py: the product's host side is C
(csrc/cli/adsbdec_amd_cli.c) and everything of substance lives behind the C-ABI.
There is no Python or CPU fallback: if the shared library is missing or no
gfx950 device is present, calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADSB_LIB_PATH") or os.path.join(PKG, "lib", "libadsbdec_amd.so")  # override: A/B runs
CLI_PATH = os.path.join(PKG, "lib", "adsbdec_amd_cli")


class AdsbError(RuntimeError):
    pass


class Frame(C.Structure):
    _fields_ = [("g", C.c_uint64), ("ts", C.c_uint64), ("pw", C.c_uint32), ("len", C.c_uint8),
                ("frame", C.c_uint8 * 14), ("reserved", C.c_uint8)]


class Candidate(C.Structure):
    _fields_ = [("g", C.c_uint64), ("pw", C.c_uint32), ("len", C.c_uint8),
                ("frame", C.c_uint8 * 14), ("reserved", C.c_uint8)]


class Stats(C.Structure):
    _fields_ = [("try_", C.c_uint64 * 3), ("ok", C.c_uint64 * 3), ("fixed", C.c_uint64)]


class ShardHead(C.Structure):
    _fields_ = [("g_begin", C.c_uint64), ("g_end", C.c_uint64), ("n_frames", C.c_uint64), ("n_head", C.c_uint64),
                ("head_end", C.c_uint64), ("skipped", C.c_uint64), ("status", C.c_uint64), ("n_bases", C.c_uint64),
                ("walk_final", C.c_uint64), ("has_tries", C.c_uint64), ("tries", C.c_uint64 * 3), ("ok", C.c_uint64 * 3),
                ("fixed", C.c_uint64)]


class ShardPart(C.Structure):
    _fields_ = [("head", C.POINTER(ShardHead)), ("frames", C.POINTER(Frame)), ("head_cands", C.POINTER(Candidate)),
                ("bases", C.POINTER(C.c_uint64)),
                ("head_tries", C.POINTER(C.c_uint64)), ("n_head_tries", C.c_uint64), ("head_tries_end", C.c_uint64),
                ("tail_tries", C.POINTER(C.c_uint64)), ("n_tail_tries", C.c_uint64), ("tail_from", C.c_uint64)]


class ShardFix(C.Structure):
    _fields_ = [("new_first", C.c_uint64), ("n_new", C.c_uint64), ("drop_front", C.c_uint64), ("keep", C.c_uint64),
                ("ts_sub", C.c_int64)]


class Config(C.Structure):
    """adsb_config, ABI 5 (include/adsbdec_amd.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("abi", C.c_uint32), ("df18", C.c_int32), ("device", C.c_int32),
                ("collect_stats", C.c_int32), ("profile", C.c_int32), ("stage_samples", C.c_uint64), ("stream", C.c_void_p),
                ("all_candidates", C.c_int32), ("fix_1bit", C.c_int32), ("push_overlap", C.c_int32), ("host_threads", C.c_int32),
                ("wait_timeout_s", C.c_int32), ("warm_start", C.c_int32), ("debug", C.c_void_p)]


class DebugConfig(C.Structure):
    """adsb_debug_config (include/adsbdec_amd_diag.h): the test knobs behind adsb_config.debug."""
    _fields_ = [("struct_size", C.c_uint32), ("queue_cap", C.c_int32), ("cand_cap", C.c_int32), ("try_cap", C.c_int32),
                ("clist_cap", C.c_int32), ("no_streaming", C.c_int32), ("frames_cap", C.c_int32), ("reader_min_tiles", C.c_int32),
                ("shard_head", C.c_int32), ("passes", C.c_int32), ("big_tiles", C.c_int32), ("gang_min", C.c_int32)]


class ConfigV4(C.Structure):
    """adsb_config as ABI 4 had it (rounds 4-5): only for same-box A/B runs against an older build of the library
    (ADSB_LIB_PATH); the tree's own library refuses it."""
    _fields_ = [("struct_size", C.c_uint32), ("df18", C.c_int32), ("device", C.c_int32),
                ("collect_stats", C.c_int32), ("profile", C.c_int32), ("debug_queue_cap", C.c_int32),
                ("stage_samples", C.c_uint64), ("stream", C.c_void_p), ("all_candidates", C.c_int32),
                ("fix_1bit", C.c_int32), ("debug_cand_cap", C.c_int32), ("debug_try_cap", C.c_int32),
                ("debug_clist_cap", C.c_int32), ("push_overlap", C.c_int32),
                ("host_threads", C.c_int32), ("debug_no_streaming", C.c_int32), ("debug_frames_cap", C.c_int32),
                ("debug_reader_min_tiles", C.c_int32), ("debug_shard_head", C.c_int32), ("debug_passes", C.c_int32),
                ("debug_stagger", C.c_int32), ("wait_timeout_s", C.c_int32), ("debug_gang_min", C.c_int32)]


class MultiInfo(C.Structure):
    _fields_ = [("shards", C.c_int32), ("fallback", C.c_int32), ("calls_walked", C.c_uint64), ("calls_jumped", C.c_uint64),
                ("create_ms", C.c_double), ("workers_ms", C.c_double), ("stitch_us", C.c_double), ("serial_us", C.c_double),
                ("total_ms", C.c_double), ("workers_bound", C.c_int32), ("helper_threads", C.c_int32)]


class WorkerPlacement(C.Structure):
    _fields_ = [("device", C.c_int32), ("device_node", C.c_int32), ("thread_bound", C.c_int32), ("slice_node", C.c_int32),
                ("local_fraction", C.c_double)]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("relaunches", C.c_uint64), ("offsets", C.c_uint64),
                ("kernel_ms", C.c_double), ("last_kernel_ms", C.c_double), ("last_offsets", C.c_uint64),
                ("candidates", C.c_uint64), ("tries", C.c_uint64), ("host_ms", C.c_double), ("wait_ms", C.c_double), ("big_offsets", C.c_uint64),
                ("big_launches", C.c_uint64), ("big_ms", C.c_double),
                ("host_threads_running", C.c_uint32), ("gang_launches", C.c_uint32), ("gang_batches", C.c_uint64)]


# every symbol include/adsbdec_amd.h and include/adsbdec_amd_diag.h declare: (restype, argtypes)
SYMBOLS = {
    "adsb_abi_version": (C.c_int, []),
    "adsb_config_default": (None, [C.c_void_p]),          # (the symbol binaries of ABI <= 4 call: leaves a struct adsb_create refuses)
    "adsb_config_init": (None, [C.c_void_p, C.c_size_t]),
    "adsb_create": (C.c_void_p, [C.c_void_p]),
    "adsb_destroy": (None, [C.c_void_p]),
    "adsb_reset": (C.c_int, [C.c_void_p]),
    "adsb_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "adsb_push_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "adsb_sync": (C.c_int, [C.c_void_p]),
    "adsb_device_cpulist": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "adsb_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "adsb_host_unregister": (C.c_int, [C.c_void_p]),
    "adsb_push_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "adsb_push_device_final": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "adsb_decode_device": (C.c_long, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(Frame))]),
    "adsb_finish": (C.c_int, [C.c_void_p]),
    "adsb_host_alloc": (C.c_void_p, [C.c_size_t]),
    "adsb_host_free": (None, [C.c_void_p]),
    "adsb_drain": (C.c_long, [C.c_void_p, C.POINTER(Frame), C.c_size_t]),
    "adsb_take": (C.c_long, [C.c_void_p, C.POINTER(C.POINTER(Frame))]),
    "adsb_pending": (C.c_size_t, [C.c_void_p]),
    "adsb_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "adsb_get_profile_sized": (C.c_int, [C.c_void_p, C.POINTER(Profile), C.c_size_t]),
    "adsb_last_error": (C.c_char_p, [C.c_void_p]),
    "adsb_format_frame": (C.c_int, [C.POINTER(Frame), C.c_int, C.c_char_p]),
    "adsb_resolver_create": (C.c_void_p, []),
    "adsb_resolver_destroy": (None, [C.c_void_p]),
    "adsb_resolver_set_threads": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "adsb_resolver_feed": (C.c_int, [C.c_void_p, C.POINTER(Candidate), C.c_size_t,
                                     C.POINTER(C.c_uint64), C.c_size_t]),
    "adsb_resolver_advance": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "adsb_resolver_drain": (C.c_long, [C.c_void_p, C.POINTER(Frame), C.c_size_t]),
    "adsb_resolver_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "adsb_handoff_walk": (C.c_long, [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "adsb_shard_layout_check": (C.c_int, [C.c_size_t, C.c_size_t]),
    "adsb_host_cpu_refusal": (C.c_char_p, []),
    "adsb_device_numa_node": (C.c_int, [C.c_int]),
    "adsb_host_alloc_on": (C.c_void_p, [C.c_size_t, C.c_int]),
    "adsb_host_alloc_sharded": (C.c_void_p, [C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "adsb_host_placement": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "adsb_host_release_mapped": (C.c_int, [C.c_void_p]),
    "adsb_multi_host_alloc": (C.c_void_p, [C.c_void_p, C.c_uint64]),
    "adsb_multi_worker_placement": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(WorkerPlacement)]),
    "adsb_resolver_advance_stream": (C.c_long, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64,
                                                C.c_uint64, C.c_uint64, C.c_int]),
    "adsb_scan_shard_resolved": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64, C.c_uint64,
                                           C.POINTER(ShardHead), C.POINTER(Frame), C.c_size_t, C.POINTER(Candidate),
                                           C.c_size_t]),
    "adsb_scan_shard_resolved_walk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64,
                                                C.POINTER(ShardHead), C.POINTER(Frame), C.c_size_t, C.POINTER(Candidate),
                                                C.c_size_t, C.POINTER(C.c_uint64), C.c_size_t]),
    "adsb_scan_shard_resolved_take": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64,
                                                C.POINTER(ShardHead), C.POINTER(C.POINTER(Frame)), C.POINTER(C.POINTER(Candidate)),
                                                C.POINTER(C.c_uint64), C.c_size_t]),
    "adsb_stitch_shards": (C.c_int, [C.POINTER(ShardPart), C.c_int, C.c_uint64, C.POINTER(ShardFix), C.POINTER(Frame),
                                     C.c_size_t, C.POINTER(C.c_size_t)]),
    "adsb_stitch_shards_ex": (C.c_int, [C.POINTER(ShardPart), C.c_int, C.c_uint64, C.POINTER(ShardFix), C.POINTER(Frame),
                                        C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]),
    "adsb_stitch_shards_stats": (C.c_int, [C.POINTER(ShardPart), C.c_int, C.c_uint64, C.POINTER(ShardFix), C.POINTER(Frame),
                                           C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(Stats)]),
    "adsb_shard_begin": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.c_size_t]),
    "adsb_shard_end": (C.c_int, [C.c_void_p, C.POINTER(ShardHead), C.POINTER(C.POINTER(Frame)), C.POINTER(C.POINTER(Candidate))]),
    "adsb_shard_walk": (C.c_size_t, [C.POINTER(ShardHead), C.POINTER(Frame), C.c_uint64, C.POINTER(C.c_uint64), C.c_size_t]),
    "adsb_shard_apply_fix": (None, [C.POINTER(Frame), C.c_size_t, C.c_int64]),
    "adsb_resolver_start_chain": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64]),
    "adsb_resolver_start_walk": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.c_size_t]),
    "adsb_resolver_walk_result": (C.c_size_t, [C.c_void_p, C.POINTER(C.c_int)]),
    "adsb_resolver_head": (C.c_long, [C.c_void_p, C.POINTER(Candidate), C.c_size_t]),
    "adsb_resolver_skipped": (C.c_uint64, [C.c_void_p]),
    "adsb_plan_shards": (C.c_int, [C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "adsb_scan_shard_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64,
                                       C.c_uint64, C.POINTER(Candidate), C.c_size_t,
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_size_t,
                                       C.POINTER(C.c_size_t)]),
    "adsb_multi_create": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "adsb_multi_destroy": (None, [C.c_void_p]),
    "adsb_multi_devices": (C.c_int, [C.c_void_p]),
    "adsb_multi_decode_host": (C.c_long, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(Frame))]),
    "adsb_multi_decode_file": (C.c_long, [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(Frame))]),
    "adsb_multi_decode_device": (C.c_long, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.POINTER(Frame))]),
    "adsb_multi_plan": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "adsb_multi_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "adsb_multi_decode_streams_host": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "adsb_multi_decode_streams_file": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]),
    "adsb_multi_stream_frames": (C.c_long, [C.c_void_p, C.c_int, C.POINTER(C.POINTER(Frame))]),
    "adsb_multi_stream_stats": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Stats)]),
    "adsb_multi_get_info": (C.c_int, [C.c_void_p, C.POINTER(MultiInfo)]),
    "adsb_multi_worker_profile_sized": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Profile), C.c_size_t]),
    "adsb_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "adsb_scan_shard": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_size_t, C.c_uint64,
                                  C.c_uint64, C.POINTER(Candidate), C.c_size_t,
                                  C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.c_size_t,
                                  C.POINTER(C.c_size_t)]),
}

_lib = None


def load():
    """dlopen the in-tree library and bind every declared symbol. Raises if absent.
    In a process that also uses PyTorch, import torch BEFORE calling this: torch ships its own
    libamdhip64, and two HIP runtimes in one process do not share the device (the second one
    reports "no ROCm-capable device")."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AdsbError(f"{LIB_PATH} is missing: run `python -m adsbdec_amd._build` "
                            "(there is no fallback implementation)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(L, name)
            except AttributeError:
                if os.environ.get("ADSB_LIB_PATH"):  # an A/B run against an older build of the library: it has what it has
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _frames_to_dicts(arr, n):
    return [dict(g=int(f.g), ts=int(f.ts), pw=int(f.pw), frame=bytes(f.frame[: f.len])) for f in arr[:n]]


def _stats_to_dict(st: Stats, with_fixed: bool = False):
    d = {"try": {11: int(st.try_[0]), 17: int(st.try_[1]), 18: int(st.try_[2])},
         "ok": {11: int(st.ok[0]), 17: int(st.ok[1]), 18: int(st.ok[2])}}
    if with_fixed:
        d["fixed"] = int(st.fixed)
    return d


def format_frame(fr: dict, outformat: int) -> bytes:
    f = Frame()
    f.g, f.ts, f.pw, f.len = fr.get("g", 0), fr["ts"], fr["pw"], len(fr["frame"])
    for i, b in enumerate(fr["frame"]):
        f.frame[i] = b
    buf = C.create_string_buffer(256)
    n = load().adsb_format_frame(C.byref(f), outformat, buf)
    return buf.raw[:n]


DEBUG_KNOBS = ("queue_cap", "cand_cap", "try_cap", "clist_cap", "no_streaming", "frames_cap", "reader_min_tiles", "shard_head",
               "passes", "big_tiles", "gang_min")


def make_config(df18: bool = False, device: int = -1, collect_stats: bool = False,
                profile: bool = False, stage_samples: int = 0, stream: int | None = None,
                all_candidates: bool = False, fix_1bit: bool = False, push_overlap: bool = False,
                host_threads: int = 0, wait_timeout_s: int = 0, warm_start: bool = False, **debug):
    """adsb_config from keywords (adsb_config_default + the members named).  Keywords debug_<knob> (DEBUG_KNOBS) fill an
    adsb_debug_config that the returned struct points at (and keeps alive: cfg._debug)."""
    L = load()
    unknown = [k for k in debug if not k.startswith("debug_") or k[6:] not in DEBUG_KNOBS]
    if unknown:
        raise TypeError(f"make_config: unknown keyword(s) {unknown}")
    if L.adsb_abi_version() < 5:   # an A/B run against a build of rounds 4-5 (ADSB_LIB_PATH): its struct, its member names
        cfg = ConfigV4()
        L.adsb_config_init(C.byref(cfg), C.sizeof(cfg))
        for k, v in debug.items():
            if hasattr(ConfigV4, k) and cfg.struct_size >= getattr(ConfigV4, k).offset + 4:
                setattr(cfg, k, int(v))
        if cfg.struct_size >= ConfigV4.wait_timeout_s.offset + 4:
            cfg.wait_timeout_s = wait_timeout_s
    else:
        cfg = Config()
        L.adsb_config_init(C.byref(cfg), C.sizeof(cfg))
        cfg.wait_timeout_s = wait_timeout_s
        cfg.warm_start = int(warm_start)
        if any(debug.values()):
            dbg = DebugConfig()
            dbg.struct_size = C.sizeof(dbg)
            for k, v in debug.items():
                setattr(dbg, k[6:], int(v))
            cfg._debug = dbg                      # (the library copies it in adsb_create; until then it must live)
            cfg.debug = C.addressof(dbg)
    cfg.df18 = int(df18)
    cfg.device = device
    cfg.collect_stats = int(collect_stats)
    cfg.profile = int(profile)
    cfg.stage_samples = stage_samples
    cfg.stream = stream
    cfg.all_candidates = int(all_candidates)
    cfg.fix_1bit = int(fix_1bit)
    cfg.push_overlap = int(push_overlap)
    cfg.host_threads = int(host_threads)
    return cfg


class Decoder:
    """One stream (== the statics of air.c / demod.c / valid.c).  Keywords: make_config."""

    def __init__(self, **cfg_kw):
        L = load()
        cfg = make_config(**cfg_kw)
        self._L = L
        self._fix = bool(cfg_kw.get("fix_1bit"))
        self._h = L.adsb_create(C.byref(cfg))
        if not self._h:
            raise AdsbError("adsb_create failed: " + (L.adsb_last_error(None) or b"").decode())
        self._out = C.POINTER(Frame)()          # decode_device_raw's result pointer and its reference, made once
        self._out_ref = C.byref(self._out)
        self._decode_device = L.adsb_decode_device

    def _check(self, rc, what):
        if rc != 0:
            raise AdsbError(f"{what} failed: " + (self._L.adsb_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._L.adsb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._check(self._L.adsb_reset(self._h), "adsb_reset")

    def push(self, x: np.ndarray):
        x = np.ascontiguousarray(x)
        assert x.dtype == np.uint16
        self._check(self._L.adsb_push(self._h, x.ctypes.data, x.size), "adsb_push")

    def push_async(self, x):
        """adsb_push_async: x (ndarray or (ptr, n)) stays borrowed until the next push/finish/sync returns."""
        if isinstance(x, tuple):
            ptr, n = x
        else:
            assert x.dtype == np.uint16 and x.flags["C_CONTIGUOUS"]
            ptr, n = x.ctypes.data, x.size
        self._check(self._L.adsb_push_async(self._h, ptr, n), "adsb_push_async")

    def sync(self):
        self._check(self._L.adsb_sync(self._h), "adsb_sync")

    def push_device(self, ptr: int, n: int):
        self._check(self._L.adsb_push_device(self._h, ptr, n), "adsb_push_device")

    def push_device_final(self, ptr: int, n: int):
        self._check(self._L.adsb_push_device_final(self._h, ptr, n), "adsb_push_device_final")

    def decode_device_raw(self, ptr: int, n: int):
        """adsb_decode_device: reset + push_device_final + take in one call -> (Frame pointer, count).  The pointer object is
        the decoder's own, reused by every call (a timed loop pays for the call, not for Python objects): what it points at
        is valid until the next call of this decoder, as adsb_take says."""
        k = self._decode_device(self._h, ptr, n, self._out_ref)
        if k < 0:
            self._check(-1, "adsb_decode_device")
        return self._out, k

    def finish(self):
        self._check(self._L.adsb_finish(self._h), "adsb_finish")

    def pending(self) -> int:
        return int(self._L.adsb_pending(self._h))

    def drain_raw(self, reuse: bool = False):
        """All pending frames as one ctypes Frame array (no per-frame Python work).
        reuse=True hands out the handle's own output array (valid until the next such
        call) instead of allocating -- and page-faulting in -- a fresh one every time."""
        n = self.pending()
        if reuse:
            if getattr(self, "_out_cap", 0) < max(1, n):
                self._out_cap = max(1, n + n // 4)
                self._out_buf = (Frame * self._out_cap)()
            buf = self._out_buf
        else:
            buf = (Frame * max(1, n))()
        got = self._L.adsb_drain(self._h, buf, n) if n else 0
        if got < 0:
            raise AdsbError("adsb_drain failed")
        return buf, int(got)

    def take_raw(self):
        """All pending frames in place (adsb_take): (pointer to Frame, count); valid until the
        next push / finish / reset of this decoder."""
        p = C.POINTER(Frame)()
        n = self._L.adsb_take(self._h, C.byref(p))
        if n < 0:
            raise AdsbError("adsb_take failed")
        return p, int(n)

    def drain(self):
        buf, n = self.drain_raw()
        return _frames_to_dicts(buf, n)

    def stats(self):
        st = Stats()
        self._check(self._L.adsb_get_stats(self._h, C.byref(st)), "adsb_get_stats")
        return _stats_to_dict(st, self._fix)

    def profile(self):
        p = Profile()
        if hasattr(self._L, "adsb_get_profile_sized"):
            self._check(self._L.adsb_get_profile_sized(self._h, C.byref(p), C.sizeof(p)), "adsb_get_profile")
        else:   # (an A/B run against a build of rounds 4-5: its struct is a prefix of this one)
            self._L.adsb_get_profile.argtypes = [C.c_void_p, C.POINTER(Profile)]
            self._check(self._L.adsb_get_profile(self._h, C.byref(p)), "adsb_get_profile")
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    def scan_shard(self, ptr: int, first_sample: int, n: int, g_begin: int, g_end: int,
                   cand_cap: int = 1 << 16, try_cap: int = 1 << 20):
        """Stateless per-shard scan -> (Candidate array, count, tries ndarray)."""
        while True:
            cands = (Candidate * cand_cap)()
            tries = np.empty(try_cap, dtype=np.uint64)
            nc, nt = C.c_size_t(0), C.c_size_t(0)
            rc = self._L.adsb_scan_shard(self._h, ptr, first_sample, n, g_begin, g_end, cands, cand_cap,
                                         C.byref(nc), tries.ctypes.data_as(C.POINTER(C.c_uint64)),
                                         try_cap, C.byref(nt))
            if rc == -2:
                cand_cap, try_cap = max(cand_cap, nc.value), max(try_cap, nt.value)
                continue
            self._check(rc, "adsb_scan_shard")
            return cands, nc.value, tries[: nt.value].copy()

    def decode(self, x: np.ndarray, chunk: int | None = None, mode: str = "sync"):
        """Whole-buffer convenience: push (optionally in chunks), finish, drain.
        mode "async": adsb_push_async from two alternating page-locked buffers, the
        double-buffered read loop of the C host program."""
        self.reset()
        if mode == "async":
            return self._decode_async(x, chunk or x.size)
        if mode == "overlap":
            return self._decode_overlap(x, chunk or x.size)
        if chunk is None:
            self.push(x)
        else:
            for i in range(0, x.size, chunk):
                self.push(x[i:i + chunk])
        self.finish()
        return self.drain()

    def _decode_async(self, x: np.ndarray, chunk: int):
        out = []
        with PinnedBuffers(2, max(1, min(chunk, max(1, x.size)))) as bufs:
            for k, i in enumerate(range(0, x.size, chunk)):
                piece = x[i:i + chunk]
                b = bufs[k % 2][: piece.size]
                b[:] = piece                # the previous push from this buffer was two calls ago: free again
                self.push_async(b)
                out += self.drain()         # frames of the previous piece
            self.finish()
            out += self.drain()
        return out


    def _decode_overlap(self, x: np.ndarray, chunk: int):
        """cfg.push_overlap: adsb_push from ONE page-locked buffer (fileInput's single iqbuff, air.c:230-239) that is
        overwritten the moment the call returns -- the copy must be complete by then; frames follow one call later."""
        out = []
        with PinnedBuffers(1, max(1, min(chunk, max(1, x.size)))) as bufs:
            for i in range(0, x.size, chunk):
                piece = x[i:i + chunk]
                b = bufs[0][: piece.size]
                b[:] = piece
                self._check(self._L.adsb_push(self._h, b.ctypes.data, b.size), "adsb_push")
                b[:] = 0xFFFF               # the buffer is the caller's again
                out += self.drain()
            self.finish()
            out += self.drain()
        return out


class PinnedBuffers:
    """n page-locked uint16 buffers from adsb_host_alloc, as numpy views."""

    def __init__(self, n: int, samples: int):
        L = load()
        self._L, self._ptrs, self.views = L, [], []
        for _ in range(n):
            p = L.adsb_host_alloc(2 * samples)
            if not p:
                raise AdsbError("adsb_host_alloc failed")
            self._ptrs.append(p)
            self.views.append(np.ctypeslib.as_array((C.c_uint16 * samples).from_address(p)))

    def __enter__(self):
        return self.views

    def __exit__(self, *exc):
        self.views = []
        for p in self._ptrs:
            self._L.adsb_host_free(p)
        self._ptrs = []


class Resolver:
    """Host-side greedy resolver handle (usable without a GPU)."""

    def __init__(self):
        self._L = load()
        self._h = self._L.adsb_resolver_create()

    def close(self):
        if self._h:
            self._L.adsb_resolver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def feed(self, cands, tries=None):
        """cands: ctypes Candidate array slice or list of (g, pw, frame-bytes)."""
        if isinstance(cands, list):
            arr = (Candidate * max(1, len(cands)))()
            for i, (g, pw, fr) in enumerate(cands):
                arr[i].g, arr[i].pw, arr[i].len = g, pw, len(fr)
                for k, b in enumerate(fr):
                    arr[i].frame[k] = b
            n = len(cands)
        else:
            arr, n = cands
        t = np.ascontiguousarray(tries if tries is not None else np.empty(0, np.uint64), dtype=np.uint64)
        rc = self._L.adsb_resolver_feed(self._h, arr, n, t.ctypes.data_as(C.POINTER(C.c_uint64)), t.size)
        if rc != 0:
            raise AdsbError("adsb_resolver_feed failed")

    def advance(self, power_samples: int, g_complete: int):
        if self._L.adsb_resolver_advance(self._h, power_samples, g_complete) != 0:
            raise AdsbError("adsb_resolver_advance failed")

    def drain(self):
        out = []
        buf = (Frame * 4096)()
        while True:
            n = self._L.adsb_resolver_drain(self._h, buf, 4096)
            if n <= 0:
                return out
            out.extend(_frames_to_dicts(buf, n))

    def stats(self):
        st = Stats()
        self._L.adsb_resolver_stats(self._h, C.byref(st))
        return _stats_to_dict(st)


def plan_shards(total_samples: int, n_shards: int):
    arrs = [(C.c_uint64 * n_shards)() for _ in range(4)]
    if load().adsb_plan_shards(total_samples, n_shards, *arrs) != 0:
        raise AdsbError("adsb_plan_shards failed")
    return [dict(g_begin=int(arrs[0][i]), g_end=int(arrs[1][i]), first_sample=int(arrs[2][i]),
                 n_samples=int(arrs[3][i])) for i in range(n_shards)]
