"""oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front for oracle/liboracle.so (the C restatement of the reference path) and
for oracle/_ref/ref_demod (the real reference demod.c/valid.c/output.c objects).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (adsbdec_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DEMOD = os.path.join(HERE, "_ref", "ref_demod")


class OrcFrame(C.Structure):
    _fields_ = [("g", C.c_uint64), ("ts", C.c_uint64), ("pw", C.c_uint32),
                ("len", C.c_uint8), ("frame", C.c_uint8 * 14)]


_lib = None


def build(force: bool = False) -> None:
    if force or not os.path.exists(LIB_PATH) or (
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "adsb_oracle.c"))):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def build_ref() -> bool:
    """Build oracle/_ref from /root/reference when that tree is present."""
    if not os.path.isdir("/root/reference"):
        return os.path.exists(REF_DEMOD)
    subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    return True


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_decode.restype = C.c_size_t
        L.orc_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(OrcFrame), C.c_size_t,
                                 C.POINTER(C.c_uint32)]
        L.orc_demod_power.restype = C.c_size_t
        L.orc_demod_power.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(OrcFrame), C.c_size_t,
                                      C.POINTER(C.c_uint32)]
        L.orc_decode_fix1.restype = C.c_size_t
        L.orc_decode_fix1.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(OrcFrame), C.c_size_t,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_power.restype = C.c_size_t
        L.orc_power.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_crc_residual.restype = C.c_uint32
        L.orc_crc_residual.argtypes = [C.c_char_p, C.c_int]
        L.orc_crc_table.restype = C.c_uint32
        L.orc_crc_table.argtypes = [C.c_int]
        L.orc_eval_offset.restype = C.c_int
        L.orc_eval_offset.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int),
                                      C.POINTER(C.c_uint32)]
        L.orc_scan_all.restype = C.c_size_t
        L.orc_scan_all.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(OrcFrame),
                                   C.c_size_t, C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]
        L.orc_formatpkt.restype = C.c_int
        L.orc_formatpkt.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.c_char_p]
        _lib = L
    return _lib


def _as_u16(x) -> np.ndarray:
    x = np.ascontiguousarray(x)
    assert x.dtype == np.uint16
    return x


def decode(x, df18: bool = False, cap: int | None = None, fix1: bool = False):
    """Whole-buffer decode. Returns (frames, stats) with frames a list of dicts
    {g, ts, pw, frame(bytes)} and stats {'try': {11,17,18}, 'ok': {...}}.
    fix1=True enables the 1-bit correction EXTENSION (no reference parity); stats then
    also carries 'fixed'."""
    x = _as_u16(x)
    if cap is None:
        cap = max(1024, x.size // 1000)
    nfix = C.c_uint32(0)
    while True:
        out = (OrcFrame * cap)()
        st = (C.c_uint32 * 6)()
        if fix1:
            n = lib().orc_decode_fix1(x.ctypes.data, x.size, int(df18), out, cap, st, C.byref(nfix))
        else:
            n = lib().orc_decode(x.ctypes.data, x.size, int(df18), out, cap, st)
        if n <= cap:
            break
        cap = n
    frames = [dict(g=int(f.g), ts=int(f.ts), pw=int(f.pw), frame=bytes(f.frame[: f.len]))
              for f in out[:n]]
    stats = {"try": {11: st[0], 17: st[1], 18: st[2]}, "ok": {11: st[3], 17: st[4], 18: st[5]}}
    if fix1:
        stats["fixed"] = int(nfix.value)
    return frames, stats


def demod_power(a: np.ndarray, df18: bool = False, cap: int = 1 << 16):
    """demod.c/valid.c stages of the restatement on a float32 power array."""
    assert a.dtype == np.float32
    out = (OrcFrame * cap)()
    st = (C.c_uint32 * 6)()
    n = lib().orc_demod_power(a.ctypes.data, a.size, int(df18), out, cap, st)
    assert n <= cap
    frames = [dict(g=int(f.g), ts=int(f.ts), pw=int(f.pw), frame=bytes(f.frame[: f.len]))
              for f in out[:n]]
    stats = {"try": {11: st[0], 17: st[1], 18: st[2]}, "ok": {11: st[3], 17: st[4], 18: st[5]}}
    return frames, stats


def power(x) -> np.ndarray:
    x = _as_u16(x)
    m = 2 * ((x.size + 3) // 4)
    a = np.empty(m, dtype=np.float32)
    got = lib().orc_power(x.ctypes.data, x.size, a.ctypes.data)
    assert got == m
    return a


def crc_residual(frame: bytes) -> int:
    return int(lib().orc_crc_residual(frame, len(frame)))


def eval_offset(a: np.ndarray, g: int, df18: bool):
    """Stateless evaluation of offset g of a float32 power array (needs g+1196 <= len)."""
    assert a.dtype == np.float32 and g + 1196 <= a.size
    fr = (C.c_uint8 * 14)()
    ln = C.c_int(0)
    pw = C.c_uint32(0)
    k = lib().orc_eval_offset(a.ctypes.data + 4 * g, int(df18), fr, C.byref(ln), C.byref(pw))
    return k, bytes(fr[: ln.value]) if k >= 2 else b"", int(pw.value)


def scan_all(a: np.ndarray, g0: int, g1: int, df18: bool):
    """Exhaustive per-offset evaluation -> (cands [(g, pw, frame)], tries ndarray u64)."""
    assert a.dtype == np.float32 and (g1 <= g0 or g1 - 1 + 1196 <= a.size)
    cc, tc = 4096, 1 << 16
    while True:
        cands = (OrcFrame * cc)()
        tries = np.empty(tc, dtype=np.uint64)
        nc = C.c_size_t(0)
        nt = lib().orc_scan_all(a.ctypes.data, g0, g1, int(df18), cands, cc, C.byref(nc),
                                tries.ctypes.data, tc)
        if nc.value <= cc and nt <= tc:
            break
        cc, tc = max(cc, nc.value), max(tc, nt)
    return ([(int(c.g), int(c.pw), bytes(c.frame[: c.len])) for c in cands[: nc.value]],
            tries[:nt].copy())


def formatpkt(frame: bytes, ts: int, pw: int, outformat: int) -> bytes:
    buf = C.create_string_buffer(256)
    n = lib().orc_formatpkt(frame, len(frame), ts, pw, outformat, buf)
    return buf.raw[:n]


def avr_lines(frames, outformat: int = 0) -> bytes:
    return b"".join(formatpkt(f["frame"], f["ts"], f["pw"], outformat) for f in frames)


# --------------------------------------------------------------------------
# the real reference objects (oracle/_ref), driven on a power-sample buffer
# --------------------------------------------------------------------------
def ref_available() -> bool:
    return os.path.exists(REF_DEMOD)


def ref_demod(a: np.ndarray, df18: bool = False):
    """Run the REAL deqframe/validShort/validLong/formatpkt/print_stats on power
    samples `a`.  Returns (frames, stats); frames carry ts, pw, frame and the three
    formatpkt renderings (avr, mlat, beast)."""
    assert a.dtype == np.float32
    with tempfile.NamedTemporaryFile(suffix=".f32", dir="/tmp") as tf:
        a.tofile(tf.name)
        cmd = [REF_DEMOD] + (["-a"] if df18 else []) + [tf.name]
        p = subprocess.run(cmd, capture_output=True, check=True)
    frames = []
    for line in p.stdout.decode().splitlines():
        ts, pw, ln, avr, rest = line.split(" ", 4)
        # the MLAT rendering contains no spaces; beast hex is last
        mlat, beast = rest.rsplit(" ", 1)
        frames.append(dict(ts=int(ts), pw=int(pw), frame=bytes.fromhex(avr[1:-1]),
                           avr=(avr + "\n").encode(), mlat=(mlat + "\n").encode(),
                           beast=bytes.fromhex(beast)))
        assert len(frames[-1]["frame"]) == int(ln)
    err = p.stderr.decode().splitlines()
    tr = [int(v) for v in err[1].split(":")[1].split()]
    ok = [int(v) for v in err[2].split(":")[1].split()]
    stats = {"try": {11: tr[0], 17: tr[1], 18: tr[2]}, "ok": {11: ok[0], 17: ok[1], 18: ok[2]}}
    return frames, stats
