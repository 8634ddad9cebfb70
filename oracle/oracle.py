"""oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front for oracle/liboracle.so (the C restatement of the reference path) and
runner of oracle/_ref/ref_adsbdec / ref_power (the REAL reference code: air.c:29-101,
demod.c, valid.c, output.c:formatpkt compiled from /root/reference, oracle/Makefile).
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
REF_ADSBDEC = os.path.join(HERE, "_ref", "ref_adsbdec")
REF_ADSBDEC_NATIVE = os.path.join(HERE, "_ref", "ref_adsbdec_native")
REF_POWER = os.path.join(HERE, "_ref", "ref_power")
# captures handed to the reference binaries go through files; /dev/shm keeps them in memory
_TMP = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"


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
        return ref_available()
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
# the REAL reference code (oracle/_ref): one process per run, because the
# reference keeps its stream state in statics (air.c:33-34,49-50; demod.c:86;
# valid.c:30-31)
# --------------------------------------------------------------------------
def ref_available() -> bool:
    return os.path.exists(REF_ADSBDEC) and os.path.exists(REF_POWER)


def _parse_ref_output(p):
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


def ref_decode(x, df18: bool = False, chunk: int | None = None, native: bool = False, path: str | None = None):
    """The whole REAL chain on a uint16 capture: decodeiq (air.c:54-101) -> deqframe
    (demod.c) -> validShort/Long (valid.c, crc.h) -> formatpkt (output.c), fed by a
    fileInput-shaped read loop (1 Mi samples per decodeiq call unless `chunk`).
    Returns (frames, stats); frames carry ts, pw, frame and the three formatpkt
    renderings (avr, mlat, beast) -- no g: the reference does not know it.
    native=True runs the build with the reference's own -O3 -march=native flags.
    path: decode this file instead of writing x to a temporary one."""
    exe = REF_ADSBDEC_NATIVE if native else REF_ADSBDEC
    opts = (["-a"] if df18 else []) + (["-c", str(chunk)] if chunk else [])
    if path is not None:
        return _parse_ref_output(subprocess.run([exe] + opts + [path], capture_output=True, check=True))
    x = _as_u16(x)
    with tempfile.NamedTemporaryFile(suffix=".u16", dir=_TMP) as tf:
        x.tofile(tf.name)
        return _parse_ref_output(subprocess.run([exe] + opts + [tf.name], capture_output=True, check=True))


def ref_power(x, chunk: int | None = None) -> np.ndarray:
    """Power samples produced by the REAL decodeiq (air.c:54-92) for capture x: one
    per 2 input samples, 2*ceil(n/4) of them (a ragged tail's last pair is computed
    from whatever the read buffer held, like the reference)."""
    x = _as_u16(x)
    with tempfile.NamedTemporaryFile(suffix=".u16", dir=_TMP) as ti, \
            tempfile.NamedTemporaryFile(suffix=".f32", dir=_TMP) as to:
        x.tofile(ti.name)
        subprocess.run([REF_POWER] + (["-c", str(chunk)] if chunk else []) + [ti.name, to.name], check=True)
        return np.fromfile(to.name, dtype=np.float32)


def ref_demod(a: np.ndarray, df18: bool = False):
    """The REAL deqframe/validShort/validLong/formatpkt/print_stats on POWER samples
    `a` (synthetic power fuzz of getdf/getabyte; the carry loop around deqframe is
    the harness's restatement of air.c:94-99 in this mode -- pinning uses ref_decode)."""
    assert a.dtype == np.float32
    with tempfile.NamedTemporaryFile(suffix=".f32", dir=_TMP) as tf:
        a.tofile(tf.name)
        cmd = [REF_ADSBDEC] + (["-a"] if df18 else []) + ["-p", tf.name]
        return _parse_ref_output(subprocess.run(cmd, capture_output=True, check=True))
