"""Build driver: hipcc cross-compiles the gfx950 code object without a GPU.

    python -m adsbdec_amd._build            # build lib/libadsbdec_amd.so + the C CLI
    python -m adsbdec_amd._build --force

Outputs stay in-tree (adsbdec_amd/lib/), are git-ignored, and travel to the GPU
box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libadsbdec_amd.so")
CLI = os.path.join(LIBDIR, "adsbdec_amd_cli")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_SOURCES = ["scan_kernel.hip", "decoder.hip"]
C_SOURCES = ["format.c"]
# host-only C++ (no HIP): the multi-GPU driver over the C-ABI, and the part of the C-ABI that needs no device
CXX_SOURCES = ["multi.cpp", "host_abi.cpp", "numa.cpp"]
# What an object depends on is what the compiler says it read: every compile writes <object>.d (-MMD) and the next build
# reads it.  (Until round 6 this was a hand-kept list of headers, and gang.hpp -- included by resolver.hpp -- was not on
# it: editing it alone left stale objects.  tests/test_host_logic.py touches a header and watches the rebuild.)
# -ffp-contract=off: the reference arithmetic is binary32 multiply THEN add
# (SURVEY Q3); a fused multiply-add would change rounding.
# -amdgpu-atomic-optimizer-strategy=None: the compiler otherwise turns the survivor queue's per-lane LDS
# atomicAdd(qcount, n) into a scalar loop over the active lanes (readlane / writelane prefix sum, one iteration per
# lane) -- a fine trade for contended global atomics, a bad one for an LDS counter: kernel -3.9 % without it.
# -Xarch_host -mavx2: the HOST side of decoder.hip (the hand-off stream's check, the resolver's two loops, the frame writer) is
# what a call of the benchmark capture ends on; with 256-bit moves it is 2 % shorter (adsb_decode_device 0.1544 -> 0.1510 ms,
# profiles/r6_ab_runs.txt section 15; BMI / POPCNT alone: nothing).  Every x86-64 host an MI355X ships in has AVX2; one that has
# not is refused by adsb_create with a message (host_abi.cpp adsb_host_cpu_refusal, built without the flag).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall",
             "-Wno-unused-function", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-Xarch_host", "-mavx2"]


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _recorded_deps(obj: str) -> list[str] | None:
    """The files the compiler read for `obj` last time (its -MMD depfile), or None when there is no record."""
    path = obj + ".d"
    if not os.path.exists(path):
        return None
    with open(path) as f:
        text = f.read().replace("\\\n", " ")
    _, _, rest = text.partition(":")
    return [os.path.normpath(w) for w in rest.split()]


def _stale(obj: str, src: str) -> bool:
    deps = _recorded_deps(obj)
    return deps is None or _newer(obj, [src] + deps)


def _run(cmd: list[str]) -> None:
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + p.stdout + p.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for s in HIP_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s + ".o")
        if force or _stale(obj, src):
            if verbose:
                print("hipcc", s)
            _run([HIPCC] + HIP_FLAGS + os.environ.get("ADSB_EXTRA_HIPFLAGS", "").split() + ["-MMD", "-MF", obj + ".d", "-c", src, "-o", obj])
        objs.append(obj)
    for s in C_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s + ".o")
        if force or _stale(obj, src):
            _run(["gcc", "-O2", "-fPIC", "-Wall", "-MMD", "-MF", obj + ".d", "-c", src, "-o", obj])
        objs.append(obj)
    for s in CXX_SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s + ".o")
        if force or _stale(obj, src):
            if verbose:
                print("g++", s)
            _run(["g++", "-O2", "-fPIC", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-MMD", "-MF", obj + ".d", "-c", src, "-o", obj])
        objs.append(obj)
    if force or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lm", "-lpthread"])
    cli_src = os.path.join(CSRC, "cli", "adsbdec_amd_cli.c")
    sink_src = os.path.join(CSRC, "cli", "sink.c")
    if os.path.exists(cli_src) and (force or _newer(CLI, [cli_src, sink_src, os.path.join(CSRC, "cli", "sink.h"), LIB,
                                                           os.path.join(ROOT, "include", "adsbdec_amd.h")])):
        _run(["gcc", "-O2", "-Wall", "-o", CLI, cli_src, sink_src, "-I", os.path.join(ROOT, "include"),
              "-L", LIBDIR, "-ladsbdec_amd", "-lpthread", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
