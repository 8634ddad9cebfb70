#!/usr/bin/env python3
"""VALU instruction mix of scan_kernel's Stage A loop -> profiles/<tag>_isa_mix.json.

    python tools/isa_mix.py profiles/r2_isa_mix.json [extra hipcc flags ...]

Compiles adsbdec_amd/csrc/scan_kernel.hip to gfx950 assembly with the flags the library
is built with, takes the straight-line body of the Stage A pass loop of
adsb::scan_kernel<false> (from the end of the typed-load block to the workgroup barrier
that closes Stage A) -- ~94 % of the kernel's dynamic VALU count -- and prices every VALU
instruction at its issue cost on a SIMD-32 (MI355X_MICROARCH.md: a wave64 VALU instruction
issues over 2 cycles; the packed-f32 forms, which produce two results per lane, and the
forms tools/valu_bench.hip measured at 4.2-4.4 cycles -- v_alignbit, v_trunc, DPP moves,
v_fma/v_pk_fma, conversions, compares -- take 4).  bench.py multiplies the PMC count of
VALU wave-instructions by the resulting mean to get the launch's VALU-issue floor.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FOUR = re.compile(r"^v_(pk_|alignbit|trunc|fma|fmac|cvt|cmp|perm|lshl_or|and_or|lshlrev|lshrrev|or3|lshl_add|add3|bfe|bfi|mad|readlane|readfirstlane)")


def cost(mn: str, line: str) -> int:
    if "dpp" in mn or "dpp" in line or "sdwa" in mn:
        return 4
    return 4 if FOUR.match(mn) else 2


def main():
    dst, extra = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "scan.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
                               "-S", "--cuda-device-only", "-o", s] + extra +
                              [os.path.join(ROOT, "adsbdec_amd", "csrc", "scan_kernel.hip")],
                              stderr=subprocess.DEVNULL)
        lines = open(s).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_ZN4adsb11scan_kernelILb0EEEvNS_8ScanArgsE:"))
    body, on = [], False
    for ln in lines[start:]:
        if "buffer_load_format_xyzw" in ln:
            on, body = True, []          # restart after the last load of the block
            continue
        if on and "s_barrier" in ln:
            break
        if on:
            body.append(ln)
    hist = collections.Counter()
    cycles = 0
    n_valu = n_nop = 0
    for ln in body:
        m = re.match(r"^\t([a-z_0-9]+)\s", ln)
        if not m:
            continue
        mn = m.group(1)
        if mn.startswith("v_"):
            hist[mn] += 1
            n_valu += 1
            cycles += cost(mn, ln)
        elif mn == "s_nop":
            n_nop += 1
    out = {"what": "static VALU mix of one Stage A pass (28 power samples per lane) of adsb::scan_kernel<false>, "
                   "gfx950, flags of adsbdec_amd/_build.py" + (" + " + " ".join(extra) if extra else ""),
           "valu_instructions_per_pass": n_valu, "issue_cycles_per_pass": cycles, "s_nop_per_pass": n_nop,
           "cycles_per_valu_instruction": round(cycles / n_valu, 4),
           "cost_model": "2 cycles per wave64 VALU instruction on a SIMD-32; 4 for v_pk_*, v_alignbit, v_trunc, v_fma, DPP/SDWA "
                         "forms, conversions, compares, three-operand integer forms (tools/valu_bench.hip: 4.2-4.4 measured)",
           "histogram": dict(hist.most_common())}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "histogram"}, indent=1))


if __name__ == "__main__":
    main()
