"""The kernel's arithmetic must be binary32 multiply-THEN-add: check the
gfx950 ISA that hipcc emits for the scan kernel contains no fused multiply-add of
any flavour, no scratch (spills), and that it is built for gfx950 only."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def isa():
    from adsbdec_amd import _build
    src = os.path.join(ROOT, "adsbdec_amd", "csrc", "scan_kernel.hip")
    cmd = [_build.HIPCC] + _build.HIP_FLAGS + ["--cuda-device-only", "-S", src, "-o", "-"]
    return subprocess.run(cmd, capture_output=True, text=True, check=True).stdout


def test_no_fused_multiply_add_in_scan_kernel(isa):
    fused = re.findall(r"^\s*(v_(?:pk_)?(?:fma|mac|mad|fmac|dot)\w*f(?:32|16)\w*)", isa, flags=re.M)
    assert not fused, f"contracted arithmetic in the ISA: {sorted(set(fused))}"
    assert "-ffp-contract=off" in __import__("adsbdec_amd._build", fromlist=["HIP_FLAGS"]).HIP_FLAGS
    assert re.search(r"v_(pk_)?mul_f32", isa) and re.search(r"v_(pk_)?add_f32", isa)


def test_no_scratch_and_gfx950_only(isa):
    assert ".amdgcn_target \"amdgcn-amd-amdhsa--gfx950\"" in isa
    for m in re.finditer(r"\.private_segment_fixed_size:\s*(\d+)", isa):
        assert int(m.group(1)) == 0, "kernel spills to scratch"
