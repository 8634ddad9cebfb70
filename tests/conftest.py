"""pytest configuration: the `gpu` marker, repo-root imports, build-once fixtures."""
import json
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- BEFORE libadsbdec_amd.so is dlopen'ed: torch bundles its own libamdhip64, and a
#                process that loads /opt/rocm's runtime first and torch's second ends up with two HIP runtimes,
#                the second of which finds no device.  Loaded in this order the library binds to torch's copy.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def capi():
    """ctypes binding of the product library; builds it when sources are newer."""
    from adsbdec_amd import _build, capi as C
    _build.build()
    C.load()
    return C


def golden_cases():
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json") and f != "crc_kat.json")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        rec = json.load(f)
    x = np.load(os.path.join(GOLDEN, rec["input"]))["x"]
    assert x.size == rec["n_samples"]
    rec["stats"] = {k: {int(d): v for d, v in rec["stats"][k].items()} for k in rec["stats"]}
    return x, rec


def records(frames):
    """Comparable view of decoded frames: (g, ts, pw, frame-bytes)."""
    return [(f["g"], f["ts"], f["pw"], bytes(f["frame"])) for f in frames]


def golden_records(rec):
    return [(f["g"], f["ts"], f["pw"], bytes.fromhex(f["frame"])) for f in rec["frames"]]


def shard_power(oracle, x, shard):
    """Power samples of one planned shard computed by the oracle AT THE SHARD'S TRUE
    STREAM PHASE.  The FIR's summation order depends on the absolute sample index
    mod 14, so the shard is left-padded with silence back to a multiple
    of 28 samples.  Returns (a, off): a[k] is the stream's power sample k + off.
    Valid from 6 pairs after the shard start (the planner's pre-halo)."""
    s0 = shard["first_sample"]
    pad = s0 % 28
    xs = np.concatenate([np.full(pad, 2048, np.uint16), x[s0: s0 + shard["n_samples"]]])
    return oracle.power(xs), (s0 - pad) // 2
