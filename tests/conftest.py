"""pytest configuration: the `gpu` marker, repo-root imports, build-once fixtures."""
import faulthandler
import json
import os
import signal
import sys
import threading

import numpy as np
import pytest
import torch  # noqa: F401  -- BEFORE libadsbdec_amd.so is dlopen'ed: torch bundles its own libamdhip64, and a
#                process that loads /opt/rocm's runtime first and torch's second ends up with two HIP runtimes,
#                the second of which finds no device.  Loaded in this order the library binds to torch's copy.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--gpu-big", action="store_true", default=False,
                     help="also run the opt-in gpu_big tests (many GiB of HBM, minutes of wall time)")


# Order of the -m gpu run: cheap and most diagnostic first, the heavy full-size configurations last, so that a run cut short
# (the driver's limit, -x) has seen every golden fixture, configs[3] and the multi-process shape before anything large.
_ORDER = (
    ("golden", 0),                                   # the committed fixtures, 0.1 s each (all files)
    ("test_gpu_parity.py::test_seeded", 1),
    ("test_generator_digests_on_the_device", 1),
    ("test_multi_independent_streams", 2),           # configs[3]: one capture per handle ...
    ("test_two_rank_independent_streams", 2),        # ... and one PROCESS per device, the driver's command shape
    ("test_eight_ranks_on_one_device", 2),
    ("test_a_rank_that_dies_ends_the_job", 2),
    ("test_cli_", 4),                                # the C host program
    ("test_fuzz_30s", 8),
    ("test_round_trip_at_scale", 8),
    ("test_gpu_parity.py", 3),
    ("test_gpu_dropin.py", 5),                       # the reference's own harness with INTEGRATION.md's patch
    ("test_gpu_multi.py::test_bench_shard_mode", 8),
    ("test_gpu_multi.py", 6),
    ("test_gpu_full_configs.py", 9),                 # BASELINE's configurations at their stated sizes
)


def _rank(item):
    for key, r in _ORDER:
        if key in item.nodeid:
            return r
    return 3


def pytest_collection_modifyitems(config, items):
    if not config.getoption("--gpu-big"):
        big = [it for it in items if it.get_closest_marker("gpu_big")]
        if big:
            config.hook.pytest_deselected(items=big)
            items[:] = [it for it in items if not it.get_closest_marker("gpu_big")]
    gpu = [it for it in items if it.get_closest_marker("gpu")]
    if gpu:
        rest = [it for it in items if not it.get_closest_marker("gpu")]
        gpu.sort(key=_rank)          # stable: the files' own order inside a rank
        items[:] = rest + gpu


class TestTimeLimit(Exception):
    pass


def _dump_all_stacks(what):
    sys.stderr.write(f"\n===== {what}: stacks of every thread =====\n")
    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
    sys.stderr.flush()


@pytest.fixture(autouse=True)
def _per_test_time_limit(request):
    """No test may hang the run.  At the limit SIGALRM fails the test with every thread's stack; if the main thread sits in a C
    call that never returns (the signal handler only runs between bytecodes), a watchdog of faulthandler's own (a C thread:
    it needs no GIL) dumps the stacks 45 s later and ENDS the process -- a red run with a cause instead of a killed one."""
    m = request.node.get_closest_marker("limit")
    limit = float(m.args[0]) if m else 300.0
    if threading.current_thread() is not threading.main_thread() or not hasattr(signal, "SIGALRM"):
        yield
        return

    def on_alarm(signum, frame):
        _dump_all_stacks(f"{request.node.nodeid} exceeded its limit of {limit:.0f} s")
        raise TestTimeLimit(f"{request.node.nodeid} exceeded its limit of {limit:.0f} s (stacks on stderr)")

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.setitimer(signal.ITIMER_REAL, limit)
    faulthandler.dump_traceback_later(limit + 45.0, exit=True, file=sys.stderr)
    try:
        yield
    finally:
        signal.setitimer(signal.ITIMER_REAL, 0)
        signal.signal(signal.SIGALRM, old)
        faulthandler.cancel_dump_traceback_later()


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
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json") and f not in ("crc_kat.json", "generator_digests.json"))


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


def preamble_pass_fraction(x_host, n=4 << 20):
    """Share of the preamble offsets of a prefix that pass demod.c:102-107 (p1 > 2 s1 && p2 > 2 s2), from the oracle's
    power samples: the density figure BASELINE configs[2] quotes ("~10 % of offsets above preamble threshold")."""
    from oracle import oracle as O
    O.build()
    a = O.power(np.ascontiguousarray(x_host[:n]))
    m = a.size - 1196
    c = np.trunc(a[:-10] + a[10:]).astype(np.int64)   # c[k] = (int)(a[k] + a[k+10]): all four sums have this form
    p1, s1, s2, p2 = c[0:m], c[5:m + 5], c[30:m + 30], c[35:m + 35]
    return float(np.mean((p1 > 2 * s1) & (p2 > 2 * s2)))
