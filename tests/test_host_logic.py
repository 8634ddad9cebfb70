"""CPU tests of the product's host logic, through the C-ABI (no GPU compute calls):
library loads and exports every declared symbol, the greedy resolver replays the
reference's sequential rules, formatter bytes, shard planner, and that creating a
decoder without a GPU fails loudly instead of falling back to anything."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden_cases, golden_records, load_golden, records, shard_power


def test_library_exports_every_declared_symbol(capi):
    declared = set()
    for h in ("adsbdec_amd.h", "adsbdec_amd_diag.h"):   # comments stripped: they name macros and calls of other rounds
        header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
        declared |= set(re.findall(r"\b(adsb_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    declared -= {"adsb_get_profile", "adsb_multi_worker_profile"}   # macros that pass the caller's sizeof to the _sized calls
    L = capi.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))
    assert L.adsb_abi_version() == 5


def test_struct_layouts_match_header(capi):
    import ctypes as C
    assert C.sizeof(capi.Frame) == 40 and C.sizeof(capi.Candidate) == 32
    assert C.sizeof(capi.Stats) == 56
    assert C.sizeof(capi.ShardHead) == 17 * 8 and C.sizeof(capi.ShardPart) == 10 * 8 and C.sizeof(capi.ShardFix) == 40
    assert C.sizeof(capi.MultiInfo) == 72 and C.sizeof(capi.Profile) == 104 + 16 and capi.Profile.host_threads_running.offset == 104
    assert capi.Frame.frame.offset == 21 and capi.Candidate.frame.offset == 13


def test_product_does_not_touch_the_oracle():
    """The shipped path must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "adsbdec_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hpp", ".hip", ".cpp")):
                src = open(os.path.join(base, f), errors="ignore").read()
                assert not re.search(r"(^|\n)\s*(from|import)\s+oracle\b", src), f
                assert "liboracle" not in src and "adsb_oracle.h" not in src, f


def test_create_without_gpu_fails_loudly(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.AdsbError, match="no HIP device|no CPU fallback|failed"):
        capi.Decoder()


def test_multi_create_without_gpu_fails_loudly(capi):
    """The multi-GPU driver has no CPU path either: every worker's adsb_create fails, adsb_multi_create returns NULL and the
    message names the device and the worker; nothing hangs, no thread is left behind."""
    import threading
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from adsbdec_amd import sharding
    before = threading.active_count()
    for n in (1, 3):
        with pytest.raises(sharding.ShardError, match=r"device 0 \(worker 0\): adsb_create failed: no HIP device"):
            sharding.MultiDecoder(n)
    with pytest.raises(sharding.ShardError, match="n_devices must be"):
        sharding.MultiDecoder(0)
    assert threading.active_count() == before


def test_config_abi_guard_and_growth_rule(capi):
    """ABI 5 (round 6): adsb_config carries `abi` in second place and the debug_* knobs left it for adsb_debug_config behind
    `debug`.  adsb_create reads no further than cfg.struct_size (a struct that ends early gets past the check and, here,
    fails for the lack of a GPU); a size from the future, a struct of ABI <= 4 (its df18 lies where `abi` is; the legacy
    symbol adsb_config_default leaves one behind) and a debug struct of an unknown size are refused BY NAME."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = capi.load()
    assert L.adsb_abi_version() == 5
    assert capi.Config.abi.offset == 4 and capi.Config.host_threads.offset == 52 and capi.Config.debug.offset == 64
    assert C.sizeof(capi.Config) == 72 and C.sizeof(capi.DebugConfig) == 48
    cfg = capi.Config()
    C.memset(C.byref(cfg), 0xEE, C.sizeof(cfg))
    L.adsb_config_init(C.byref(cfg), C.sizeof(cfg))
    assert cfg.struct_size == 72 and cfg.abi == 5 and cfg.device == -1 and cfg.host_threads == 0 and not cfg.debug
    assert not L.adsb_create(C.byref(cfg)) and b"no HIP device" in L.adsb_last_error(None)
    # a caller whose struct ends in front of wait_timeout_s / debug (a future ABI-5 header may be LONGER, never shorter -- but the
    # rule is what is tested: nothing behind struct_size is read)
    for end in (capi.Config.wait_timeout_s.offset, capi.Config.debug.offset):
        short = capi.Config()
        C.memset(C.byref(short), 0xEE, C.sizeof(short))
        L.adsb_config_init(C.byref(short), end)
        assert short.struct_size == end and short.abi == 5 and short.debug == 0xEEEEEEEEEEEEEEEE   # (untouched behind `end`)
        assert not L.adsb_create(C.byref(short)) and b"no HIP device" in L.adsb_last_error(None)
    cfg.struct_size = 4096
    assert not L.adsb_create(C.byref(cfg)) and b"struct_size" in L.adsb_last_error(None)
    cfg.struct_size = 72
    for wrong in (0, 1, 4, 6):     # ABI <= 4 binaries have df18 (0 / 1) there
        cfg.abi = wrong
        assert not L.adsb_create(C.byref(cfg))
        assert b"adsb_config.abi" in L.adsb_last_error(None) and b"rebuilt" in L.adsb_last_error(None)
        assert not L.adsb_multi_create(C.byref(cfg), 1, None) and b"adsb_config.abi" in L.adsb_multi_last_error(None)
    legacy = (C.c_uint8 * 128)(*([0xEE] * 128))
    L.adsb_config_default(legacy)                      # what a binary of ABI <= 4 calls: 72 zero bytes, struct_size 72, no abi
    assert bytes(legacy[:4]) == (72).to_bytes(4, "little") and not any(legacy[4:72]) and all(v == 0xEE for v in legacy[72:])
    assert not L.adsb_create(legacy) and b"adsb_config.abi" in L.adsb_last_error(None)
    # the test knobs: copied at adsb_create, their struct has a size of its own
    cfg = capi.make_config(debug_queue_cap=256, debug_gang_min=1)
    assert cfg.debug and cfg._debug.queue_cap == 256 and cfg._debug.gang_min == 1 and cfg._debug.struct_size == 48
    assert not L.adsb_create(C.byref(cfg)) and b"no HIP device" in L.adsb_last_error(None)
    cfg._debug.struct_size = 4000
    assert not L.adsb_create(C.byref(cfg)) and b"adsb_debug_config.struct_size" in L.adsb_last_error(None)
    with pytest.raises(TypeError):
        capi.make_config(debug_no_such_knob=1)
    assert not capi.make_config(host_threads=1).debug


def test_drop_in_header_is_short_and_self_contained(tmp_path):
    """include/adsbdec_amd.h is what a drop-in and a multi-GPU host call (the reference's whole interface is one prototype,
    adsbdec.h:5): it stays short, names no test knob, and the C host program, the patch of INTEGRATION.md and format.c build
    against it ALONE; everything else is in adsbdec_amd_diag.h."""
    import re
    import subprocess
    inc = os.path.join(ROOT, "include")
    main = open(os.path.join(inc, "adsbdec_amd.h")).read()
    assert len(main.splitlines()) <= 250
    code = re.sub(r"/\*.*?\*/", "", main, flags=re.S)   # (comments may say where the knobs went)
    assert "debug_" not in code and "adsb_resolver" not in code and "adsb_stitch" not in code and "adsb_candidate" not in code
    src = tmp_path / "only_main.c"
    src.write_text('#include "adsbdec_amd.h"\nint main(void) { adsb_config c; adsb_config_default(&c); adsb_profile p; (void)p; '
                   'return c.abi == ADSB_ABI_VERSION ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "o.o")], check=True)
    for f in (os.path.join(ROOT, "adsbdec_amd", "csrc", "cli", "adsbdec_amd_cli.c"), os.path.join(ROOT, "adsbdec_amd", "csrc", "format.c"),
              os.path.join(ROOT, "oracle", "dropin_decodeiq.c")):
        text = open(f).read()
        assert "adsbdec_amd_diag.h" not in text and re.search(r'#include\s+"[./]*(include/)?adsbdec_amd\.h"', text), f
    # every entry point the two headers declare is exported, and nothing else with the library's prefix is
    declared = set()
    for h in ("adsbdec_amd.h", "adsbdec_amd_diag.h"):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, h)).read(), flags=re.S)
        declared |= set(re.findall(r"\b(adsb_[a-z0-9_]+)\s*\(", text))
    declared -= {"adsb_get_profile", "adsb_multi_worker_profile"}          # macros over the _sized calls
    from adsbdec_amd import _build
    out = subprocess.run(["nm", "-D", "--defined-only", _build.LIB], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (adsb_[a-z0-9_]+)", out))
    assert declared == exported, (sorted(declared - exported), sorted(exported - declared))


@pytest.mark.parametrize("name", golden_cases())
def test_resolver_replays_reference_order_on_golden(capi, oracle, name):
    x, rec = load_golden(name)
    a = oracle.power(x)
    m_real = 2 * (x.size // 4)
    g_end = max(0, m_real - 1196 + 1)
    cands, tries = oracle.scan_all(a, 0, g_end, rec["df18"])
    r = capi.Resolver()
    r.feed(cands, tries)
    r.advance(2 * ((x.size + 3) // 4), g_end)
    assert records(r.drain()) == golden_records(rec)
    assert r.stats() == rec["stats"]


def test_resolver_streaming_equals_one_shot(capi, oracle):
    """Feeding candidates piecewise as the stream grows gives the same frames."""
    from tools import gen_signal as G
    x, _ = G.dense_capture(1 << 19, seed=21, sigma=60.0, n_frames=200)
    a = oracle.power(x)
    g_end = a.size - 1195
    cands, tries = oracle.scan_all(a, 0, g_end, True)
    want, wstats = oracle.decode(x, df18=True)
    r = capi.Resolver()
    step, done = 28 * 500, 0
    while done < g_end:
        nxt = min(g_end, done + step)
        r.feed([c for c in cands if done <= c[0] < nxt], tries[(tries >> 2 >= done) & (tries >> 2 < nxt)])
        r.advance(min(a.size, nxt + 1195), nxt)
        done = nxt
    r.advance(a.size, g_end)
    assert records(r.drain()) == records(want)
    assert r.stats() == wstats


@pytest.mark.parametrize("name", golden_cases())
def test_format_frame_bytes(capi, name):
    _, rec = load_golden(name)
    for g in rec["frames"]:
        fr = dict(g=g["g"], ts=g["ts"], pw=g["pw"], frame=bytes.fromhex(g["frame"]))
        assert capi.format_frame(fr, 0) == g["avr"].encode()
        assert capi.format_frame(fr, 1) == g["mlat"].encode()
        assert capi.format_frame(fr, 2) == bytes.fromhex(g["beast"])


def test_format_frame_escapes_and_large_ts(capi, oracle):
    fr = dict(g=0, ts=(0x1A1A1A1A1A1A * 10) // 12 + 1, pw=123456, frame=bytes([0x8D, 0x1A, 0x1A] + [0x1A] * 11))
    for fmt in (0, 1, 2):
        assert capi.format_frame(fr, fmt) == oracle.formatpkt(fr["frame"], fr["ts"], fr["pw"], fmt)
    short = dict(g=0, ts=2**50 + 7, pw=0, frame=bytes([0x5D, 1, 2, 3, 4, 5, 0x1A]))
    for fmt in (0, 1, 2):
        assert capi.format_frame(short, fmt) == oracle.formatpkt(short["frame"], short["ts"], 0, fmt)


@pytest.mark.parametrize("total,n", [(1 << 20, 1), (1 << 20, 2), (1 << 20, 8), ((1 << 22) + 6, 3), (5000, 4), (100, 2)])
def test_plan_shards_partition(capi, total, n):
    plan = capi.plan_shards(total, n)
    m = 2 * (total // 4)
    n_off = max(0, m - 1195)
    assert plan[0]["g_begin"] == 0 and plan[-1]["g_end"] == n_off
    for i, s in enumerate(plan):
        assert s["g_begin"] % 28 == 0 and s["first_sample"] % 8 == 0
        if i:
            assert s["g_begin"] == plan[i - 1]["g_end"]
        if s["g_end"] > s["g_begin"]:
            # the buffer covers the pre-halo (6 pairs) and one full window after the last offset
            assert s["first_sample"] <= max(0, 2 * (s["g_begin"] - 6))
            assert s["first_sample"] + s["n_samples"] >= 2 * (s["g_end"] - 1 + 1196)
            assert s["first_sample"] + s["n_samples"] <= total


def test_sharded_candidates_resolve_like_one_stream(capi, oracle):
    """SURVEY 8e on CPU: per-shard exhaustive scans (oracle standing in for the device,
    tests only) over the planner's halo'd sample ranges + one host resolver == the
    sequential reference."""
    from tools import gen_signal as G
    x, _ = G.dense_capture(1 << 20, seed=33, sigma=45.0, n_frames=300)
    want, wstats = oracle.decode(x, df18=True)
    for n_shards in (2, 5):
        r = capi.Resolver()
        for s in capi.plan_shards(x.size, n_shards):
            a, off = shard_power(oracle, x, s)
            cands, tries = oracle.scan_all(a, s["g_begin"] - off, s["g_end"] - off, True)
            r.feed([(g + off, pw, fr) for g, pw, fr in cands], tries + np.uint64(off << 2))
        r.advance(2 * (x.size // 4), 2 * (x.size // 4) - 1195)
        assert records(r.drain()) == records(want)
        assert r.stats() == wstats


def test_resolver_input_paths_agree(tmp_path):
    """adsb::Resolver takes records three ways -- the candidate queue (pinned against the
    oracle above), device records through an index list, and tile ranges of the hand-off
    stream consumed in place -- and the last two only run on a GPU box otherwise.  A C++
    harness (tests/cpp/resolver_paths.cpp, g++) feeds random clustered candidate sets
    through all three (and through the third with a gang of frame-writing threads) in random batch sizes and compares frames, ts and Ok counters."""
    import subprocess
    exe = tmp_path / "resolver_paths"
    src = os.path.join(ROOT, "tests", "cpp", "resolver_paths.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-pthread", src, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), "150"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
    # and once more under AddressSanitizer + UBSan (the in-place batch paths juggle raw pointers)
    san = tmp_path / "resolver_paths_san"
    built = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-pthread", src, "-o", str(san)], capture_output=True, text=True)
    if built.returncode == 0:  # sanitizer runtimes present in this image
        out = subprocess.run([str(san), "40"], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
    # ... and under ThreadSanitizer: path (d) hands the frames to a gang of three threads (csrc/gang.hpp)
    tsan = tmp_path / "resolver_paths_tsan"
    built = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", src, "-o", str(tsan)],
                           capture_output=True, text=True)
    if built.returncode == 0:
        out = subprocess.run([str(tsan), "40"], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.startswith("ok") and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr


def test_formatter_under_sanitizers(tmp_path):
    """csrc/format.c (== formatpkt, output.c:204-262) on 300 k random frames, every output format, built with
    AddressSanitizer + UBSan: lengths stay within the documented bounds, nothing is read or written out of range."""
    import subprocess
    exe = tmp_path / "format_fuzz"
    built = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "format_fuzz.c"),
                            os.path.join(ROOT, "adsbdec_amd", "csrc", "format.c"), "-lm", "-o", str(exe)],
                           capture_output=True, text=True)
    if built.returncode != 0:
        pytest.skip("sanitizer runtimes absent: " + built.stderr[-200:])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_tile_geometry_is_consistent(tmp_path):
    """scan_kernel.h's tile geometry (which tile owns which runs, with or without the
    staggered first round) is evaluated by the kernel AND by the host that turns "tiles
    below t are in" into "offsets below g are complete": count, abutment and pass counts
    are checked on the CPU (hipcc host compile of tests/cpp/tile_geometry.hip)."""
    import subprocess
    from adsbdec_amd import _build
    exe = tmp_path / "tile_geometry"
    src = os.path.join(ROOT, "tests", "cpp", "tile_geometry.hip")
    subprocess.run([_build.HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", src, "-o", str(exe)], check=True,
                   capture_output=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


# ------------------------------------------------------------------ time-sharded stream, resolved per shard (stitch.hpp)
def _stitch_case(capi, cands, total, n_shards, head_span, tries=None):
    """Per-shard chain resolution of the exhaustive candidate list + adsb_stitch_shards[_stats]
    -> (rc, final frame records, frames the seam repairs touched, stats | None)."""
    import shard_helpers
    ss = shard_helpers.from_candidates(capi, cands, total, n_shards, head_span=head_span, tries=tries)
    rc, out, stats, _, fixes = ss.stitch(with_stats=tries is not None)
    return rc, out, sum(d + n for d, n, _ in fixes) if rc == 0 else 0, stats


@pytest.mark.parametrize("kind", ["sparse", "dense_overlapping", "back_to_back", "ragged_short"])
def test_stitched_shards_equal_the_sequential_decode(capi, kind):
    """One stream cut into 1..13 shards, every shard resolved on its own (greedy chain from its first offset), then
    adsb_stitch_shards: seam repair from the head candidates, per-shard ts offsets, end-of-file horizon.  Equal to the
    oracle's sequential decode (demod.c:86-143 + air.c:94-99) whatever the cut -- with frames packed back to back across
    every seam too -- or, when the head window is too small to decide a seam, an honest -3."""
    from tools import gen_signal as G
    from oracle import oracle as O
    O.build()
    rng = np.random.default_rng(5)
    stats_checked = []
    if kind == "sparse":
        x, _ = G.sparse_capture(1 << 20, 180, seed=11, sigma=8.0, dfs=(17, 11))
    elif kind == "dense_overlapping":
        x, _ = G.dense_capture(1 << 20, seed=12, sigma=40.0, n_frames=700, amp=(300, 1800))
    elif kind == "back_to_back":
        placed = [(10_000 + 2_400 * i, G.make_frame([17, 18, 17, 11][i % 4], rng), float(rng.uniform(500, 1500)), float(i))
                  for i in range(380)]
        x = G.synth(10_000 + 2_400 * 380 + 120_000, placed, 6.0, 7)
    else:
        x, _ = G.dense_capture(300_002, seed=13, sigma=30.0, n_frames=150, amp=(300, 1800))
    want, wstats = O.decode(x, df18=True)
    want = [(f["g"], f["ts"], f["pw"], f["frame"]) for f in want]
    a = O.power(x)
    cands, tries = O.scan_all(a, 0, max(0, a.size - 1195), True)
    assert len(want) > 20
    undecided = repairs = 0
    for n_shards in (1, 2, 3, 5, 8, 13):
        for head_span in (16384, 2400, 300):
            rc, got, rep, _ = _stitch_case(capi, cands, x.size, n_shards, head_span)
            repairs += rep
            assert rc in (0, -3), rc
            if rc == -3:
                assert head_span < 16384, "the default head window must decide these seams"
                undecided += 1
                continue
            assert got == want, (kind, n_shards, head_span, len(got), len(want))
            # the same with statistics: every shard's own Try count (against its speculative chain), corrected behind the
            # seams and beyond the horizon from the two windows of tries, must be valid.c's table for the whole stream
            rc2, got2, _, stats = _stitch_case(capi, cands, x.size, n_shards, head_span, tries=tries)
            assert rc2 in (0, -3), rc2      # (-3 here: a shard too small to hold its windows apart)
            if rc2 == 0:
                assert got2 == want and stats == wstats, (kind, n_shards, head_span, stats, wstats)
                stats_checked.append(n_shards)
    assert len(stats_checked) >= 6 and max(stats_checked) >= 5, stats_checked
    if kind == "sparse":
        assert undecided == 0
    if kind == "back_to_back":
        assert repairs > 0, "no seam of these cuts needed a repair: the test does not test"


def _synthetic_candidates(rng, n_offsets, mean_gap, frame=bytes([0x8D]) + bytes(13)):
    """Candidate records as a sparse stream would produce them, without any signal: one 'frame' per ~mean_gap offsets
    (long and short mixed), each decoding at 1-3 neighbouring offsets, now and then two frames overlapping."""
    out, g = [], int(rng.integers(0, mean_gap))
    while g < n_offsets:
        fr = frame if rng.random() < 0.8 else bytes([0x5D]) + bytes(6)
        for d in range(int(rng.integers(1, 4))):
            if g + d < n_offsets:
                out.append((g + d, 100 + d, fr))
        g += int(rng.integers(3, 2 * mean_gap)) if rng.random() < 0.9 else int(rng.integers(3, 1300))
    return out


@pytest.mark.parametrize("seed,total,n_shards", [(1, 64 << 20, 8), (2, 64 << 20, 3), (3, (96 << 20) + 1234, 13), (4, 32 << 20, 2),
                                                 (5, (16 << 20) + 40990 * 4 + 6, 5)])
def test_stitcher_horizon_walk_jumps_onto_the_shards_own_walks(capi, seed, total, n_shards):
    """The end-of-file horizon needs the whole deqframe call chain (air.c:94-99: a frame that straddles a call's limit moves
    every later call).  Every shard walks its own calls from a guessed entry base; the stitcher walks the true chain only
    until it meets one of a shard's bases and jumps to that shard's exit.  Checked against the ONE sequential resolver
    (resolver.hpp, itself pinned by the oracle) on synthetic candidate streams of up to 48 Mi offsets (~1 200 calls): same
    frames, same ts, same cut at the horizon.  (Two chains only meet when both straddle the same frame, so how much is
    jumped depends on the traffic: the test asks that jumps happen and that every call is accounted for, not for a share.)"""
    import shard_helpers
    rng = np.random.default_rng(seed)
    m = 2 * (total // 4)
    n_off = m - 1195
    cands = _synthetic_candidates(rng, n_off, 9000)
    one = capi.Resolver()
    one.feed(cands)
    one.advance(2 * ((total + 3) // 4), n_off)
    want = [(f["g"], f["ts"], f["pw"], f["frame"]) for f in one.drain()]
    assert len(want) > 800
    ss = shard_helpers.from_candidates(capi, cands, total, n_shards)
    assert all(h.status == 0 and h.n_bases >= 1 for h in ss.heads)
    rc, got, _, (walked, jumped), fixes = ss.stitch(with_bases=True)
    assert rc == 0 and got == want
    # every call of the chain is either walked by the stitcher or covered by a jump: count them with the plain walk
    rc2, got2, _, (walked2, jumped2), fixes2 = ss.stitch(with_bases=False)
    assert rc2 == 0 and got2 == want
    assert jumped2 == 0 and walked + jumped == walked2 > 150
    assert fixes2 == fixes
    _JUMPS.append(jumped)


_JUMPS = []


def test_stitcher_jumps_happened():
    assert sum(_JUMPS) > 0, "no run of the previous test ever met a shard's own walk: the jump path is untested"


@pytest.mark.parametrize("seed,drain_between", [(1, False), (2, False), (3, False), (1, True), (4, True)])
def test_incremental_call_walk_equals_the_walk_over_the_finished_shard(capi, seed, drain_between):
    """adsb_scan_shard_resolved_walk advances the shard's walk of the deqframe calls beside the greedy chain, as the frames come
    in (a call is replayed once the chain has passed its limit).  Whatever the batching of the records, the bases must be the
    ones adsb_shard_walk finds on the finished shard -- also when the caller drains the frames between two advances (the walk
    keeps its own record of the accepted frames: round 3's advisor found it indexing the output queue, which a drain recycles)."""
    import ctypes as C
    L = capi.load()
    rng = np.random.default_rng(seed)
    total = (24 << 20) + 4 * int(rng.integers(0, 50_000))
    n_off = 2 * (total // 4) - 1195
    cands = _synthetic_candidates(rng, n_off, 7000)
    for g_begin, g_end in ((0, n_off), (28 * 100_000, 28 * 300_000), (28 * 200_001, n_off)):
        mine = [c for c in cands if g_begin <= c[0] < g_end]
        cap = (g_end - g_begin) // 39780 + 8
        inc = (C.c_uint64 * cap)()
        r = capi.Resolver()
        assert L.adsb_resolver_start_chain(r._h, g_begin, min(g_end, g_begin + 16384)) == 0
        assert L.adsb_resolver_start_walk(r._h, g_begin, g_end, total, inc, cap) == 0
        k = 0
        frames = (capi.Frame * max(1, len(mine)))()
        nf = 0
        while k < len(mine):       # records arrive in batches, each followed by an advance to somewhere behind the batch
            step = int(rng.integers(1, 400))
            batch = mine[k:k + step]
            k += step
            r.feed(batch)
            # everything below g_complete must have been fed: anywhere up to the next record still to come
            r.advance(0, max(g_begin, mine[k][0] - int(rng.integers(0, 3)) * int(rng.integers(0, 20_000))) if k < len(mine) else g_end)
            if drain_between and rng.random() < 0.5:
                part = (capi.Frame * len(frames)).from_address(C.addressof(frames) + nf * C.sizeof(capi.Frame))
                nf += int(L.adsb_resolver_drain(r._h, part, len(frames) - nf))
        r.advance(0, g_end)
        fin = C.c_int(0)
        n_inc = int(L.adsb_resolver_walk_result(r._h, C.byref(fin)))
        part = (capi.Frame * len(frames)).from_address(C.addressof(frames) + nf * C.sizeof(capi.Frame))
        nf += int(L.adsb_resolver_drain(r._h, part, len(frames) - nf))
        hd = capi.ShardHead(g_begin, g_end, nf, 0, 0, 0, 0, 0, 0)
        ref = (C.c_uint64 * cap)()
        n_ref = int(L.adsb_shard_walk(C.byref(hd), frames, total, ref, cap))
        assert n_inc == n_ref > 100 and list(inc[:n_inc]) == list(ref[:n_ref])
        assert fin.value == int(hd.walk_final) and (fin.value == 1) == (g_end == n_off)
        r.close()


def test_host_placement_query_and_mapped_block_registry(capi):
    """numa.cpp without a device: adsb_host_placement says on which node the pages of a range live (one node here: all of
    them on it, or "unknown" where move_pages is not allowed); adsb_host_release_mapped knows its own blocks only;
    adsb_host_alloc_on needs the runtime to page-lock and fails cleanly without one."""
    import ctypes as C
    L = capi.load()
    a = np.ones(1 << 21, np.uint16)
    node, frac = C.c_int(-5), C.c_double(-1.0)
    rc = L.adsb_host_placement(a.ctypes.data, a.nbytes, 0, C.byref(node), C.byref(frac))
    assert rc in (0, -1)
    if rc == 0:
        assert node.value >= 0 and 0.0 <= frac.value <= 1.0
        if not os.path.isdir("/sys/devices/system/node/node1"):
            assert node.value == 0 and frac.value == 1.0
    assert L.adsb_host_placement(None, 16, 0, C.byref(node), C.byref(frac)) == -1
    assert L.adsb_host_release_mapped(a.ctypes.data) == 0      # not one of its mappings
    import torch
    if not torch.cuda.is_available():
        assert not L.adsb_host_alloc_on(1 << 20, 0)            # no runtime, no page-locking: NULL, and nothing leaks


def test_shard_layout_check_and_config_growth(capi):
    """adsb_shard_layout_check: the run-time answer to "was this caller built against the layout the library writes?" (ABI 4
    grew adsb_shard_head from 80 to 136 bytes); adsb_config_init fills what the caller's struct holds, wait_timeout_s included."""
    import ctypes as C
    L = capi.load()
    assert L.adsb_shard_layout_check(C.sizeof(capi.ShardHead), C.sizeof(capi.ShardPart)) == 0
    assert C.sizeof(capi.ShardHead) == 136
    assert L.adsb_shard_layout_check(80, C.sizeof(capi.ShardPart)) == -1          # an ABI-3 caller's head
    assert L.adsb_shard_layout_check(C.sizeof(capi.ShardHead), C.sizeof(capi.ShardPart) - 32) == -1
    cfg = capi.make_config(wait_timeout_s=7)
    assert cfg.struct_size == C.sizeof(capi.Config) and cfg.wait_timeout_s == 7 and cfg.abi == 5


def test_slicer_column_gather_equals_the_definition(tmp_path):
    """csrc/slicer_bits.h -- the word-wide form of the PPM slicer's bit gather the kernel runs since round 6 -- compiled
    for the host and compared, column byte by column byte, with the definition (demod.c:31-44,109: frame bit k of the
    candidate at g is a[g+80+10k] > a[g+85+10k]) for every offset-in-run, on random, sparse, dense and periodic planes."""
    import subprocess
    exe = tmp_path / "slicer_bits"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "cpp", "slicer_bits.cpp"),
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok: 23520000 column bytes"), out.stdout + out.stderr


def test_build_follows_the_headers_the_compiler_read(tmp_path):
    """adsbdec_amd/_build.py rebuilds an object when ANY file its last compile read is newer (-MMD depfiles).  Round 5 kept
    a list of headers by hand and gang.hpp (included by resolver.hpp) was not on it: editing it alone left stale objects."""
    from adsbdec_amd import _build
    _build.build()
    obj = os.path.join(_build.LIBDIR, "host_abi.cpp.o")
    deps = _build._recorded_deps(obj)
    gang = os.path.normpath(os.path.join(_build.CSRC, "gang.hpp"))
    assert deps and gang in deps and os.path.normpath(os.path.join(ROOT, "include", "adsbdec_amd_diag.h")) in deps
    for o in ("scan_kernel.hip.o", "decoder.hip.o", "multi.cpp.o"):
        assert _build._recorded_deps(os.path.join(_build.LIBDIR, o)), o
    assert os.path.normpath(os.path.join(_build.CSRC, "slicer_bits.h")) in _build._recorded_deps(os.path.join(_build.LIBDIR, "scan_kernel.hip.o"))
    assert not _build._stale(obj, os.path.join(_build.CSRC, "host_abi.cpp"))
    st = os.stat(gang)
    try:
        newest = max(os.path.getmtime(os.path.join(_build.LIBDIR, f)) for f in os.listdir(_build.LIBDIR) if f.endswith(".o"))
        os.utime(gang, (st.st_atime, newest + 5))                       # "edited" after the objects were built
        assert _build._stale(obj, os.path.join(_build.CSRC, "host_abi.cpp"))
        assert _build._stale(os.path.join(_build.LIBDIR, "decoder.hip.o"), os.path.join(_build.CSRC, "decoder.hip"))
        assert not _build._stale(os.path.join(_build.LIBDIR, "numa.cpp.o"), os.path.join(_build.CSRC, "numa.cpp"))   # (does not include it)
    finally:
        os.utime(gang, (st.st_atime, st.st_mtime))
