"""GPU tests of the multi-GPU host in C (adsb_multi_*, csrc/multi.cpp) and of the piecewise shard stream it is built on
(adsb_shard_begin / adsb_shard_end).  This pool has one GPU per box, so "N devices" here means N handles and N worker
threads on device 0: everything but the links is exercised -- plan, per-shard copy / scan / chain resolution, windows of
tries, stitch, fallback, gather -- and every result is compared bit for bit with the sequential decode (oracle, and the
real reference chain where oracle/_ref travelled)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import records

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    torch.cuda.set_device(0)
    return torch


def _dev(torch, x):
    return torch.from_numpy(x.view(np.int16)).cuda()


def _back_to_back(n_frames, seed):
    from tools import gen_signal as G
    rng = np.random.default_rng(seed)
    placed = [(10_000 + 2_400 * i, G.make_frame([17, 18, 17, 11][i % 4], rng), float(rng.uniform(500, 1500)), float(i))
              for i in range(n_frames)]
    return G.synth(10_000 + 2_400 * n_frames + 120_000, placed, 6.0, seed)


def _recs(capi, raw):
    return records(capi._frames_to_dicts(raw[0], raw[1]))


@pytest.fixture(scope="module")
def captures(oracle):
    """(name, capture, frames, Try/Ok) of: overlapping frames in noise; frames packed back to back, so that EVERY seam cuts
    through one and the last call's horizon lands inside a run of them; dense noise (thousands of tries per window)."""
    from tools import gen_signal as G
    out = []
    for name, x in (("noisy", G.dense_capture((3 << 20) + 4, seed=91, sigma=40.0, n_frames=1500, amp=(200, 1800))[0]),
                    ("back_to_back", _back_to_back(1300, 45)),
                    ("dense_noise", G.dense_capture((2 << 20) + 1002, seed=92, sigma=300.0, n_frames=400, amp=(1200, 2000))[0])):
        want, wstats = oracle.decode(x, df18=True)
        out.append((name, x, records(want), wstats))
    return out


@pytest.mark.parametrize("handles", [1, 2, 3, 8])
def test_multi_decode_host_equals_the_sequential_decode(capi, captures, torch_cuda, handles):
    """adsb_multi_decode_host, K handles on one device, page-locked and pageable captures, with and without the Try/Ok
    table: frames (g, ts, pw, bytes) and statistics equal to the oracle's sequential decode."""
    from adsbdec_amd import sharding
    for stats in (False, True):
        md = sharding.MultiDecoder(handles, [0] * handles, df18=True, collect_stats=stats)
        try:
            for name, x, want, wstats in captures:
                with capi.PinnedBuffers(1, x.size) as bufs:
                    bufs[0][:] = x
                    for src in (bufs[0], x):            # page-locked, then pageable (the runtime's bounce buffers)
                        got = _recs(capi, md.decode_host(src))
                        assert got == want, (name, handles, stats, len(got), len(want))
                        inf = md.info()
                        assert inf["fallback"] == 0 and inf["shards"] == min(handles, (x.size // 2 - 1195) // (1 << 17))
                        if stats:
                            assert md.stats() == wstats, (name, handles)
        finally:
            md.close()


def test_multi_decode_with_small_pieces_and_small_staging(capi, captures, torch_cuda):
    """The shard stream is fed in pieces (32 MiB by default: one piece for these captures).  With a 256 Ki-sample staging
    buffer the pieces are 128 Ki samples: dozens of copies, scans and compactions per shard, seams between pieces."""
    from adsbdec_amd import sharding
    md = sharding.MultiDecoder(3, [0, 0, 0], df18=True, collect_stats=True, stage_samples=1 << 18)
    try:
        for name, x, want, wstats in captures:
            assert _recs(capi, md.decode_host(x)) == want, name
            assert md.stats() == wstats, name
    finally:
        md.close()


def test_multi_decode_file_and_device(capi, captures, torch_cuda, tmp_path):
    """The same capture as a file (every worker preads its own slice) and as slices resident in HBM (adsb_multi_plan +
    adsb_multi_decode_device: one call and one launch per shard, the path bench.py --mode shard times)."""
    from adsbdec_amd import sharding
    md = sharding.MultiDecoder(4, [0] * 4, df18=True, collect_stats=True)
    try:
        for name, x, want, wstats in captures:
            path = str(tmp_path / f"{name}.u16")
            x.tofile(path)
            assert _recs(capi, md.decode_file(path)) == want, name
            assert md.stats() == wstats
            t = _dev(torch_cuda, x)
            plan = md.plan(x.size)
            assert len(plan) == 4 and plan[0]["first_sample"] == 0
            ptrs = [t.data_ptr() + 2 * p["first_sample"] for p in plan]
            assert _recs(capi, md.decode_device(x.size, ptrs)) == want, name
            assert md.stats() == wstats
        with pytest.raises(sharding.ShardError, match="slices"):
            md.decode_device(captures[0][1].size, ptrs[:3])
        with pytest.raises(sharding.ShardError, match="not a readable regular file"):
            md.decode_file(str(tmp_path / "missing.u16"))
    finally:
        md.close()


@pytest.mark.parametrize("n", [0, 3, 2390, 2392, 81_960, 120_006, 262_144 + 2390, 600_002])
def test_multi_short_captures(capi, oracle, torch_cuda, n):
    """Captures shorter than one window, than the first deqframe call (81 960 samples: nothing is ever visited), than one
    shard's worth: the plan falls back to fewer shards and the answer stays the reference's (frames AND Try/Ok)."""
    from adsbdec_amd import sharding
    from tools import gen_signal as G
    x = G.dense_capture(max(n, 4), seed=7 + n, sigma=60.0, n_frames=max(1, n // 6000), amp=(300, 1800))[0][:n] if n else np.zeros(0, np.uint16)
    want, wstats = oracle.decode(x, df18=True) if n else ([], {"try": {11: 0, 17: 0, 18: 0}, "ok": {11: 0, 17: 0, 18: 0}})
    md = sharding.MultiDecoder(4, [0] * 4, df18=True, collect_stats=True)
    try:
        got = _recs(capi, md.decode_host(np.ascontiguousarray(x)))
        assert got == records(want)
        assert md.stats() == wstats
    finally:
        md.close()


def test_multi_undecidable_seam_falls_back_to_one_stream(capi, captures, torch_cuda):
    """A head window of 600 offsets cannot decide a seam that cuts through back-to-back frames: the stitcher says -3 and the
    driver sends the capture through one handle as an ordinary stream.  Same frames, same table, and info says so."""
    from adsbdec_amd import sharding
    name, x, want, wstats = captures[1]
    fell = 0
    for handles in (2, 5, 8):
        md = sharding.MultiDecoder(handles, [0] * handles, df18=True, collect_stats=True, debug_shard_head=600)
        try:
            assert _recs(capi, md.decode_host(x)) == want
            assert md.stats() == wstats
            fell += md.info()["fallback"]
        finally:
            md.close()
    assert fell > 0, "no cut of this capture needed the fallback: the test does not test"


def test_multi_independent_streams(capi, oracle, torch_cuda, tmp_path):
    """BASELINE configs[3] in one process: N different captures, stream s on worker s mod K, each with its own ts and Try/Ok
    table; more streams than workers too."""
    from adsbdec_amd import sharding
    from tools import gen_signal as G
    xs = [G.dense_capture((1 << 20) + 4 * s, seed=300 + s, sigma=[8.0, 40.0, 300.0][s % 3], n_frames=300 + 50 * s, amp=(300, 1800))[0]
          for s in range(5)]
    wants = [oracle.decode(x, df18=True) for x in xs]
    paths = []
    for s, x in enumerate(xs):
        paths.append(str(tmp_path / f"s{s}.u16"))
        x.tofile(paths[-1])
    for handles in (1, 3, 5):
        md = sharding.MultiDecoder(handles, [0] * handles, df18=True, collect_stats=True)
        try:
            for call, arg in ((md.decode_streams_host, xs), (md.decode_streams_file, paths)):
                call(arg)
                for s in range(5):
                    assert _recs(capi, md.stream_frames(s)) == records(wants[s][0]), (handles, s)
                    assert md.stream_stats(s) == wants[s][1]
            with pytest.raises(sharding.ShardError, match="not a readable regular file"):
                md.decode_streams_file(paths[:2] + [str(tmp_path / "missing.u16")])
        finally:
            md.close()


def test_multi_one_bit_repair_extension(capi, oracle, torch_cuda):
    """cfg.fix_1bit through the sharded path == the oracle's own rule on the whole stream (no reference parity exists for the
    extension); `fixed` is counted from the final frames."""
    from adsbdec_amd import sharding
    from tools import gen_signal as G
    rng = np.random.default_rng(17)
    placed = []
    for i in range(500):
        fr = bytearray(G.make_frame(17, rng))
        if i % 3 == 0:
            k = int(rng.integers(5, 112))
            fr[k >> 3] ^= 0x80 >> (k & 7)
        placed.append((20_000 + 5_300 * i, bytes(fr), float(rng.uniform(400, 1500)), float(i)))
    x = G.synth(20_000 + 5_300 * 500 + 100_000, placed, 10.0, 3)
    want, wstats = oracle.decode(x, df18=True, fix1=True)
    assert wstats["fixed"] > 100
    md = sharding.MultiDecoder(4, [0] * 4, df18=True, collect_stats=True, fix_1bit=True)
    try:
        assert _recs(capi, md.decode_host(x)) == records(want)
        assert md.stats() == wstats
    finally:
        md.close()


@pytest.mark.parametrize("mode", ["push", "push_async", "push_device"])
def test_shard_stream_primitives_equal_the_one_call_scan(capi, captures, torch_cuda, mode):
    """adsb_shard_begin + pushes + adsb_shard_end (a shard fed piecewise through the handle's ordinary stream machinery) gives
    what adsb_scan_shard_resolved_walk gives for the same shard in one call: head, speculative frames, head candidates,
    bases of the deqframe calls, and -- collect_stats -- the shard's own Try count."""
    L = capi.load()
    name, x, _, _ = captures[0]
    t = _dev(torch_cuda, x)
    total = x.size
    d1 = capi.Decoder(df18=True, collect_stats=True)
    d2 = capi.Decoder(df18=True, collect_stats=True, stage_samples=1 << 18)
    every = None
    try:
        for p in capi.plan_shards(total, 5):
            cap = (p["g_end"] - p["g_begin"]) // 39780 + 8
            h1, f1, c1, b1 = capi.ShardHead(), (capi.Frame * 8192)(), (capi.Candidate * 4096)(), (C.c_uint64 * cap)()
            assert L.adsb_scan_shard_resolved_walk(d1._h, t.data_ptr() + 2 * p["first_sample"], p["first_sample"], p["n_samples"],
                                                   p["g_begin"], p["g_end"], total, C.byref(h1), f1, 8192, c1, 4096, b1, cap) == 0
            h2, b2 = capi.ShardHead(), (C.c_uint64 * cap)()
            assert L.adsb_shard_begin(d2._h, p["first_sample"], p["g_begin"], p["g_end"], total, b2, cap) == 0, L.adsb_last_error(d2._h)
            lo, hi = p["first_sample"], p["first_sample"] + p["n_samples"]
            step = 100_003 * 2
            for a in range(lo, hi, step):
                piece = x[a:min(hi, a + step)]
                if mode == "push":
                    d2.push(piece)
                elif mode == "push_async":
                    d2.push_async(np.ascontiguousarray(piece))
                    d2.sync()
                else:
                    d2.push_device(t.data_ptr() + 2 * a, piece.size)
            fp, cp = C.POINTER(capi.Frame)(), C.POINTER(capi.Candidate)()
            assert L.adsb_shard_end(d2._h, C.byref(h2), C.byref(fp), C.byref(cp)) == 0, L.adsb_last_error(d2._h)
            for k, _ in capi.ShardHead._fields_:
                if k == "n_head":   # (see below: the two lists may differ in candidates no chain can reach)
                    continue
                v1, v2 = getattr(h1, k), getattr(h2, k)
                assert (list(v1) == list(v2)) if k in ("tries", "ok") else (v1 == v2), (k, v1, v2)
            assert h1.n_frames > 50 and h1.has_tries == 1 and sum(h1.tries) > 0
            key = lambda f: (int(f.g), int(f.ts), int(f.pw), int(f.len), bytes(f.frame), int(f.reserved))
            assert [key(f1[i]) for i in range(h1.n_frames)] == [key(fp[i]) for i in range(h2.n_frames)]
            # The head candidates are what the device's never-visited filter let through of the first ADSB_SHARD_HEAD (262 144)
            # offsets, and that filter is conservative at tile starts: small pieces make other tiles than one launch does, so
            # the two lists may differ -- in candidates that no chain can reach.  Both are ascending, lie inside the window, are
            # CRC-valid candidates of the capture (the exhaustive list of a handle with all_candidates = 1), and hold every
            # candidate either speculative chain accepted there.
            ckey = lambda c: (int(c.g), int(c.pw), int(c.len), bytes(c.frame))
            hc1, hc2 = [ckey(c1[i]) for i in range(h1.n_head)], [ckey(cp[i]) for i in range(h2.n_head)]
            if every is None:
                d3 = capi.Decoder(df18=True, all_candidates=True)
                cs, nc, _ = d3.scan_shard(t.data_ptr(), 0, total, 0, max(0, total // 2 - 1195), cand_cap=1 << 18)
                every = {ckey(cs[i]) for i in range(nc)}
                d3.close()
            for hc in (hc1, hc2):
                assert hc == sorted(hc) and all(p["g_begin"] <= c[0] < h1.head_end for c in hc) and set(hc) <= every
                accepted = {(int(f1[i].g), int(f1[i].pw), int(f1[i].len), bytes(f1[i].frame)[: f1[i].len].ljust(14, b"\0")) for i in range(h1.n_frames)
                            if f1[i].g < h1.head_end}
                assert {(c[0], c[1], c[2], c[3][: c[2]].ljust(14, b"\0")) for c in hc} >= accepted
            assert abs(h1.n_head - h2.n_head) <= max(8, h1.n_head // 8)
            assert list(b1[: h1.n_bases]) == list(b2[: h2.n_bases])
            # a shard stream refuses what belongs to an ordinary stream, and the other way round
            assert L.adsb_push(d2._h, x.ctypes.data, 8) != 0
        assert L.adsb_shard_begin(d2._h, 0, 0, 28 * 1000, total, None, 0) == 0
        assert L.adsb_finish(d2._h) != 0 and b"adsb_shard_end" in L.adsb_last_error(d2._h)
        fp, cp, h2 = C.POINTER(capi.Frame)(), C.POINTER(capi.Candidate)(), capi.ShardHead()
        assert L.adsb_shard_end(d2._h, C.byref(h2), C.byref(fp), C.byref(cp)) != 0     # (no samples fed)
        assert b"samples" in L.adsb_last_error(d2._h)
        d2.reset()
        assert L.adsb_shard_end(d2._h, C.byref(h2), C.byref(fp), C.byref(cp)) != 0
        assert records(d2.decode(x)) == captures[0][2]                              # the handle is an ordinary one again
    finally:
        d1.close()
        d2.close()


@pytest.mark.parametrize("n_shards", [2, 5, 13])
def test_shard_primitives_with_statistics_any_cut(capi, captures, torch_cuda, n_shards):
    """The primitives under the driver, cut where the driver never would (13 shards of a 3 Mi-sample capture): every shard's
    own Try count from the device + the two windows of tries from adsb_scan_shard, stitched by adsb_stitch_shards_stats."""
    import shard_helpers
    for name, x, want, wstats in captures:
        t = _dev(torch_cuda, x)
        d = capi.Decoder(df18=True, collect_stats=True)
        try:
            ss = shard_helpers.from_device(capi, d, t.data_ptr(), x.size, n_shards, stats=True)
            rc, got, stats, _, _ = ss.stitch(with_stats=True)
            assert rc in (0, -3), rc
            if rc == 0:
                assert got == want and stats == wstats, (name, n_shards)
            else:
                assert n_shards == 13       # shards of ~100 k offsets cannot keep their windows apart
        finally:
            d.close()


def _ref_or_skip(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) did not travel with this snapshot")


def test_cli_sharded_equals_the_real_reference_chain(capi, oracle, tmp_path):
    """`adsbdec_amd_cli -G 0,0,0,0 -f <64 Mi-sample file>` (four handles on this box's one GPU): stdout and the stderr Try/Ok
    table byte-identical to the unpatched reference chain (oracle/_ref/ref_adsbdec), with and without -a, AVR and MLAT; -d,
    and files shorter than one window per shard.  `-G 4` on a box with fewer than four GPUs fails and names the device."""
    _ref_or_skip(oracle)
    from tools import gen_signal as G
    x, _ = G.dense_capture((64 << 20) + 6, seed=411, sigma=30.0, n_frames=9000, amp=(150, 1800))
    big = str(tmp_path / "big.u16")
    x.tofile(big)
    small = str(tmp_path / "small.u16")
    x[:200_000].tofile(small)
    tiny = str(tmp_path / "tiny.u16")
    x[:1000].tofile(tiny)
    import torch
    if torch.cuda.device_count() < 4:
        p = subprocess.run([capi.CLI_PATH, "-G", "4", "-f", tiny], capture_output=True, timeout=600)
        assert p.returncode == 255 and b"adsb_multi_create() failed: device" in p.stderr and b"worker" in p.stderr
    for path, opts in ((big, ["-G", "0,0,0,0"]), (big, ["-G", "0,0,0"]), (big, ["-d", "0"]), (small, ["-G", "0,0,0,0,0,0,0,0"]),
                       (tiny, ["-G", "0,0"])):
        for df18 in (False, True):
            rf, rstats = oracle.ref_decode(None, df18, path=path)
            for flag, key in (([], "avr"), (["-m"], "mlat")) if path == big and opts[0] == "-G" else (([], "avr"),):
                p = subprocess.run([capi.CLI_PATH] + (["-a"] if df18 else []) + flag + opts + ["-f", path], capture_output=True, timeout=600)
                assert p.returncode == 0, p.stderr
                assert p.stdout == b"".join(f[key] for f in rf), (path, opts, df18, key)
                err = p.stderr.decode().splitlines()
                assert [int(v) for v in err[1].split(":")[1].split()] == [rstats["try"][k] for k in (11, 17, 18)], (opts, err)
                assert [int(v) for v in err[2].split(":")[1].split()] == [rstats["ok"][k] for k in (11, 17, 18)]
    assert len(rf) == 0     # (the tiny file: nothing decodes, and both programs say so)


def test_cli_several_captures_one_per_handle(capi, oracle, tmp_path):
    """`-G 0,0,0 -f a -f b -f c -f d`: configs[3] from the C host program; capture k's packets in <file k>.avr, its table on
    stderr -- each equal to what the reference prints for that file alone."""
    _ref_or_skip(oracle)
    from tools import gen_signal as G
    paths = []
    for s in range(4):
        x, _ = G.dense_capture((2 << 20) + 8 * s, seed=500 + s, sigma=20.0 + 90 * s, n_frames=600, amp=(200, 1800))
        paths.append(str(tmp_path / f"cap{s}.u16"))
        x.tofile(paths[-1])
    args = [capi.CLI_PATH, "-a", "-G", "0,0,0"]
    for pth in paths:
        args += ["-f", pth]
    p = subprocess.run(args, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr
    assert p.stdout == b""
    err = p.stderr.decode().splitlines()
    for s, pth in enumerate(paths):
        rf, rstats = oracle.ref_decode(None, True, path=pth)
        assert open(pth + ".avr", "rb").read() == b"".join(f["avr"] for f in rf)
        at = next(i for i, ln in enumerate(err) if ln.startswith(f"== {pth}:"))
        assert f"{len(rf)} frames" in err[at]
        assert [int(v) for v in err[at + 2].split(":")[1].split()] == [rstats["try"][k] for k in (11, 17, 18)]
        assert [int(v) for v in err[at + 3].split(":")[1].split()] == [rstats["ok"][k] for k in (11, 17, 18)]
    # several captures without -G, and -d together with -G, are usage errors like any other unknown combination (main.c:85-87)
    assert subprocess.run([capi.CLI_PATH, "-f", paths[0], "-f", paths[1]], capture_output=True).returncode == 1
    assert subprocess.run([capi.CLI_PATH, "-d", "0", "-G", "2", "-f", paths[0]], capture_output=True).returncode == 1


def test_bench_shard_mode_runs_the_c_driver(tmp_path):
    """`python bench.py --mode shard --gpus K --one-device-test`: ONE process, K handles on this GPU through adsb_multi_*
    (device-resident slices, and a page-locked host capture), gated against the single-handle decode of the same stream."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for extra in (["--gpus", "3"], ["--gpus", "8", "--stats"], ["--gpus", "2", "--shard-source", "host", "--stats"],
                  ["--gpus", "2", "--shard-source", "file"]):
        p = subprocess.run([sys.executable, "bench.py", "--mode", "shard", "--one-device-test", "--samples", str(64 << 20), "--steps", "3",
                            "--warmup", "1", "--preroll-ms", "0"] + extra, cwd=root, capture_output=True, timeout=900, env=env)
        assert p.returncode == 0, p.stderr.decode()[-3000:]
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
        assert len(lines) == 1
        line = json.loads(lines[0])
        assert line["n_gpus"] == int(extra[1]) and line["scaling"] == "strong"
        assert line["config"]["shards"] == int(extra[1]) and line["config"]["fallback_steps"] == 0
        assert line["config"]["parity"].startswith("equal to the single-handle decode") and "CPU path" in line["config"]["parity"]
        assert line["config"]["frames_decoded"] > 3000 and line["config"]["serial_us"] > 0
        if "--stats" in extra:
            assert sum(line["config"]["statistics"]["ok"].values()) == line["config"]["frames_decoded"]


def test_cli_signals_and_a_closed_pipe(capi, oracle, tmp_path):
    """main.c:91-99 on the -f path: SIGPIPE is ignored -- a reader of stdout that goes away makes the program end with status 1
    and the Try/Ok table, not die by signal -- and SIGTERM in the middle of a capture (here: a FIFO that is being fed slowly)
    ends the run in an orderly way: what was pushed is decoded and written (a prefix of the full output), the table is
    printed, status 0."""
    import signal
    import threading
    import time
    from tools import gen_signal as G
    x, _ = G.dense_capture((48 << 20) + 4, seed=611, sigma=25.0, n_frames=9000, amp=(150, 1800))
    path = str(tmp_path / "cap.u16")
    x.tofile(path)
    full = subprocess.run([capi.CLI_PATH, "-a", "-f", path], capture_output=True, timeout=600)
    assert full.returncode == 0 and full.stdout.count(b"\n") > 3000
    # (1) the reader of stdout closes after the first bytes
    p = subprocess.Popen([capi.CLI_PATH, "-a", "-f", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    p.stdout.read(64)
    p.stdout.close()
    err = p.stderr.read().decode()
    assert p.wait(120) == 1, (p.returncode, err)          # not -SIGPIPE
    assert [ln.split(":")[0].strip() for ln in err.splitlines()[-3:]] == ["Try", "Ok", "Total"], err
    # (2) SIGTERM while the capture is still arriving through a FIFO
    fifo = str(tmp_path / "cap.fifo")
    os.mkfifo(fifo)
    raw = x.tobytes()
    fed = {"n": 0}

    def feed():
        try:
            with open(fifo, "wb", buffering=0) as w:
                for at in range(0, len(raw), 4 << 20):
                    w.write(raw[at:at + (4 << 20)])
                    fed["n"] = at + (4 << 20)
                    time.sleep(0.25)
        except BrokenPipeError:
            pass
    p = subprocess.Popen([capi.CLI_PATH, "-a", "-f", fifo], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t = threading.Thread(target=feed, daemon=True)
    t.start()
    time.sleep(3.0)                                        # the runtime is up, some buffers have been read
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=120)
    assert p.returncode == 0, (p.returncode, err)
    err = err.decode()
    assert [ln.split(":")[0].strip() for ln in err.splitlines()[-3:]] == ["Try", "Ok", "Total"], err
    assert len(out) < len(full.stdout) and full.stdout.startswith(out[: out.rfind(b"\n", 0, max(0, len(out) - 4000)) + 1])
    t.join(30)


def test_cli_G_accepts_what_the_one_device_run_accepts(capi, oracle, tmp_path):
    """-G with a single -f that is not a regular file (a FIFO: the reference reads pipes, air.c:224-239) streams it through one
    device, same bytes as the plain run; a file that cannot be opened ends the run silently with an empty table and status
    0 (air.c:225-228), with and without -G."""
    import threading
    from tools import gen_signal as G
    x, _ = G.dense_capture((6 << 20) + 2, seed=612, sigma=25.0, n_frames=900, amp=(150, 1800))
    path = str(tmp_path / "cap.u16")
    x.tofile(path)
    plain = subprocess.run([capi.CLI_PATH, "-a", "-f", path], capture_output=True, timeout=600)
    assert plain.returncode == 0 and plain.stdout.count(b"\n") > 300
    fifo = str(tmp_path / "cap.fifo")
    os.mkfifo(fifo)
    t = threading.Thread(target=lambda: open(fifo, "wb").write(x.tobytes()), daemon=True)
    t.start()
    p = subprocess.run([capi.CLI_PATH, "-a", "-G", "0,0", "-f", fifo], capture_output=True, timeout=600)
    t.join(30)
    assert p.returncode == 0, p.stderr
    assert p.stdout == plain.stdout and p.stderr.splitlines()[-3:] == plain.stderr.splitlines()[-3:]
    for opts in ([], ["-G", "0,0"]):
        q = subprocess.run([capi.CLI_PATH, "-a"] + opts + ["-f", str(tmp_path / "nothing-here.u16")], capture_output=True, timeout=600)
        assert q.returncode == 0 and q.stdout == b"", (opts, q.stderr)
        tail = q.stderr.decode().splitlines()[-3:]
        assert [ln.split(":")[0].strip() for ln in tail] == ["Try", "Ok", "Total"] and tail[-1].split()[-1] == "0"


def test_multi_host_alloc_places_the_capture_and_says_where(capi, captures, torch_cuda):
    """adsb_multi_host_alloc (csrc/numa.cpp): one page-locked array laid out shard by shard on the NUMA node of the device
    that pulls it; the decode from it equals the oracle's, adsb_multi_worker_placement reports for every worker the node of
    its device, whether its thread is bound there and where its slice lives -- on this pool's boxes (one GPU: every handle
    on device 0) every slice must be on the device's node -- and adsb_host_free takes the mapping back.  adsb_host_alloc_on
    is the one-device form."""
    import ctypes as C
    from adsbdec_amd import sharding
    L = capi.load()
    node = L.adsb_device_numa_node(0)
    md = sharding.MultiDecoder(3, [0, 0, 0], df18=True, collect_stats=True)
    try:
        for name, x, want, wstats in captures:
            arr, addr = md.host_alloc(x.size)
            try:
                arr[:] = x
                assert _recs(capi, md.decode_host(addr, x.size)) == want, name
                assert md.stats() == wstats
                inf = md.info()
                for i in range(inf["shards"]):
                    pl = md.placement(i)
                    assert pl["device"] == 0 and pl["device_node"] == node
                    if node >= 0 and pl["slice_node"] >= 0:       # (the platform names the device's node and answers move_pages)
                        assert pl["slice_node"] == node and pl["local_fraction"] == 1.0, (name, i, pl)
                        assert pl["thread_bound"] == 1
            finally:
                md.host_free(addr)
            assert L.adsb_host_release_mapped(addr) == 0          # gone: a second release finds nothing
    finally:
        md.close()
    p = L.adsb_host_alloc_on(3 << 20, 0)
    assert p
    n2, fr = C.c_int(-1), C.c_double(-1.0)
    if L.adsb_host_placement(p, 3 << 20, node, C.byref(n2), C.byref(fr)) == 0 and node >= 0:
        assert n2.value == node and fr.value == 1.0
    d = capi.Decoder(df18=True)
    try:
        name, x, want, _ = captures[0]
        buf = np.frombuffer((C.c_uint16 * (3 << 19)).from_address(p), dtype=np.uint16)
        k = min(buf.size, x.size)
        buf[:k] = x[:k]
        d.push_async((p, k))
        d.finish()
        assert len(d.drain()) > 10
    finally:
        d.close()
        L.adsb_host_free(p)
