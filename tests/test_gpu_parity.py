"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
C-ABI, against the oracle on the same seeded inputs and against the committed
golden fixtures.  Bit-exact is the bar: frame bytes, g, ts, pw, Try/Ok counters.
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import golden_cases, golden_records, load_golden, records, shard_power

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    torch.cuda.set_device(0)
    return torch


@pytest.fixture(scope="module")
def dec_factory(capi, torch_cuda):
    made = []

    def make(**kw):
        d = capi.Decoder(**kw)
        made.append(d)
        return d
    yield make
    for d in made:
        d.close()


def _dev(torch, x):
    return torch.from_numpy(x.view(np.int16)).cuda()


# ------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("name", golden_cases())
def test_golden_host_push(capi, dec_factory, name):
    x, rec = load_golden(name)
    d = dec_factory(df18=rec["df18"], collect_stats=True)
    frames = d.decode(x)
    assert records(frames) == golden_records(rec)
    assert d.stats() == rec["stats"]
    for f, g in zip(frames, rec["frames"]):
        assert capi.format_frame(f, 0) == g["avr"].encode()
        assert capi.format_frame(f, 1) == g["mlat"].encode()
        assert capi.format_frame(f, 2) == bytes.fromhex(g["beast"])


@pytest.mark.parametrize("name", golden_cases())
def test_golden_device_resident(capi, dec_factory, torch_cuda, name):
    x, rec = load_golden(name)
    t = _dev(torch_cuda, x)
    d = dec_factory(df18=rec["df18"], collect_stats=True)
    d.reset()
    d.push_device(t.data_ptr(), t.numel())
    d.finish()
    assert records(d.drain()) == golden_records(rec)
    assert d.stats() == rec["stats"]


@pytest.mark.parametrize("name", golden_cases())
def test_golden_device_final_one_pass(capi, dec_factory, torch_cuda, name):
    """adsb_push_device_final == adsb_push_device + adsb_finish."""
    x, rec = load_golden(name)
    t = _dev(torch_cuda, x)
    d = dec_factory(df18=rec["df18"], collect_stats=True)
    d.reset()
    d.push_device_final(t.data_ptr(), t.numel())
    assert records(d.drain()) == golden_records(rec)
    assert d.stats() == rec["stats"]
    with pytest.raises(capi.AdsbError):
        d.push_device(t.data_ptr(), 8)   # the stream is finished


def test_device_final_after_earlier_pushes(oracle, dec_factory, torch_cuda):
    from tools import gen_signal as G
    x, _ = G.dense_capture((1 << 20) + 6, seed=19, sigma=45.0, n_frames=250)
    want, wstats = oracle.decode(x, df18=True)
    t = _dev(torch_cuda, x)
    d = dec_factory(df18=True, collect_stats=True)
    split = 8 * 40_000
    d.reset()
    d.push_device(t.data_ptr(), split)
    d.push_device_final(t.data_ptr() + 2 * split, t.numel() - split)
    assert records(d.drain()) == records(want)
    assert d.stats() == wstats


# ------------------------------------------------------------------ seeded vs oracle
@pytest.mark.parametrize("seed,sigma,nfr,df18", [(101, 8.0, 80, False), (102, 40.0, 300, True),
                                                 (103, 300.0, 60, True), (104, 120.0, 500, False)])
def test_seeded_vs_oracle(oracle, dec_factory, seed, sigma, nfr, df18):
    from tools import gen_signal as G
    x, _ = G.dense_capture((1 << 20) + 4 * seed, seed=seed, sigma=sigma, n_frames=nfr, amp=(150, 1900))
    want, wstats = oracle.decode(x, df18=df18)
    d = dec_factory(df18=df18, collect_stats=True)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    assert len(want) > 10


def test_full_range_12bit_noise_vs_oracle(oracle, dec_factory):
    """Uniform noise over the whole 12-bit code range: maximal FIR magnitudes, every
    summation-order phase exercised on rounding-sensitive data."""
    rng = np.random.default_rng(7)
    x = rng.integers(0, 4096, 1 << 20, dtype=np.uint16)
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    assert sum(wstats["try"].values()) > 1000


@pytest.mark.parametrize("hi", [24000, 32000])
def test_beyond_12_bit_codes_vs_oracle(oracle, dec_factory, hi):
    """uint16 codes far outside the ADC's 12 bits (|x-2048| up to ~22 k): power sums
    exceed 2^24 (so float truncation is the identity and pair sums round), yet stay
    below 2^31 where the reference's float->int conversion is defined."""
    from tools import gen_signal as G
    rng = np.random.default_rng(hi)
    x = rng.integers(0, hi, 1 << 20, dtype=np.uint16)
    fr = [G.make_frame(17, rng) for _ in range(40)]
    sig = np.zeros(x.size, np.float32)
    for i, f in enumerate(fr):
        s0 = 20_000 + 25_000 * i
        env = G.frame_envelope(f)
        sig[s0:s0 + env.size] += 9000.0 * env * np.cos(np.pi * np.arange(s0, s0 + env.size) / 2 + i)
    x = np.clip(x.astype(np.float32) * 0.05 + 2048 - hi * 0.025 + sig, 0, hi).astype(np.uint16)
    y = rng.integers(0, hi, 1 << 19, dtype=np.uint16)          # second half: raw wide noise
    x = np.concatenate([x, y])
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    a = oracle.power(x)
    assert len(want) >= 30 and float((a[:-10] + a[10:]).max()) > 2.0 ** 24   # pair sums beyond 2^24


@pytest.mark.parametrize("mode", ["sync", "async", "overlap"])
@pytest.mark.parametrize("chunk", [4, 1000, 4096, 65536 + 12, 1 << 18])
def test_chunked_pushes_equal_one_shot(oracle, dec_factory, chunk, mode):
    """The stream is the concatenation of pushes (decodeiq's statics, air.c:33-34,49-50),
    with adsb_push, with the overlapped adsb_push_async (copy of chunk k+1 beside the
    scan of chunk k, frames one call later) and with cfg.push_overlap (adsb_push returns when the
    copy is done; ONE buffer, scribbled over right after every call) alike."""
    from tools import gen_signal as G
    n = 200_000 if chunk < 1000 else 600_000
    x, _ = G.dense_capture(n, seed=5, sigma=35.0, n_frames=100)
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True, push_overlap=(mode == "overlap"))
    assert records(d.decode(x, chunk=chunk, mode=mode)) == records(want)
    assert d.stats() == wstats


@pytest.mark.parametrize("stage", [0, 1 << 16])
def test_push_overlap_at_the_reference_call_size(capi, oracle, dec_factory, torch_cuda, stage):
    """cfg.push_overlap at IQBUFFSZ = 1 Mi samples per call (air.c:218) from one reused buffer, then mixed with
    adsb_push_async, device pushes and an adsb_sync; odd sizes and a small staging buffer (a compaction per piece)."""
    from tools import gen_signal as G
    x, _ = G.dense_capture((5 << 20) + 6, seed=62, sigma=30.0, n_frames=1200, amp=(150, 1800))
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True, stage_samples=stage, push_overlap=True)
    for chunk in (1 << 20, 65546, 300_001):
        assert records(d.decode(x, chunk=chunk, mode="overlap")) == records(want)
        assert d.stats() == wstats
    t = _dev(torch_cuda, x)
    d.reset()
    got = []
    with capi.PinnedBuffers(2, (1 << 20) + 8) as bufs:
        cuts = [0, 1 << 20, (1 << 20) + 4097, 2 << 20, 3 << 20, (3 << 20) + 70_001, 4 << 20, x.size]
        kinds = ["overlap", "async", "overlap", "async", "device", "overlap", "overlap"]
        for k, (a, b, kind) in enumerate(zip(cuts, cuts[1:], kinds)):
            if kind == "device":
                d.push_device(t.data_ptr() + 2 * a, b - a)
            else:
                buf = bufs[k % 2][: b - a]
                buf[:] = x[a:b]
                if kind == "async":
                    d.push_async(buf)
                else:
                    d.push(buf)
                    buf[:] = 0xFFFF
            if k == 3:
                d.sync()
            got += d.drain()
        d.finish()
        got += d.drain()
    assert records(got) == records(want)
    assert d.stats() == wstats


@pytest.mark.parametrize("stage", [0, 1 << 16])
def test_async_pushes_reference_call_size_and_mixed_calls(capi, oracle, dec_factory, torch_cuda, stage):
    """adsb_push_async at the reference's own call size (IQBUFFSZ = 1 Mi samples,
    air.c:218) from two alternating page-locked buffers; then the same stream with
    async, sync and device pushes mixed and an adsb_sync in the middle.  A small staging
    buffer makes every push several pieces."""
    from tools import gen_signal as G
    x, _ = G.dense_capture((5 << 20) + 6, seed=61, sigma=30.0, n_frames=1200, amp=(150, 1800))
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True, stage_samples=stage)
    assert records(d.decode(x, chunk=1 << 20, mode="async")) == records(want)
    assert d.stats() == wstats
    t = _dev(torch_cuda, x)
    d.reset()
    got = []
    with capi.PinnedBuffers(2, (1 << 20) + 8) as bufs:
        cuts = [0, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, (3 << 20) + 70_000, 4 << 20, x.size]
        kinds = ["async", "async", "sync", "async", "device", "async", "async"]
        for k, (a, b, kind) in enumerate(zip(cuts, cuts[1:], kinds)):
            if kind == "device":
                d.push_device(t.data_ptr() + 2 * a, b - a)
            elif kind == "sync":
                d.push(x[a:b])
            else:
                buf = bufs[k % 2][: b - a]
                buf[:] = x[a:b]
                d.push_async(buf)
            if k == 3:
                d.sync()
                n_mid = d.pending()
                assert n_mid > 0
            got += d.drain()
        d.finish()
        got += d.drain()
    assert records(got) == records(want)
    assert d.stats() == wstats


def test_statistics_read_patterns(capi, oracle, dec_factory, torch_cuda):
    """The count passes of a statistics run are prepared when a launch has been resolved and enqueued later (behind
    the next scan launch, or when the table is asked for), on a stream of their own, and an adsb_reset queues its
    clearing behind a pass that is still pending.  Every order of reading, not reading, resetting and destroying
    must give the table of the stream that was decoded last -- here 60 steps over three captures with the table
    read never, once or twice per step, between pushes of one stream too, and handles closed with a pass pending."""
    from tools import gen_signal as G
    rng = np.random.default_rng(77)
    caps = []
    for k in range(3):
        x, _ = G.dense_capture((1 << 20) + 8 * k, seed=500 + k, sigma=[30.0, 120.0, 300.0][k], n_frames=200, amp=(200, 1800))
        want, wstats = oracle.decode(x, df18=True)
        caps.append((x, _dev(torch_cuda, x), records(want), wstats))
    d = dec_factory(df18=True, collect_stats=True)
    for step in range(60):
        x, t, want, wstats = caps[int(rng.integers(0, 3))]
        d.reset()
        mode = int(rng.integers(0, 3))
        if mode == 0:
            d.push_device_final(t.data_ptr(), t.numel())
        elif mode == 1:                                   # two pushes, the table asked for in between (a partial one)
            sp = 8 * int(rng.integers(20_000, x.size // 8 - 20_000))
            d.push_device(t.data_ptr(), sp)
            if rng.random() < 0.5:
                mid = d.stats()
                assert sum(mid["try"].values()) <= sum(wstats["try"].values())
            d.push_device_final(t.data_ptr() + 2 * sp, x.size - sp)
        else:
            d.push(x)
            d.finish()
        assert records(d.drain()) == want
        for _ in range(int(rng.integers(0, 3))):
            assert d.stats() == wstats, f"step {step} mode {mode}"
    x, t, want, wstats = caps[0]
    for _ in range(3):                                    # destroyed with a pass pending
        e = capi.Decoder(df18=True, collect_stats=True)
        e.reset()
        e.push_device_final(t.data_ptr(), t.numel())
        e.close()
    d.reset()
    d.push_device_final(t.data_ptr(), t.numel())
    assert d.stats() == wstats and d.stats() == wstats


def test_accepted_frame_log_regrows(capi, oracle, torch_cuda):
    """Statistics runs: the resolver logs the frames it accepts straight into the page-locked array the count pass uploads
    from; when a pass has more frames than the array holds, the rest goes to a vector and the arrays are regrown.  Start
    with room for 8 frames (cfg.debug_frames_cap) and decode streams with hundreds of frames per
    launch, several streams on one handle, chunked and in one piece: the Try/Ok table must equal the oracle's every time."""
    from tools import gen_signal as G
    d = capi.Decoder(df18=True, collect_stats=True, debug_frames_cap=8)
    try:
        for seed, n, nfr in ((41, 3 << 20, 900), (42, 1 << 20, 300), (43, (2 << 20) + 6, 1500)):
            x, _ = G.dense_capture(n, seed=seed, sigma=30.0, n_frames=nfr, amp=(200, 1800))
            want, wstats = oracle.decode(x, df18=True)
            t = _dev(torch_cuda, x)
            d.reset()
            d.push_device_final(t.data_ptr(), t.numel())
            assert records(d.drain()) == records(want) and d.stats() == wstats
            assert records(d.decode(x, chunk=300_000)) == records(want) and d.stats() == wstats
    finally:
        d.close()


def test_decode_device_is_reset_push_final_take(capi, oracle, dec_factory, torch_cuda):
    """adsb_decode_device: one call per device-resident capture, several captures on one handle (statistics too)."""
    from tools import gen_signal as G
    d = dec_factory(df18=True, collect_stats=True)
    for seed, n in ((11, 1 << 21), (12, (1 << 20) + 6), (13, 90_000), (14, 3 << 20)):
        x, _ = G.dense_capture(n, seed=seed, sigma=30.0, n_frames=n // 5000, amp=(200, 1800))
        want, wstats = oracle.decode(x, df18=True)
        t = _dev(torch_cuda, x)
        p, k = d.decode_device_raw(t.data_ptr(), t.numel())
        assert records(capi._frames_to_dicts(p, k)) == records(want)
        assert d.stats() == wstats and d.drain() == []


def test_stream_of_2_to_32_samples_is_refused(capi, dec_factory, torch_cuda):
    """The reference's sample counter is a uint32_t (air.c:34): at 2^32 samples its ring phase jumps
    and no parity is defined, so the library refuses such a stream -- loudly, before it touches the buffer."""
    d = dec_factory()
    t = torch_cuda.zeros(1 << 16, dtype=torch_cuda.int16, device="cuda")
    d.reset()
    d.push_device(t.data_ptr(), t.numel())
    with pytest.raises(capi.AdsbError, match="2\\^32"):
        d.push_device(t.data_ptr(), (1 << 32) - t.numel())
    h = np.zeros(64, np.uint16)
    with pytest.raises(capi.AdsbError, match="2\\^32"):
        d._check(capi.load().adsb_push(d._h, h.ctypes.data, 1 << 32), "adsb_push")
    d.push_device(t.data_ptr(), t.numel())      # the handle is still usable below the limit
    d.finish()


@pytest.mark.parametrize("passes", [7, 10])
def test_try_counting_with_more_than_64_frames_per_tile(oracle, dec_factory, passes):
    """Statistics runs count the tries on the device, one wave per tile, against the window of accepted frames
    that can shadow the tile's offsets -- one frame per lane.  Short frames packed back to back (640 offsets
    apart: the greedy scan lands exactly on the next preamble, demod.c:128) put 75+ accepted frames into the
    window of a 7-pass tile, more than a wave holds: the per-try binary search takes over.  Tile size forced
    through cfg.debug_passes (launches this small would take 2..6 passes)."""
    from tools import gen_signal as G
    rng = np.random.default_rng(11)
    n = 3 << 19
    frames = [(5_000 + 1_280 * i, G.make_frame(11, rng), float(rng.uniform(600, 1500)), float(rng.uniform(0, 6.28)))
              for i in range((n - 10_000) // 1_280)]
    frames = [f for k, f in enumerate(frames) if k % 97 != 50]        # a few gaps, so that noise tries exist between runs
    x = G.synth(n, frames, 25.0, 5)
    want, wstats = oracle.decode(x, df18=True)
    assert len(want) > 1000 and sum(wstats["try"].values()) > len(want)
    d = dec_factory(df18=True, collect_stats=True, debug_passes=passes)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    t = __import__("torch").from_numpy(x.view(np.int16)).cuda()
    d.reset()
    d.push_device_final(t.data_ptr(), t.numel())
    assert records(d.drain()) == records(want)
    assert d.stats() == wstats


@pytest.mark.parametrize("stats", [False, True])
def test_async_small_staging_seam_is_ordered(capi, dec_factory, stats):
    """Regression for a race found in round 2: with a 64 Ki-sample staging buffer every asynchronous piece
    compacts the buffer, and the next piece's host-to-device copy (copy engine, own stream) lands right behind the
    compaction's tail copy (scan stream) -- inside one cache line.  Unordered, one of the two writes was lost in
    5-35 % of the runs and a frame straddling the seam disappeared (tools/async_race.py reproduces it on a
    -DADSB_TUNING build with the ordering rule switched off).  The copy streams now wait for the tail copy; 150 decodes, odd push sizes included."""
    from tools import gen_signal as G
    x, _ = G.dense_capture((5 << 20) + 6, seed=61, sigma=30.0, n_frames=1200, amp=(150, 1800))
    ref = dec_factory(df18=True, collect_stats=stats)
    want = ref.decode(x)
    wstats = ref.stats() if stats else None
    d = dec_factory(df18=True, collect_stats=stats, stage_samples=1 << 16)
    for i in range(150):
        chunk = (1 << 20) if i % 3 else 65546 + 2 * i
        assert records(d.decode(x, chunk=chunk, mode="async")) == records(want), f"run {i}, chunk {chunk}"
        if stats:
            assert d.stats() == wstats, f"run {i}, chunk {chunk}"


def test_async_pushes_from_pageable_memory(oracle, dec_factory):
    """adsb_push_async does not require page-locked buffers (the runtime then stages the copy itself and
    the overlap is lost, not the result): plain numpy arrays, each kept alive until the next call returned."""
    from tools import gen_signal as G
    x, _ = G.dense_capture((3 << 20) + 2, seed=81, sigma=30.0, n_frames=700, amp=(150, 1800))
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True)
    d.reset()
    got, keep = [], []
    for i in range(0, x.size, 300_001):
        piece = np.ascontiguousarray(x[i:i + 300_001]).copy()
        keep.append(piece)                 # borrowed until the NEXT push returns
        d.push_async(piece)
        got += d.drain()
        del keep[:-2]
    d.finish()
    got += d.drain()
    assert records(got) == records(want)
    assert d.stats() == wstats


def test_reset_with_launches_in_flight(oracle, dec_factory, capi):
    """adsb_reset right after adsb_push_async (scans and copies still running): the old
    stream's records are dropped and the next stream decodes cleanly on the same slots."""
    from tools import gen_signal as G
    xa, _ = G.dense_capture(3 << 20, seed=71, sigma=30.0, n_frames=700)
    xb, _ = G.dense_capture((1 << 20) + 4, seed=72, sigma=50.0, n_frames=300)
    want, wstats = oracle.decode(xb, df18=True)
    d = dec_factory(df18=True, collect_stats=True)
    with capi.PinnedBuffers(1, xa.size) as bufs:
        bufs[0][:] = xa
        for _ in range(3):
            d.reset()
            d.push_async(bufs[0])
            d.reset()                       # launches of the push above are in flight
            assert records(d.decode(xb)) == records(want)
            assert d.stats() == wstats


def test_streaming_frames_available_before_eof(oracle, dec_factory):
    """Frames come out as soon as the reference would have emitted them (each
    deqframe call), not only at adsb_finish."""
    from tools import gen_signal as G
    x, _ = G.sparse_capture(1 << 20, 100, seed=1)
    want, _ = oracle.decode(x)
    d = dec_factory()
    d.reset()
    d.push(x[: 1 << 19])
    early = d.drain()
    assert 0 < len(early) < len(want)
    d.push(x[1 << 19:])
    d.finish()
    assert records(early + d.drain()) == records(want)


@pytest.mark.parametrize("split", [8 * 50_000, 8 * 50_000 + 4, 70_001])
def test_device_pushes_in_place_and_staged(oracle, dec_factory, torch_cuda, split):
    """Two device-resident pushes: an aligned split is scanned in place (seam through
    the staging buffer), an unaligned one is staged; both must equal the oracle."""
    from tools import gen_signal as G
    x, _ = G.dense_capture(1 << 20, seed=9, sigma=45.0, n_frames=250)
    want, wstats = oracle.decode(x, df18=True)
    t = _dev(torch_cuda, x)
    d = dec_factory(df18=True, collect_stats=True)
    d.reset()
    d.push_device(t.data_ptr(), split)
    d.push_device(t.data_ptr() + 2 * split, t.numel() - split)
    d.finish()
    assert records(d.drain()) == records(want)
    assert d.stats() == wstats


def test_small_staging_buffer_many_pieces(oracle, dec_factory, torch_cuda):
    """A 64 Ki-sample staging buffer forces every push through many stage/scan/carry
    cycles (host pushes and unaligned device pushes alike)."""
    from tools import gen_signal as G
    x, _ = G.dense_capture(700_001, seed=23, sigma=40.0, n_frames=150)
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True, stage_samples=1 << 16)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    t = _dev(torch_cuda, x)
    d.reset()
    d.push_device(t.data_ptr() + 2, 3)            # unaligned start: staged copies
    d.push_device(t.data_ptr() + 8, x.size - 4)
    d.finish()
    want2, wstats2 = oracle.decode(x[1:], df18=True)
    assert records(d.drain()) == records(want2)
    assert d.stats() == wstats2


# ------------------------------------------------------------------ edge cases
@pytest.mark.parametrize("n", [0, 1, 3, 4, 2390, 2392, 81_956, 81_960, 81_964])
def test_tiny_and_threshold_lengths(oracle, dec_factory, n):
    """Empty / tiny inputs and lengths around the first deqframe call
    (40980 power samples = 81960 input samples, air.c:94)."""
    from tools import gen_signal as G
    rng = np.random.default_rng(n)
    fr = G.make_frame(17, rng)
    x = G.synth(n, [(1000, fr, 800.0, 1.0)] if n > 4000 else [], 10.0, n) if n else np.empty(0, np.uint16)
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    if n >= 81_960:
        assert len(want) == 1


def test_back_to_back_and_overlapping_frames(oracle, dec_factory):
    """Greedy skip (demod.c:128,134): a frame starting inside an accepted one is never
    reported; one starting right at its end is."""
    from tools import gen_signal as G
    rng = np.random.default_rng(2)
    frs = [G.make_frame(df, rng) for df in (17, 17, 11, 18, 17)]
    s0 = 30_000
    placed = [(s0, frs[0], 900.0, 0.1), (s0 + 2400, frs[1], 900.0, 0.5),      # exactly back to back
              (s0 + 2400 + 2400 + 600, frs[2], 700.0, 0.9),
              (s0 + 9000, frs[3], 500.0, 1.3), (s0 + 9000 + 1100, frs[4], 1500.0, 2.0)]  # overlap
    x = G.synth(1 << 18, placed, 6.0, 3)
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True)
    got = d.decode(x)
    assert records(got) == records(want)
    assert d.stats() == wstats
    assert frs[0] in [f["frame"] for f in got] and frs[1] in [f["frame"] for f in got]


def test_saturated_and_dc_inputs(oracle, dec_factory):
    n = 1 << 18
    for val in (0, 2048, 4095):
        x = np.full(n, val, np.uint16)
        want, wstats = oracle.decode(x, df18=True)
        d = dec_factory(df18=True, collect_stats=True)
        assert records(d.decode(x)) == records(want) == []
        assert d.stats() == wstats


def test_record_buffer_overflow_is_regrown(oracle, dec_factory):
    """More DF-gate passes / loose records than the (shrunken: cfg.debug_*_cap) launch
    buffers hold: the launch is repeated with regrown buffers and nothing is lost or
    delivered twice.  adsb_profile.relaunches proves the path ran."""
    rng = np.random.default_rng(12)
    x = rng.integers(0, 4096, 1 << 22, dtype=np.uint16)   # ~0.65 % of 2 Mi offsets pass the DF gate
    want, wstats = oracle.decode(x, df18=True)
    assert sum(wstats["try"].values()) > 10_000
    # (a) the device-side try list overflows (statistics run of a stream)
    d = dec_factory(df18=True, collect_stats=True, debug_try_cap=1000)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    assert d.profile()["relaunches"] >= 1
    # (b) the loose list overflows: all_candidates sends every record there
    from tools import gen_signal as G
    xb = _back_to_back(600, 43)
    wb, wbstats = oracle.decode(xb, df18=True)
    d = dec_factory(df18=True, all_candidates=True, debug_cand_cap=64)
    assert records(d.decode(xb)) == records(wb) and len(wb) > 400
    assert d.profile()["relaunches"] >= 1


@pytest.mark.parametrize("clist", [1, 2, 7])
def test_staged_list_overflow_and_partial_relaunch(oracle, dec_factory, torch_cuda, clist):
    """cfg.debug_clist_cap shrinks the per-tile staged candidate list: tiles with more
    CRC-valid offsets than that emit the surplus straight to the loose list (cl_over),
    flag their marker, and the launch is finished after completion -- part streamed, part
    gathered.  With the loose list shrunk too, the gather needs a relaunch, after which
    records of tiles that the streamed part had already delivered must not be fed twice."""
    x = _back_to_back(400, 47)
    want, wstats = oracle.decode(x, df18=True)
    assert len(want) > 300
    t = _dev(torch_cuda, x)
    for kw in (dict(), dict(debug_cand_cap=16), dict(collect_stats=True, debug_cand_cap=16, debug_try_cap=64)):
        d = dec_factory(df18=True, debug_clist_cap=clist, **kw)
        for _ in range(2):
            d.reset()
            d.push_device_final(t.data_ptr(), t.numel())
            assert records(d.drain()) == records(want)
            if kw.get("collect_stats"):
                assert d.stats() == wstats
        if "debug_cand_cap" in kw:
            assert d.profile()["relaunches"] >= 1


def test_survivor_queue_overflow_fallback(oracle, dec_factory):
    """With the per-workgroup survivor queue shrunk to 256 entries about half the
    tiles of a noise capture overflow it and are redone in ranges of chunks (a chunk
    that still does not fit: bit position by bit position); results must not change."""
    rng = np.random.default_rng(31)
    x = rng.integers(0, 4096, 1 << 21, dtype=np.uint16)
    want, wstats = oracle.decode(x, df18=True)
    d = dec_factory(df18=True, collect_stats=True, debug_queue_cap=256)
    assert records(d.decode(x)) == records(want)
    assert d.stats() == wstats
    assert sum(wstats["try"].values()) > 5000


def test_overflow_rounds_ranges_and_bit_positions(oracle, dec_factory, torch_cuda):
    """A capture whose survivors are dense AND uneven -- a stretch of nothing but frame starts (7 % of the offsets pass
    the DF gate: ~500 per chunk of 256 runs), then frames packed back to back, then a stretch of both -- against queues of
    256, 512 and 1 024 entries: tiles are redone in ranges of chunks sized from the failed round's count, single chunks
    that still do not fit go bit position by bit position, every round's candidates are staged in the tile's one list.
    Frames and the Try/Ok table must not change, with the list at its normal size and shrunk (candidates go loose)."""
    from tools import gen_signal as G
    rng = np.random.default_rng(77)
    storm = np.tile(G._frame_start_wave(), 1 + 300_000 // 260)[:300_000]
    placed = [(10_000 + 2_400 * i, G.make_frame([17, 18, 17, 11][i % 4], rng), float(rng.uniform(500, 1500)), float(i))
              for i in range(250)]
    back = G.synth(10_000 + 2_400 * 250 + 30_000, placed, 6.0, 5).astype(np.float32) - 2048.0
    sig = np.concatenate([storm, back, storm[:150_000] * 0.5 + back[:150_000], rng.normal(0, 300, 200_000)]).astype(np.float32)
    sig += rng.normal(0, 30, sig.size).astype(np.float32)
    x = np.clip(np.rint(sig + 2048.0), 0, 4095).astype(np.uint16)
    want, wstats = oracle.decode(x, df18=True)
    assert len(want) > 200 and sum(wstats["try"].values()) > 10_000
    t = _dev(torch_cuda, x)
    for qcap in (256, 512, 1024):
        for kw in (dict(), dict(collect_stats=True), dict(collect_stats=True, debug_clist_cap=3), dict(all_candidates=True)):
            d = dec_factory(df18=True, debug_queue_cap=qcap, **kw)
            for _ in range(2):
                d.reset()
                d.push_device_final(t.data_ptr(), t.numel())
                assert records(d.drain()) == records(want), (qcap, kw)
                if kw.get("collect_stats"):
                    assert d.stats() == wstats, (qcap, kw)
    # tiles of 2 .. 7 passes (2 .. 7 chunks of 256 runs): every shape of the range loop
    for passes in (2, 3, 5, 7):
        d = dec_factory(df18=True, collect_stats=True, debug_queue_cap=256, debug_passes=passes)
        d.reset()
        d.push_device_final(t.data_ptr(), t.numel())
        assert records(d.drain()) == records(want), passes
        assert d.stats() == wstats, passes


def test_one_bit_repair_extension_vs_oracle(oracle, dec_factory):
    """cfg.fix_1bit (EXTENSION: the reference has no error correction; no
    reference parity exists) against the oracle's restatement of the same rule; with
    the flag off the very same capture decodes exactly like the reference."""
    from tools import gen_signal as G
    rng = np.random.default_rng(3)
    placed = []
    for i in range(120):
        f = bytearray(G.make_frame([17, 18, 11][i % 3], rng))
        if i % 2:
            k = int(rng.integers(0, 8 * len(f)))      # any bit, also the DF field / short frames
            f[k >> 3] ^= 0x80 >> (k & 7)
        placed.append((20_000 + 6_000 * i, bytes(f), float(rng.uniform(300, 1500)), float(i)))
    x = G.synth(1 << 20, placed, 25.0, 3)
    plain, pstats = oracle.decode(x, df18=True)
    want, wstats = oracle.decode(x, df18=True, fix1=True)
    assert wstats["fixed"] >= 25 and len(want) > len(plain)
    d = dec_factory(df18=True, collect_stats=True, fix_1bit=True)
    got = d.decode(x)
    assert records(got) == records(want)
    assert d.stats() == wstats
    d0 = dec_factory(df18=True, collect_stats=True)
    assert records(d0.decode(x)) == records(plain)
    assert d0.stats() == pstats


def _back_to_back(n_frames, seed):
    from tools import gen_signal as G
    rng = np.random.default_rng(seed)
    placed = [(10_000 + 2_400 * i, G.make_frame([17, 18, 17, 11][i % 4], rng), float(rng.uniform(500, 1500)), float(i))
              for i in range(n_frames)]
    return G.synth(10_000 + 2_400 * n_frames + 120_000, placed, 6.0, seed)


def test_tile_region_overflow_falls_back_to_loose_list(capi, oracle, dec_factory, torch_cuda):
    """Frames packed back to back, all_candidates=1 (every CRC-valid offset is reported,
    ~4 per frame, emitted outside the whole-tile round): the records travel through the
    launch-wide loose list and the launch is finished after completion.  Same frames."""
    x = _back_to_back(1500, 41)
    want, wstats = oracle.decode(x, df18=True)
    assert len(want) > 1000
    t = _dev(torch_cuda, x)
    for kw in (dict(all_candidates=True), dict()):
        d = dec_factory(df18=True, **kw)
        d.reset()
        d.push_device_final(t.data_ptr(), t.numel())
        assert records(d.drain()) == records(want)
    # candidate-level: every CRC-valid offset survives the overflow path
    a = oracle.power(x)
    wc, _ = oracle.scan_all(a, 0, a.size - 1195, True)
    d = dec_factory(df18=True, all_candidates=True)
    cands, nc, _ = d.scan_shard(t.data_ptr(), 0, x.size, 0, a.size - 1195)
    assert [(int(c.g), int(c.pw), bytes(c.frame[: c.len])) for c in cands[:nc]] == wc


def test_streaming_handoff_is_stable_over_many_launches(oracle, dec_factory, torch_cuda):
    """The host consumes a launch's tiles WHILE the kernel runs, out of pinned memory that
    still holds the previous launch's bytes: a tile is taken when its marker's 64-bit check
    word matches the XOR of the record granules behind it, mixed with a per-launch `gen`
    (scan_kernel.h; nothing orders the device's stores).  To be able to fail, the test
    alternates DIFFERENT captures of different sizes on one handle -- the same hand-off
    buffers are rewritten with different records at different positions every launch, so
    a marker or a granule left over from the previous launch is a wrong record here, not
    the same one -- and compares every record of every launch."""
    from tools import gen_signal as G
    caps = []
    for seed, n, nfr in ((55, 1 << 22, 1500), (56, (1 << 22) - 300_000, 900), (57, (1 << 21) + 4096, 1100)):
        x, _ = G.dense_capture(n, seed=seed, sigma=30.0, n_frames=nfr, amp=(150, 1800))
        caps.append((_dev(torch_cuda, x), records(oracle.decode(x, df18=True)[0])))
    d = dec_factory(df18=True)
    for it in range(300):
        t, exp = caps[(it * 7 + it // 5) % 3]
        d.reset()
        d.push_device_final(t.data_ptr(), t.numel())
        assert records(d.drain()) == exp, it


@pytest.mark.parametrize("threads", [2, 5])
def test_reader_thread_consumes_the_handoff_stream(capi, oracle, dec_factory, torch_cuda, threads):
    """cfg.host_threads = 2: a second host thread reads and checks the hand-off stream while the caller resolves behind
    it (decoder.hip StreamReader); cfg.host_threads = 5: that thread, and three more that decide every batch of tiles ahead
    of the caller and write the frames (gang.hpp, Resolver::speculate_tiles).  Same frames, same statistics, on every path a
    launch's collect can take: rotating captures on one handle (stale bytes of the previous launch must not be taken),
    chunked pushes, a statistics run, tiles that flag "finish after completion" (staged-list overflow, loose list, relaunch)."""
    from tools import gen_signal as G
    # also the small launches of this test go through the threads
    rd = dict(host_threads=threads, debug_reader_min_tiles=1, **(dict(debug_gang_min=1) if threads > 2 else {}))
    caps = []
    for seed, n, nfr in ((61, 1 << 22, 1500), (62, (1 << 22) - 300_000, 900), (63, (1 << 21) + 4096, 1100)):
        x, _ = G.dense_capture(n, seed=seed, sigma=30.0, n_frames=nfr, amp=(150, 1800))
        want, wstats = oracle.decode(x, df18=True)
        caps.append((x, _dev(torch_cuda, x), records(want), wstats))
    d = dec_factory(df18=True, **rd)
    for it in range(150):
        _, t, exp, _ = caps[(it * 7 + it // 5) % 3]
        d.reset()
        d.push_device_final(t.data_ptr(), t.numel())
        assert records(d.drain()) == exp, it
    ds = dec_factory(df18=True, collect_stats=True, **rd)
    for x, t, exp, wstats in caps:
        assert records(ds.decode(x)) == exp
        assert ds.stats() == wstats
        ds.reset()
        got = []
        for k in range(0, x.size, 300_000):
            ds.push_async(x[k:k + 300_000])
            got += ds.drain()
        ds.finish()
        got += ds.drain()
        assert records(got) == exp
        assert ds.stats() == wstats
    xb = _back_to_back(400, 47)
    wantb, wstatsb = oracle.decode(xb, df18=True)
    tb = _dev(torch_cuda, xb)
    for kw in (dict(debug_clist_cap=2), dict(debug_clist_cap=2, collect_stats=True, debug_cand_cap=16, debug_try_cap=64),
               dict(all_candidates=True)):
        db = dec_factory(df18=True, **rd, **kw)
        for _ in range(2):
            db.reset()
            db.push_device_final(tb.data_ptr(), tb.numel())
            assert records(db.drain()) == records(wantb)
            if kw.get("collect_stats"):
                assert db.stats() == wstatsb


@pytest.mark.parametrize("name", golden_cases())
def test_golden_without_streaming_handoff(capi, dec_factory, torch_cuda, name):
    """cfg.debug_no_streaming: every launch is collected after completion
    from the launch-wide lists -- the fallback the streaming path drops to on overflow."""
    x, rec = load_golden(name)
    d = dec_factory(df18=rec["df18"], collect_stats=True, debug_no_streaming=True)
    assert records(d.decode(x)) == golden_records(rec)
    assert d.stats() == rec["stats"]
    t = _dev(torch_cuda, x)
    d.reset()
    d.push_device_final(t.data_ptr(), t.numel())
    assert records(d.drain()) == golden_records(rec)
    assert d.stats() == rec["stats"]


@pytest.mark.parametrize("seed,sigma,nfr,df18", [(111, 8.0, 80, False), (112, 300.0, 60, True)])
def test_seeded_without_streaming_handoff(oracle, dec_factory, seed, sigma, nfr, df18):
    from tools import gen_signal as G
    x, _ = G.dense_capture((3 << 20) + 4 * seed, seed=seed, sigma=sigma, n_frames=nfr, amp=(150, 1900))
    want, wstats = oracle.decode(x, df18=df18)
    d = dec_factory(df18=df18, collect_stats=True, debug_no_streaming=True)
    assert records(d.decode(x, chunk=1 << 20)) == records(want)
    assert d.stats() == wstats
    assert records(d.decode(x, chunk=1 << 20, mode="async")) == records(want)


# ------------------------------------------------------------------ against the REAL reference, executed
def _ref_or_skip(oracle):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref (the compiled reference) did not travel with this snapshot")


@pytest.mark.parametrize("seed,n,sigma,nfr,df18", [(201, 1 << 20, 8.0, 100, False), (202, (1 << 21) + 6, 45.0, 500, True),
                                                   (203, 1 << 20, 300.0, 40, True), (204, 3 * 81_960 + 3, 20.0, 60, True)])
def test_hip_equals_real_reference_chain(capi, oracle, dec_factory, seed, n, sigma, nfr, df18):
    """The HIP path against the reference's own code EXECUTED on this box: oracle/_ref/
    ref_adsbdec = air.c:29-101 decodeiq + demod.c + valid.c + output.c:formatpkt, compiled
    from /root/reference in the build container (the binary travels, the sources do not).
    No restatement in between: ts, pw, frame bytes, Try/Ok and all three output formats."""
    _ref_or_skip(oracle)
    from tools import gen_signal as G
    x, _ = G.dense_capture(n, seed=seed, sigma=sigma, n_frames=nfr, amp=(150, 1900))
    rf, rstats = oracle.ref_decode(x, df18)
    d = dec_factory(df18=df18, collect_stats=True)
    got = d.decode(x, chunk=1 << 20)            # the reference's own call size
    assert [(f["ts"], f["pw"], f["frame"]) for f in got] == [(f["ts"], f["pw"], f["frame"]) for f in rf]
    assert d.stats() == rstats
    assert len(rf) > 5
    for f, r in zip(got, rf):
        assert capi.format_frame(f, 0) == r["avr"]
        assert capi.format_frame(f, 1) == r["mlat"]
        assert capi.format_frame(f, 2) == r["beast"]


def test_cli_equals_real_reference_chain_on_a_file(capi, oracle, tmp_path):
    """The C host program on a file against the real chain on the same file: stdout bytes
    (AVR, AVR-MLAT) and the stderr Try/Ok table."""
    _ref_or_skip(oracle)
    from tools import gen_signal as G
    x, _ = G.dense_capture((40 << 20) + 2, seed=207, sigma=25.0, n_frames=4000, amp=(150, 1800))  # > 2 ring buffers
    path = str(tmp_path / "capture.u16")
    x.tofile(path)
    rf, rstats = oracle.ref_decode(None, True, path=path)
    for flag, key in (([], "avr"), (["-m"], "mlat")):
        for env in ({}, {"ADSB_CLI_REGISTER": "0"}):
            p = subprocess.run([capi.CLI_PATH, "-a"] + flag + ["-f", path], capture_output=True, timeout=600,
                               env={**os.environ, **env})
            assert p.returncode == 0, p.stderr
            assert p.stdout == b"".join(f[key] for f in rf)
            err = p.stderr.decode().splitlines()
            assert [int(v) for v in err[1].split(":")[1].split()] == [rstats["try"][k] for k in (11, 17, 18)]
            assert [int(v) for v in err[2].split(":")[1].split()] == [rstats["ok"][k] for k in (11, 17, 18)]


def test_fuzz_30s(oracle):
    """tools/fuzz_parity.py for 30 s: random captures x random ways of feeding them (host
    pushes sync/async, device pushes aligned or not, one-pass final, shard scans), each
    compared with the oracle, every 4th also oracle-vs-real-chain.  The summary line is
    kept (gpurun_out/ -> profiles/)."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools import fuzz_parity
    summary = fuzz_parity.run(30.0, seed=20_000)
    assert summary["captures"] > 50 and summary["frames"] > 1000
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "fuzz_gpu_test_summary.json"), "w") as f:
            json.dump(summary, f, indent=1)


# ------------------------------------------------------------------ sharding on one device
def test_shard_scan_and_host_gather(capi, oracle, dec_factory, torch_cuda):
    """SURVEY 8e with every shard on this one GPU: per-shard stateless scans over the
    planner's halo'd ranges + one host resolver == the sequential reference."""
    from tools import gen_signal as G
    x, _ = G.dense_capture(1 << 21, seed=77, sigma=50.0, n_frames=500)
    want, wstats = oracle.decode(x, df18=True)
    t = _dev(torch_cuda, x)
    d = dec_factory(df18=True, collect_stats=True)
    for n_shards in (1, 3, 8):
        r = capi.Resolver()
        for s in capi.plan_shards(x.size, n_shards):
            cands, nc, tries = d.scan_shard(t.data_ptr() + 2 * s["first_sample"], s["first_sample"],
                                            s["n_samples"], s["g_begin"], s["g_end"])
            r.feed((cands, nc), tries)
        m = 2 * (x.size // 4)
        r.advance(m, m - 1195)
        assert records(r.drain()) == records(want)
        assert r.stats() == wstats


def test_shard_candidates_equal_oracle_exhaustive(capi, oracle, dec_factory, torch_cuda):
    """Candidate-level parity (before resolution): every CRC-valid offset and every
    DF-gate pass the kernel reports equals the oracle's exhaustive evaluation."""
    from tools import gen_signal as G
    x, _ = G.dense_capture(1 << 19, seed=78, sigma=200.0, n_frames=80)
    a = oracle.power(x)
    g_end = a.size - 1195
    wc, wt = oracle.scan_all(a, 0, g_end, True)
    t = _dev(torch_cuda, x)
    d = dec_factory(df18=True, collect_stats=True, all_candidates=True)
    cands, nc, tries = d.scan_shard(t.data_ptr(), 0, x.size, 0, g_end)
    got = [(int(c.g), int(c.pw), bytes(c.frame[: c.len])) for c in cands[:nc]]
    assert got == wc
    assert np.array_equal(tries, wt)
    # default mode: candidates the greedy scan can never visit are dropped on the
    # device; what is left is a subset that resolves to exactly the same frames
    d2 = dec_factory(df18=True, collect_stats=True)
    cands2, nc2, tries2 = d2.scan_shard(t.data_ptr(), 0, x.size, 0, g_end)
    got2 = [(int(c.g), int(c.pw), bytes(c.frame[: c.len])) for c in cands2[:nc2]]
    assert set(got2) <= set(wc) and len(got2) < len(wc)
    assert np.array_equal(tries2, wt)
    want, wstats = oracle.decode(x, df18=True)
    r = capi.Resolver()
    r.feed(got2, tries2)
    r.advance(a.size, g_end)
    assert records(r.drain()) == records(want)
    assert r.stats() == wstats


@pytest.mark.parametrize("n_shards", [2, 5, 8, 13])
def test_resolved_shards_equal_the_sequential_decode(capi, oracle, dec_factory, torch_cuda, n_shards):
    """adsb_scan_shard_resolved_walk + adsb_stitch_shards (every shard resolved on its own, the stitcher only repairs seams,
    hands out ts offsets and applies the end-of-file horizon) against the oracle: a noisy capture with overlapping frames, and
    frames packed back to back so that every seam cuts through one; with a head window too small to decide such a seam
    the stitcher must say so (-3) rather than guess.  (The product's driver of these calls is adsb_multi_*: test_gpu_multi.py.)"""
    import shard_helpers
    from tools import gen_signal as G
    xa, _ = G.dense_capture((3 << 20) + 4, seed=91, sigma=40.0, n_frames=1500, amp=(200, 1800))
    xb = _back_to_back(1100, 45)
    for x in (xa, xb):
        want, _ = oracle.decode(x, df18=True)
        t = _dev(torch_cuda, x)
        d = dec_factory(df18=True)
        rc, got, _, _, _ = shard_helpers.from_device(capi, d, t.data_ptr(), x.size, n_shards).stitch()
        assert rc == 0 and got == records(want)
    # a head window of 600 offsets cannot decide a seam that cuts through back-to-back frames: -3 (or 0 where no seam does)
    t = _dev(torch_cuda, xb)
    d = dec_factory(df18=True, debug_shard_head=600)
    rc, got, _, _, _ = shard_helpers.from_device(capi, d, t.data_ptr(), xb.size, n_shards).stitch()
    assert rc in (0, -3)
    if rc == 0:
        assert got == records(oracle.decode(xb, df18=True)[0])


def _bench_line(extra, timeout=900):
    """`python bench.py ...` from a plain shell -- the driver's command shape: with --gpus N > 1 and no WORLD_SIZE
    the script starts its N ranks itself (before any GPU call) and relays rank 0's JSON line."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "bench.py"] + extra, cwd=root, capture_output=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry rank 0's JSON line and nothing else, got {len(lines)} lines: {lines[:3]}"
    return json.loads(lines[0])


def test_two_rank_independent_streams_are_gated_on_every_rank():
    """BASELINE configs[3] plumbing: `python bench.py --gpus 2` (stream mode), two ranks on this one GPU, each
    with its own stream and handle; EVERY rank compares its frames with the oracle after the timed region."""
    line = _bench_line(["--gpus", "2", "--one-device-test", "--samples", str(16 << 20), "--steps", "3", "--warmup", "1",
                        "--preroll-ms", "0", "--no-extras"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["parity_vs_cpu"] is True and line["config"]["ranks_gated"] == 2
    assert line["config"]["frames_decoded_rank0"] > 500
    assert line["cpu_baseline"]["value"] > 0


# ------------------------------------------------------------------ size-independent properties
def test_eight_ranks_on_one_device(capi):
    """The driver's command shape at N = 8 (`python bench.py --gpus 8`: eight processes, one stream and one handle each, gloo
    for the barriers / the max of the times / the AND of the parity flags), with every rank on this box's one GPU: the
    plumbing an 8-GPU node runs, minus the other seven devices.  Every rank gates its captures against the oracle."""
    line = _bench_line(["--gpus", "8", "--one-device-test", "--samples", str(8 << 20), "--steps", "3", "--warmup", "1",
                        "--preroll-ms", "0", "--no-extras"], timeout=1200)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert line["config"]["parity_vs_cpu"] is True and line["config"]["ranks_gated"] == 8
    assert line["cpu_baseline"]["value"] > 0


def test_a_rank_that_dies_ends_the_job(capi):
    """Rank 1 of 3 vanishes (os._exit, no goodbye -- what an OOM kill looks like) right in front of the timed region.  The job
    must end non-zero within the deadline instead of hanging in a barrier, and the report must name the rank."""
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "3", "--one-device-test", "--samples", str(8 << 20), "--steps", "3",
                        "--warmup", "1", "--preroll-ms", "0", "--no-extras", "--die-rank", "1"], cwd=root, capture_output=True,
                       timeout=600, env=env)
    assert p.returncode != 0
    assert time.time() - t0 < 300
    err = p.stderr.decode()
    assert ("rank      : 1" in err or "rank: 1" in err or "local_rank: 1" in err) and "exitcode  : 9" in err.replace("exitcode: 9", "exitcode  : 9"), err[-2500:]
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.strip().startswith("{")], "no JSON line may come out of a failed job"


def test_round_trip_at_scale(dec_factory, torch_cuda):
    """64 Mi samples generated on the device: every injected, well-separated frame
    must come back exactly once, in order, with ts == g+1-skipped (a checksum of the
    whole greedy replay), independent of any oracle."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.gen_signal import make_workload
    torch = torch_cuda
    n = 64 << 20
    t, truth = make_workload(torch, n, n_frames=3000, seed=5, sigma=8.0)
    d = dec_factory()
    d.reset()
    d.push_device(t.data_ptr(), t.numel())
    d.finish()
    got = d.drain()
    horizon_g = None
    sent = [fr for _, fr in truth]
    got_frames = [f["frame"] for f in got]
    # every decoded frame is one we sent, in the order sent
    it = iter(sent)
    assert all(any(fr == s for s in it) for fr in got_frames)
    # all but the EOF tail are recovered (the last ~41k power samples are unscanned)
    assert len(got) >= len([s for s, _ in truth if s // 2 < n // 2 - 45_000]) - 2
    skipped = 0
    for f in got:
        assert f["ts"] == f["g"] + 1 - skipped
        skipped += 80 + 80 * len(f["frame"]) - 1


# ------------------------------------------------------------------ the C host program
def test_cli_matches_golden_avr_mlat_beast(capi, tmp_path):
    x, rec = load_golden("mixed_df_a_384Ki")
    path = tmp_path / "in.bin"
    x.tofile(path)
    for flag, key in (([], "avr"), (["-m"], "mlat"), (["-b"], "beast")):
        p = subprocess.run([capi.CLI_PATH, "-a"] + flag + ["-f", str(path)], capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr
        if key == "beast":
            want = b"".join(bytes.fromhex(f["beast"]) for f in rec["frames"])
        else:
            want = "".join(f[key] for f in rec["frames"]).encode()
        assert p.stdout == want
        err = p.stderr.decode().splitlines()
        assert [int(v) for v in err[1].split(":")[1].split()] == [rec["stats"]["try"][k] for k in (11, 17, 18)]
        assert [int(v) for v in err[2].split(":")[1].split()] == [rec["stats"]["ok"][k] for k in (11, 17, 18)]
    # unknown flags print the usage text and exit 1 (main.c:85-87)
    assert subprocess.run([capi.CLI_PATH, "-e"], capture_output=True).returncode == 1


def test_cli_tcp_sinks_carry_the_same_packets(capi, tmp_path):
    """-s (connect) and -l (listen), the reference's outmode 1 / 2 (main.c:65-72, output.c:59-157): the packets a loopback
    peer receives are the golden capture's AVR / MLAT / Beast bytes (minted through the reference's own formatpkt: over a
    socket the reference writes Beast with its real length, output.c:318-320), stdout stays empty, stderr says
    "connected" ("listening" first with -l) and then prints the Try/Ok table; one device and sharded over two handles."""
    import socket
    import threading
    x, rec = load_golden("mixed_df_a_384Ki")
    path = tmp_path / "in.bin"
    x.tofile(path)

    def want_bytes(key):
        if key == "beast":
            return b"".join(bytes.fromhex(f["beast"]) for f in rec["frames"])
        return "".join(f[key] for f in rec["frames"]).encode()

    def table_ok(lines):
        assert [int(v) for v in lines[1].split(":")[1].split()] == [rec["stats"]["try"][k] for k in (11, 17, 18)]
        assert [int(v) for v in lines[2].split(":")[1].split()] == [rec["stats"]["ok"][k] for k in (11, 17, 18)]

    for flag, key in (([], "avr"), (["-m"], "mlat"), (["-b"], "beast")):
        for opts in ([], ["-G", "0,0"]):
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.bind(("127.0.0.1", 0))
            srv.listen(1)
            got = []

            def serve():
                c, _ = srv.accept()
                with c:
                    while True:
                        b = c.recv(1 << 16)
                        if not b:
                            break
                        got.append(b)
            t = threading.Thread(target=serve, daemon=True)
            t.start()
            p = subprocess.run([capi.CLI_PATH, "-a"] + flag + opts + ["-s", f"127.0.0.1:{srv.getsockname()[1]}", "-f", str(path)],
                               capture_output=True, timeout=300)
            t.join(30)
            srv.close()
            assert p.returncode == 0, p.stderr
            assert p.stdout == b"" and b"".join(got) == want_bytes(key), (key, opts)
            err = p.stderr.decode().splitlines()
            assert err[0] == "connected"
            table_ok(err[1:])
    # -l: the program listens, this test is the peer
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.Popen([capi.CLI_PATH, "-a", "-m", "-l", f"127.0.0.1:{port}", "-f", str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.stderr.readline() == b"listening\n"
    c = socket.create_connection(("127.0.0.1", port), timeout=30)
    data = b""
    while True:
        b = c.recv(1 << 16)
        if not b:
            break
        data += b
    c.close()
    out, err = p.communicate(timeout=120)
    assert p.returncode == 0 and out == b"" and data == want_bytes("mlat")
    err = err.decode().splitlines()
    assert err[0] == "connected"
    table_ok(err[1:])
    # an unusable address ends the run before anything is decoded (runOutput() == -1 -> 255); -s with several captures is a usage error
    p = subprocess.run([capi.CLI_PATH, "-s", "[::1", "-f", str(path)], capture_output=True, timeout=60)
    assert p.returncode == 255 and p.stderr == b"Invalid IPV6 address\n" and p.stdout == b""
    assert subprocess.run([capi.CLI_PATH, "-G", "0,0", "-s", "127.0.0.1:9", "-f", str(path), "-f", str(path)], capture_output=True).returncode == 1


def test_a_launch_that_ends_in_small_tiles(oracle, dec_factory, torch_cuda):
    """A large launch ends in tiles of four passes (scan_kernel.h tile_passes, choose_big_tiles): the kernel, the count pass
    and the host's walk of the hand-off stream share one geometry.  Forced on a small capture (adsb_debug_config.passes /
    .big_tiles: the first N tiles are whole, the rest small; N beyond the launch's tiles = none small) and compared with the
    oracle, statistics included.  (Not the host's default: the kernel is 1.5-2 % faster with such a tail and the call 3 % slower,
    scan_kernel.h choose_big_tiles.)"""
    from tools import gen_signal as G
    x, _ = G.dense_capture(1 << 22, seed=77, sigma=25.0, n_frames=1200, amp=(150, 1800))
    want, wstats = oracle.decode(x, df18=True)
    t = _dev(torch_cuda, x)
    for passes, big in ((5, 8), (6, 1), (7, 3), (5, 60), (4, 2)):
        for stats in (False, True):
            d = dec_factory(df18=True, collect_stats=stats, debug_passes=passes, debug_big_tiles=big)
            d.reset()
            d.push_device_final(t.data_ptr(), t.numel())
            assert records(d.drain()) == records(want)
            if stats:
                assert d.stats() == wstats


def test_two_streams_interleaved_on_one_gpu(oracle, dec_factory, torch_cuda):
    """one adsb_decoder per stream, several may live in one process.  Two
    handles (own HIP streams, launch slots and resolvers) are fed their captures in
    interleaved pushes of unequal sizes -- host pushes into one, device pushes into the
    other -- and each must equal its own stream decoded alone; then both are reset and
    swap captures."""
    from tools import gen_signal as G
    xa, _ = G.dense_capture(1 << 21, seed=201, sigma=20.0, n_frames=500, amp=(150, 1800))
    xb, _ = G.dense_capture((1 << 21) + 4096, seed=202, sigma=60.0, n_frames=300, amp=(200, 1500))
    wa, sa = oracle.decode(xa, df18=True)
    wb, sb = oracle.decode(xb, df18=True)
    da = dec_factory(df18=True, collect_stats=True)
    db = dec_factory(df18=True, collect_stats=True)
    for first, second, wf, ws, sf, ss in ((xa, xb, wa, wb, sa, sb), (xb, xa, wb, wa, sb, sa)):
        da.reset()
        db.reset()
        t2 = _dev(torch_cuda, second)
        got_a, got_b = [], []
        pa = pb = 0
        rng = np.random.default_rng(9)
        while pa < first.size or pb < second.size:
            na = int(min(first.size - pa, rng.integers(1, 300_000)))
            if na:
                da.push(first[pa: pa + na])
                pa += na
                got_a += da.drain()
            nb = int(min(second.size - pb, 8 * rng.integers(1, 40_000)))
            if nb:
                db.push_device(t2.data_ptr() + 2 * pb, nb)
                pb += nb
                got_b += db.drain()
        da.finish()
        db.finish()
        got_a += da.drain()
        got_b += db.drain()
        assert records(got_a) == records(wf) and da.stats() == sf
        assert records(got_b) == records(ws) and db.stats() == ss


def capi_frames(p, n):
    from adsbdec_amd import capi
    return capi._frames_to_dicts(p, n)


def test_take_is_drain_without_the_copy(oracle, dec_factory):
    """adsb_take hands out the queued frames in place; mixing it with adsb_drain and with
    further pushes must neither lose nor repeat a frame."""
    from tools import gen_signal as G
    x, _ = G.dense_capture(1 << 21, seed=303, sigma=20.0, n_frames=400, amp=(150, 1800))
    want, _ = oracle.decode(x, df18=True)
    d = dec_factory(df18=True)
    got = []
    cuts = [0, 300_000, 300_004, 900_000, 1_500_000, x.size]
    for i, (a, b) in enumerate(zip(cuts, cuts[1:])):
        d.push(x[a:b])
        if i % 2:
            got += d.drain()
        else:
            p, n = d.take_raw()
            got += capi_frames(p, n)
            assert d.take_raw()[1] == 0 and d.drain() == []
    d.finish()
    p, n = d.take_raw()
    got += capi_frames(p, n)
    assert records(got) == records(want)


def test_config_of_another_abi_is_refused_by_name_and_the_threads_are_visible(capi, oracle, torch_cuda):
    """ABI 5 (round 6) on the device: a struct that carries another `abi` -- what a binary built against rounds 4-5's header
    hands over, its df18 lying where `abi` is -- is refused with a message that says so (not misread); the legacy symbol
    adsb_config_default leaves such a struct behind; adsb_profile reports the handle's own threads (none by default under
    sparse traffic, reader + gang with host_threads = 5, never any with host_threads = 1), and the caller's struct may be
    shorter than the library's (adsb_get_profile_sized fills what fits)."""
    import ctypes as C
    L = capi.load()
    cfg = capi.make_config(df18=True)
    cfg.abi = 1                                            # (an ABI-4 binary's df18 = 1)
    assert not L.adsb_create(C.byref(cfg))
    msg = L.adsb_last_error(None)
    assert b"adsb_config.abi" in msg and b"rebuilt" in msg
    legacy = (C.c_uint8 * 128)()
    L.adsb_config_default(legacy)
    assert not L.adsb_create(legacy) and b"adsb_config.abi" in L.adsb_last_error(None)
    x, _ = __import__("tools.gen_signal", fromlist=["dense_capture"]).dense_capture(1 << 20, seed=77, sigma=40.0, n_frames=150, amp=(300, 1800))
    want, _ = oracle.decode(x, df18=True)
    for ht, threads in ((0, 0), (1, 0), (2, 1), (5, 4)):
        d = capi.Decoder(df18=True, host_threads=ht, profile=True)
        try:
            assert records(d.decode(x)) == records(want), ht
            p = d.profile()
            assert p["host_threads_running"] == threads, (ht, p)
            assert p["launches"] >= 1 and (p["gang_launches"] > 0) == (ht >= 3), (ht, p)
            short = capi.Profile()
            C.memset(C.byref(short), 0xEE, C.sizeof(short))
            assert L.adsb_get_profile_sized(d._h, C.byref(short), capi.Profile.host_threads_running.offset) == 0   # an ABI-4-sized struct
            assert short.launches == p["launches"] and short.host_threads_running == 0xEEEEEEEE                  # ... nothing written behind it
        finally:
            d.close()
