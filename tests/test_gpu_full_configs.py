"""BASELINE.json's configurations at their STATED single-GPU sizes, through the C-ABI,
bit for bit against the oracle (and, where the compiled reference travelled with the
snapshot, against the real chain too):

  configs[1]  256 Mi samples, sparse frames (~1 k frames/s)           one stream, one pass
  configs[2]  256 Mi samples of dense noise (~7 % preamble hits), -a  statistics included
  configs[4]  ONE 2 Gi-sample stream time-sharded 8 ways  all 8 shards on this one
              GPU, one host resolver; 1-bit repair extension off (reference parity) and on

configs[0] is tests/golden/config1_1Mi_100xDF17 (every test file uses it); configs[3]
(8 independent streams) is configs[1] once per GPU -- bench.py --gpus N.
These take a minute or two of CPU oracle time in all; they are ordinary `gpu` tests.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, records

pytestmark = pytest.mark.gpu


sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    torch.cuda.set_device(0)
    return torch


def _host(t):
    return t.cpu().numpy().view(np.uint16)


def _threads_named(name):
    n = 0
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/comm") as f:
                n += f.read().strip() == name
        except OSError:
            pass
    return n


def _reader_threads():
    """Threads of this process named "adsb-reader" (handoff.hpp StreamReader: a handle's second host thread)."""
    return _threads_named("adsb-reader")


def _format_threads():
    """... and "adsb-format" (gang.hpp FormatGang: the threads that decide batches ahead and write the frames)."""
    return _threads_named("adsb-format")


def _ts_checksum(frames):
    """ts == g + 1 - sum(span - 1): a checksum of the whole greedy replay (demod.c:86,99,128,134)."""
    skipped = 0
    for f in frames:
        assert f["ts"] == f["g"] + 1 - skipped
        skipped += 80 + 80 * len(f["frame"]) - 1


def test_config1_256Mi_sparse(capi, oracle, torch_cuda):
    from tools.gen_signal import make_workload
    n = (256 << 20) - (256 << 20) % 28
    t, truth = make_workload(torch_cuda, n, seed=1)
    d = capi.Decoder(df18=False, collect_stats=True)
    try:
        d.push_device_final(t.data_ptr(), t.numel())
        got, gstats = d.drain(), d.stats()
        x = _host(t)
        want, wstats = oracle.decode(x, df18=False)
        assert records(got) == records(want)
        assert gstats == wstats
        assert len(got) > 12_000
        _ts_checksum(got)
        sent = iter(fr for _, fr in truth)       # every decoded frame is one that was sent, in order
        assert all(any(f["frame"] == s for s in sent) for f in got)
        if oracle.ref_available():               # the real reference chain on the same 512 MiB
            path = "/dev/shm/adsb_cfg1.u16" if os.access("/dev/shm", os.W_OK) else "/tmp/adsb_cfg1.u16"
            x.tofile(path)
            try:
                rf, rstats = oracle.ref_decode(None, False, path=path)
            finally:
                os.unlink(path)
            assert [(f["ts"], f["pw"], f["frame"]) for f in got] == [(f["ts"], f["pw"], f["frame"]) for f in rf]
            assert gstats == rstats
        # the same capture through the overlapped host path, 1 Mi samples per call (air.c:218)
        d2 = capi.Decoder(df18=False)
        try:
            assert records(d2.decode(x[: 64 << 20], chunk=1 << 20, mode="async")) == \
                records(oracle.decode(x[: 64 << 20], df18=False)[0])
        finally:
            d2.close()
    finally:
        d.close()


def test_config2_256Mi_dense_noise(capi, oracle, torch_cuda):
    from tools.gen_signal import make_dense
    n = (256 << 20) - (256 << 20) % 28
    t = make_dense(torch_cuda, n, 100)   # sigma = 300 noise + one strong 112-bit frame per ms
    d = capi.Decoder(df18=True, collect_stats=True)
    try:
        d.push_device_final(t.data_ptr(), t.numel())
        got, gstats = d.drain(), d.stats()
        want, wstats = oracle.decode(_host(t), df18=True)
        assert records(got) == records(want)
        assert gstats == wstats
        assert sum(wstats["try"].values()) > 500_000      # ~0.65 % of 128 Mi offsets pass the DF gate
        assert len(got) > 3_000                            # a good part of the 13 k frames survives the noise
        _ts_checksum(got)
    finally:
        d.close()


def test_config2_256Mi_at_ten_percent_density_and_the_gate_storm(capi, oracle, torch_cuda):
    """BASELINE configs[2] at the density it states (~10 % of the offsets pass the preamble test: 112-bit frames packed back to
    back in sigma = 300 noise + slots of frame starts), and the adversarial capture made of frame starts only (7 % of ALL offsets
    pass the DF gate: every tile overflows its survivor queue and is redone in ranges of chunks, the launch-wide try list is regrown):
    frames, ts, pw and the Try/Ok table equal to the oracle's on the full 256 Mi samples."""
    from conftest import preamble_pass_fraction
    from tools.gen_signal import make_dense10, make_gate_storm
    n = (256 << 20) - (256 << 20) % 28
    for make, seed, lo, hi, min_frames in ((make_dense10, 101, 0.095, 0.108, 80_000), (make_gate_storm, 102, 0.25, 0.35, 0)):
        t = make(torch_cuda, n, seed)
        x = _host(t)
        assert lo < preamble_pass_fraction(x) < hi
        want, wstats = oracle.decode(x, df18=True)
        assert len(want) >= min_frames
        for stats in (False, True):
            assert _reader_threads() == 0 and _format_threads() == 0
            d = capi.Decoder(df18=True, collect_stats=stats, profile=True)
            try:
                # three times on one handle: cfg.host_threads = 0 (auto) hands the stream of a launch that FOLLOWS a dense one
                # (65 536 records or more) to the handle's reader thread and its gang of four (batches decided ahead, frames
                # written by the gang), which are started then -- and gives that launch tiles of six passes instead of seven
                # (scan_kernel.hip choose_passes: a dense channel); same records either way
                for rep in range(3):
                    d.reset()
                    d.push_device_final(t.data_ptr(), t.numel())
                    got = d.drain()
                    assert records(got) == records(want), rep
                    if stats:
                        assert d.stats() == wstats, rep
                    extra = (_reader_threads(), _format_threads())
                    if make is make_dense10:        # 123 k records per launch
                        gang = 4 if len(os.sched_getaffinity(0)) >= 12 else 0   # (a confined process keeps to the reader: decoder.hip)
                        assert extra == ((0, 0) if rep == 0 else (1, gang)), (rep, extra)
                    else:                            # one record: the threads never exist
                        assert extra == (0, 0)
                if make is make_gate_storm:
                    assert sum(wstats["try"].values()) > 0.05 * (n // 2)
            finally:
                d.close()
            assert _reader_threads() == 0 and _format_threads() == 0
        del t
        torch_cuda.cuda.empty_cache()


def test_config4_2Gi_one_stream_in_8_shards(capi, oracle, torch_cuda):
    """contiguous shards starting on multiples of 28 offsets, halo = 8 pairs
    before + one 1196-sample window after (adsb_plan_shards), stateless per-shard scans
    (adsb_scan_shard), candidates gathered in shard order into ONE resolver that replays
    the sequential rules (greedy skip demod.c:128,134; ts demod.c:86,99; deqframe call
    pattern and EOF horizon air.c:94-99).  Here all eight shards run on this GPU."""
    from tools.gen_signal import make_workload
    n = (2 << 30) - (2 << 30) % 28
    t, _ = make_workload(torch_cuda, n, seed=9, damage_share=0.2)
    x = _host(t)
    for fix in (False, True):
        want, wstats = oracle.decode(x, df18=True, fix1=fix)
        d = capi.Decoder(df18=True, collect_stats=True, fix_1bit=fix)
        r = capi.Resolver()
        try:
            for s in capi.plan_shards(x.size, 8):
                cands, nc, tries = d.scan_shard(t.data_ptr() + 2 * s["first_sample"], s["first_sample"],
                                                s["n_samples"], s["g_begin"], s["g_end"])
                r.feed((cands, nc), tries)
            m = 2 * (x.size // 4)
            r.advance(2 * ((x.size + 3) // 4), m - 1195)
            got = r.drain()
            assert records(got) == records(want)
            st = r.stats()
            assert st["try"] == wstats["try"] and st["ok"] == wstats["ok"]
            assert len(got) > 80_000
            if fix:
                assert wstats["fixed"] > 5_000 and sum(1 for _ in got) > n_plain
            else:
                n_plain = len(got)
            _ts_checksum(got)
            # the scalable form of the same job, through the product's own driver: adsb_multi_decode_device with 8 handles
            # on this GPU -- every shard resolved while its kernels run AND counting its own tries on the device, then only
            # seams, ts offsets, the end-of-file horizon and the corrections of the Try table (adsb_stitch_shards_stats)
            from adsbdec_amd import sharding
            md = sharding.MultiDecoder(8, [0] * 8, df18=True, fix_1bit=fix, collect_stats=True)
            try:
                plan = md.plan(x.size)
                assert len(plan) == 8
                raw = md.decode_device(x.size, [t.data_ptr() + 2 * p["first_sample"] for p in plan])
                assert records(capi._frames_to_dicts(*raw)) == records(want)
                assert md.stats() == wstats
                inf = md.info()
                assert inf["fallback"] == 0 and inf["serial_us"] < 20_000   # stitch + the gather of ~100 k frames, microseconds
            finally:
                md.close()
        finally:
            d.close()
            r.close()
    del t, x, want
    torch_cuda.cuda.empty_cache()


def test_config2_dense_traffic_through_the_multi_gpu_driver(capi, oracle, torch_cuda):
    """BASELINE configs[2]'s traffic (a channel at its capacity: 112-bit frames packed back to back, ~10 % of the offsets pass
    the preamble test) through the multi-GPU driver: configs[4] x configs[2] -- ONE 512 Mi-sample capture resident in HBM cut
    into 1 / 4 / 8 shards, the Try/Ok table included -- and configs[3] x configs[2] -- four independent dense captures on two
    workers.  Frames, ts, pw and the table equal to the oracle's; and the helper threads are VISIBLE: every worker's handle
    reports its reader thread and gang (adsb_profile.host_threads_running through adsb_multi_worker_profile, the sum in
    adsb_multi_info.helper_threads), none exists with cfg.host_threads = 1.  (demod.c:125-141 is what makes a shard's host
    side sequential; main.c:60-89 is N x -f.)"""
    from adsbdec_amd import sharding
    from tools.gen_signal import make_dense10
    n = (512 << 20) - (512 << 20) % 28
    t = make_dense10(torch_cuda, n, 131)
    x = _host(t)
    want, wstats = oracle.decode(x, df18=True)
    want_r = records(want)
    assert len(want) > 180_000
    roomy = len(os.sched_getaffinity(0)) >= 12
    assert _reader_threads() == 0 and _format_threads() == 0
    # handles, cfg.host_threads, helper threads per worker once the traffic has been seen: auto (0) starts reader + 4 behind a
    # launch of >= 65 536 records -- a 512 Mi-sample shard has two such launches, a 128 Mi-sample one stays below -- explicit 6
    # starts them at adsb_create, 1 never starts any
    # (round 6: "near its capacity" is a density -- a record per 2 048 offsets -- so the 128 Mi-sample shards of four workers,
    # 61 k records per launch, start their helpers by themselves too)
    for handles, host_threads, per_worker in ((1, 0, 5 if roomy else 1), (4, 0, 5 if roomy else 1), (4, 6, 5), (8, 6, 5), (4, 1, 0)):
        md = sharding.MultiDecoder(handles, [0] * handles, df18=True, collect_stats=True, host_threads=host_threads, profile=True)
        try:
            plan = md.plan(n)
            assert len(plan) == handles
            for rep in range(2):
                raw = md.decode_device(n, [t.data_ptr() + 2 * p["first_sample"] for p in plan])
                assert records(capi._frames_to_dicts(*raw)) == want_r, (handles, host_threads, rep)
                assert md.stats() == wstats, (handles, host_threads, rep)
            inf = md.info()
            assert inf["fallback"] == 0
            profs = [md.worker_profile(i) for i in range(handles)]
            assert [p["host_threads_running"] for p in profs] == [per_worker] * handles, (handles, host_threads, profs)
            assert inf["helper_threads"] == per_worker * handles
            assert (_reader_threads(), _format_threads()) == ((handles, 4 * handles) if per_worker == 5 else (handles if per_worker else 0, 0))
            if per_worker == 5:
                assert all(p["gang_launches"] > 0 for p in profs), profs
            else:
                assert all(p["gang_launches"] == 0 and p["gang_batches"] == 0 for p in profs)
        finally:
            md.close()
        assert _reader_threads() == 0 and _format_threads() == 0
    del t
    torch_cuda.cuda.empty_cache()
    # configs[3] x configs[2]: four dense captures, stream s on worker s mod 2, each with its own ts and table
    m = (64 << 20) - (64 << 20) % 28
    xs = [_host(make_dense10(torch_cuda, m, 140 + s)) for s in range(4)]
    wants = [oracle.decode(v, df18=True) for v in xs]
    # (a stream fed in 32 MiB pieces has launches of 8 Mi offsets: 64 tiles of them hold fewer than the 2 048 records from
    # which a batch is decided ahead by default -- the knob makes every batch go that way, as on a resident 256 Mi-sample capture)
    md = sharding.MultiDecoder(2, [0, 0], df18=True, collect_stats=True, host_threads=6, debug_gang_min=1)
    try:
        with capi.PinnedBuffers(4, m) as bufs:
            for s in range(4):
                bufs[s][:] = xs[s]
            md.decode_streams_host([bufs[s] for s in range(4)])
            for s in range(4):
                assert records(capi._frames_to_dicts(*md.stream_frames(s))) == records(wants[s][0]), s
                assert md.stream_stats(s) == wants[s][1], s
        assert md.info()["helper_threads"] == 10
        profs = [md.worker_profile(i) for i in range(2)]
        assert all(p["gang_launches"] > 0 and p["gang_batches"] > 0 for p in profs), profs   # batches decided ahead by the gang
    finally:
        md.close()


@pytest.mark.gpu_big          # opt-in (--gpu-big): 8 GiB of HBM, 215 k frames synthesised on the host, minutes of wall time
@pytest.mark.limit(1500)
def test_a_stream_just_below_the_sample_counter_limit(capi, torch_cuda):
    """The longest stream there is: 2^32 - 4 samples (3.6 minutes of signal, 8 GiB resident in HBM; the reference's sample
    counter wraps at 2^32, air.c:34, and the library refuses a stream that would reach it).  No CPU oracle at this size: the
    size-independent properties instead -- ts is the checksum of the whole greedy replay (demod.c:86,99,128,134), every decoded
    frame is one that was sent, in order -- and two independent paths through the library against each other, frame for frame
    and Try/Ok table for table: one handle decoding the stream in one pass (32 launches), and the multi-GPU driver with eight
    handles on this device (eight shards, each resolved on its own, seams and horizon stitched)."""
    from adsbdec_amd import sharding
    from tools.gen_signal import make_workload
    n = ((1 << 32) - 1) // 28 * 28
    assert (1 << 32) - n == 4
    t, truth = make_workload(torch_cuda, n, seed=21)
    d = capi.Decoder(df18=False, collect_stats=True)
    try:
        d.push_device_final(t.data_ptr(), t.numel())
        raw = d.take_raw()
        got, gstats = capi._frames_to_dicts(*raw), d.stats()
        assert len(got) > 0.9 * len(truth) and len(truth) > 200_000
        _ts_checksum(got)
        assert got[-1]["g"] > (n // 2) - 40_980 - 1200 - 30_000          # frames right up to the end-of-file horizon
        sent = iter(fr for _, fr in truth)
        assert all(any(f["frame"] == s for s in sent) for f in got)
        md = sharding.MultiDecoder(8, [0] * 8, df18=False, collect_stats=True)
        try:
            plan = md.plan(n)
            assert len(plan) == 8
            other = capi._frames_to_dicts(*md.decode_device(n, [t.data_ptr() + 2 * p["first_sample"] for p in plan]))
            assert records(other) == records(got)
            assert md.stats() == gstats
            assert md.info()["fallback"] == 0
        finally:
            md.close()
        # one sample more than the reference can count: refused, before anything is read
        d.reset()
        d.push_device(t.data_ptr(), 1 << 20)
        with pytest.raises(capi.AdsbError, match="2\\^32"):
            d.push_device(t.data_ptr(), (1 << 32) - (1 << 20))
        d.push_device_final(t.data_ptr() + 2 * (1 << 20), n - (1 << 20))   # ... and one that just fits still goes
        assert records(capi._frames_to_dicts(*d.take_raw())) == records(got)
    finally:
        d.close()
    del t
    torch_cuda.cuda.empty_cache()
