"""world_size-2 gloo tests, on CPU, of the multi-PROCESS shape bench.py runs at N > 1 (stream mode, BASELINE configs[3]):
one process and one independent stream per GPU, no data-path collective -- the ranks only meet at the
barriers around the timed region, for the max of their times and for the AND of their parity flags (bench.Ranks).

On CPU each rank's "decode" is the host resolver over the oracle's exhaustive candidate list of ITS stream (a test
stand-in for the HIP kernel; the GPU version is tests/test_gpu_parity.py::test_two_rank_independent_streams_...).
Also here: a rank that dies mid-job -- the survivors must give up non-zero, quickly, and say where.
(The time-sharded mode, configs[4], is ONE process with a worker thread per device -- csrc/multi.cpp -- and has no ranks.)"""
import os
import pickle
import socket
import sys
import time

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, records


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream_of(rank):
    from tools import gen_signal as G
    return G.dense_capture((1 << 19) + 4 * rank, seed=44 + rank, sigma=50.0, n_frames=150 + 20 * rank)[0]


def _worker(rank, world, port, out_dir, die_at):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import bench
    from adsbdec_amd import capi
    from oracle import oracle as O
    t_start = time.time()
    try:
        ranks = bench.Ranks(world, rank, timeout_s=20.0)
        x = _stream_of(rank)                       # every rank has a stream of its own
        a = O.power(x)
        g_end = a.size - 1195
        cands, tries = O.scan_all(a, 0, g_end, True)
        ranks.fence(what="opening barrier")
        if die_at == "step" and rank == 1:
            os._exit(9)                            # SIGKILL's moral equivalent: no goodbye to the process group
        t0 = time.perf_counter()
        r = capi.Resolver()
        r.feed(cands, tries)
        r.advance(2 * ((x.size + 3) // 4), g_end)
        frames, stats = r.drain(), r.stats()
        dt = time.perf_counter() - t0 + (0.25 if rank == 1 else 0.0)   # rank 1 is "slower": the job's time is the max
        ranks.fence(what="closing barrier")
        dt_max = ranks.max_over(dt)
        want, wstats = O.decode(x, df18=True)
        ok = records(frames) == records(want) and stats == wstats
        all_ok = ranks.all_ok(ok)
        # a mismatch on ONE rank fails the job on every rank
        all_ok_2 = ranks.all_ok(rank != 1)
        with open(os.path.join(out_dir, f"r{rank}.pkl"), "wb") as f:
            pickle.dump(dict(n=len(frames), ok=ok, all_ok=all_ok, all_ok_2=all_ok_2, dt=dt, dt_max=dt_max), f)
        ranks.close()
    except SystemExit as e:
        with open(os.path.join(out_dir, f"r{rank}.exit"), "w") as f:
            f.write(f"{time.time() - t_start:.1f}\n{e}")
        os._exit(3)


def _spawn(world, out_dir, die_at=None):
    ctx = mp.get_context("spawn")
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, out_dir, die_at)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
    return [p.exitcode for p in ps]


def test_two_ranks_two_streams_gated_on_every_rank(tmp_path, oracle, capi):
    assert _spawn(2, str(tmp_path)) == [0, 0]
    res = [pickle.load(open(tmp_path / f"r{r}.pkl", "rb")) for r in range(2)]
    for r in res:
        assert r["ok"] and r["all_ok"] and not r["all_ok_2"]
        assert r["dt_max"] == pytest.approx(res[1]["dt"]) and r["dt_max"] >= res[0]["dt"] + 0.2
    assert res[0]["n"] != res[1]["n"] and min(res[0]["n"], res[1]["n"]) > 50     # two different streams


def test_a_dead_rank_takes_the_job_down_quickly(tmp_path, oracle, capi):
    """Rank 1 disappears between the opening barrier and the closing one (os._exit: no clean-up, like an OOM kill).  Rank 0
    must not hang in its next collective: it exits non-zero within the group's deadline (20 s here) and names the step."""
    codes = _spawn(2, str(tmp_path), die_at="step")
    assert codes[1] == 9 and codes[0] == 3, codes
    took, why = open(tmp_path / "r0.exit").read().split("\n", 1)
    assert float(took) < 60
    assert "rank 0: lost the other ranks at `closing barrier`" in why and "died or hung" in why
