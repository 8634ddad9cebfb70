"""world_size-2 gloo test of the multi-rank path on CPU: each rank scans
its planned shard, candidate records are gathered on rank 0 with no data-path
collective other than that gather, and one resolver reproduces the sequential
reference.  On CPU the per-shard scan is the oracle's exhaustive evaluation (a
test stand-in for the HIP kernel; the GPU version of this test is in
test_gpu_parity.py)."""
import os
import pickle
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, records


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adsbdec_amd import capi, sharding
    from oracle import gen_signal as G, oracle as O
    x, _ = G.dense_capture(1 << 19, seed=44, sigma=50.0, n_frames=150)   # same stream on every rank
    plan = capi.plan_shards(x.size, world)[rank]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import shard_power
    a, off = shard_power(O, x, plan)
    cands, tries = O.scan_all(a, plan["g_begin"] - off, plan["g_end"] - off, True)
    cands = [(g + off, pw, fr) for g, pw, fr in cands]
    tries = tries + np.uint64(off << 2)
    frames, stats = sharding.gather_and_resolve(cands, tries, x.size, dst=0)
    if rank == 0:
        with open(out_path, "wb") as f:
            pickle.dump((frames, stats), f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_gather_resolve(tmp_path, oracle, capi):
    from oracle import gen_signal as G
    out = str(tmp_path / "r0.pkl")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    frames, stats = pickle.load(open(out, "rb"))
    x, _ = G.dense_capture(1 << 19, seed=44, sigma=50.0, n_frames=150)
    want, wstats = oracle.decode(x, df18=True)
    assert records(frames) == records(want)
    assert stats == wstats
