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
    # the tensor path of the GPU bench (ShardRank: fixed-layout adsb_candidate arrays, one gather per
    # step), with the oracle's records standing in for adsb_scan_shard
    sr = sharding.ShardRank(x.size, df18=True, group=sharding.gloo_group(), collect_stats=True)
    assert (sr.g_begin, sr.g_end) == (plan["g_begin"], plan["g_end"])
    tensor_out = None
    for _ in range(2):                      # buffers are reused from step to step
        sr.load_records(cands, tries)
        got = sr.exchange()
        if rank == 0:
            arr, n, st = got
            tensor_out = (capi._frames_to_dicts(arr, n), st)
        else:
            assert got is None
    # the scalable path (ResolvedShard): every rank resolves its own shard, writes frames + head candidates into its
    # region of a shared-memory board, rank 0 stitches (seams, ts offsets, horizon), every rank fixes its own ts
    rs = sharding.ResolvedShard(x.size, df18=True, group=sharding.gloo_group(), timeout_s=60)
    assert (rs.g_begin, rs.g_end) == (plan["g_begin"], plan["g_end"])
    resolved_out = None
    for _ in range(3):                      # the board is reused from step to step
        res = rs.step(cands=cands)
        if rank == 0:
            arr, n = res.collect()
            resolved_out = (capi._frames_to_dicts(arr, n), res.serial_us, rs.fallbacks)
        else:
            assert res is None
    # a rank whose scan fails: everybody raises, rank 0 names it
    try:
        rs.step(cands=cands if rank == 0 else [(rs.g_begin, 1, b"\x8d" * 14)] * (rs.board.frame_cap + 1))
        raised2 = None
    except sharding.ShardError as e:
        raised2 = str(e)
    assert raised2 is not None and ("rank 1" in raised2 or rank == 1), raised2
    rs.close()
    # error propagation: rank 1 reports a failed scan, rank 0 must raise and name it
    if rank == 1:
        sr._hdr[0] = sharding._ERR
    try:
        sr.exchange()
        raised = None
    except sharding.ShardError as e:
        raised = str(e)
    assert (raised is not None and "rank 1" in raised) if rank == 0 else raised is None
    if rank == 0:
        with open(out_path, "wb") as f:
            pickle.dump((frames, stats, tensor_out, resolved_out), f)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_gather_resolve(tmp_path, oracle, capi):
    from oracle import gen_signal as G
    out = str(tmp_path / "r0.pkl")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    frames, stats, tensor_out, resolved_out = pickle.load(open(out, "rb"))
    x, _ = G.dense_capture(1 << 19, seed=44, sigma=50.0, n_frames=150)
    want, wstats = oracle.decode(x, df18=True)
    assert records(frames) == records(want)
    assert stats == wstats
    assert records(tensor_out[0]) == records(want)
    assert tensor_out[1] == wstats
    assert records(resolved_out[0]) == records(want) and resolved_out[2] == 0
    assert resolved_out[1] < 5000          # the stitcher's serial part, microseconds (two shards, 512 Ki samples)
