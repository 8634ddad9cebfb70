"""Multi-rank glue for one stream time-sharded over several GPUs.

One process per GPU (torch.distributed; backend "nccl" == RCCL on the GPU box,
"gloo" in CPU tests).  The data path needs NO collective: every rank scans the
offsets it owns, given a one-window halo of input samples.  The only exchange is
the gather of the sparse candidate records (tens of bytes per frame) to the rank
that runs the sequential resolver, done with gather_object over the default
process group.
"""
from __future__ import annotations

import numpy as np
import torch.distributed as dist

from . import capi


def gather_and_resolve(cands, tries, total_samples: int, dst: int = 0):
    """cands: [(g, pw, frame_bytes)] ascending; tries: uint64 ndarray ((g<<2)|code).
    Returns (frames, stats) on rank `dst`, (None, None) elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    payload = (cands, np.asarray(tries, dtype=np.uint64))
    gathered = [None] * world if rank == dst else None
    dist.gather_object(payload, gathered, dst=dst)
    if rank != dst:
        return None, None
    r = capi.Resolver()
    for c, t in gathered:  # rank order == ascending g: shards are contiguous and ordered
        r.feed(c, t)
    m_real = 2 * (total_samples // 4)
    r.advance(2 * ((total_samples + 3) // 4), max(0, m_real - capi_window() + 1))
    return r.drain(), r.stats()


def capi_window() -> int:
    return 1196  # ADSB_WINDOW
