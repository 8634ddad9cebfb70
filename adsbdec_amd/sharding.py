"""Multi-rank glue for ONE stream time-sharded over several GPUs (SURVEY.md 8e,
BASELINE configs[4]).

One process per GPU (torch.distributed).  The data path needs NO collective: every
rank scans the preamble offsets it owns on its own halo'd slice of the stream
(adsb_plan_shards: 8 pairs before, one 1196-sample window after; adsb_scan_shard is
stateless).  The only exchange is the gather of the sparse CRC-valid candidate
records -- fixed-layout `adsb_candidate` structs (32 bytes each), which the library
hands out in HOST memory -- to the rank that replays the reference's sequential
rules once (greedy skip demod.c:128,134; ts demod.c:86,99; deqframe call pattern and
end-of-file horizon air.c:94-99): one tensor gather per step over a gloo (CPU) group.
RCCL/xGMI are not used: moving ~30 bytes per frame through device memory and a GPU
collective would only add two PCIe hops.

    sr = ShardRank(total_samples, df18=True, device=local_rank, group=gloo_group)
    x = <device tensor holding stream samples sr.first_sample .. +sr.n_samples>
    frames = sr.step(x.data_ptr())      # rank 0: (Frame array, count); others: None

Errors: a rank whose scan fails sends an error marker instead of a count and raises;
rank 0 raises ShardError naming that rank.  The other ranks find out when the process
group is torn down (torchrun ends every rank when one fails; the group is created with
a timeout so that a stand-alone launcher cannot hang for ever).
"""
from __future__ import annotations

import ctypes as C
import datetime

import numpy as np

from . import capi

WINDOW = 1196                      # ADSB_WINDOW
_ERR = (1 << 64) - 1               # count slot value: "this rank's scan failed"
CAND_BYTES = C.sizeof(capi.Candidate)


class ShardError(RuntimeError):
    pass


def gloo_group(timeout_s: int = 300):
    """A CPU-side group next to the default one (nccl == RCCL on the GPU box) for the record gather."""
    import torch.distributed as dist
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    return dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))


class ShardRank:
    """This rank's share of one time-sharded stream."""

    def __init__(self, total_samples: int, df18: bool = False, device: int = -1, group=None,
                 fix_1bit: bool = False, collect_stats: bool = False, dst: int = 0, profile: bool = False,
                 rank: int | None = None, world: int | None = None, cand_cap: int = 0, try_cap: int = 0):
        import torch
        self._torch = torch
        if world is None:
            import torch.distributed as dist
            self._dist = dist
            rank, world = dist.get_rank(), dist.get_world_size()
        else:
            self._dist = None      # a single process driving the shards itself (tests)
        self.rank, self.world, self.dst, self.group = rank, world, dst, group
        self.total = total_samples
        self.collect_stats = collect_stats
        self.plan = capi.plan_shards(total_samples, world)
        me = self.plan[rank]
        self.g_begin, self.g_end = me["g_begin"], me["g_end"]
        self.first_sample, self.n_samples = me["first_sample"], me["n_samples"]
        self._dec_args = dict(df18=df18, device=device, fix_1bit=fix_1bit, collect_stats=collect_stats, profile=profile)
        self._dec = None           # created by the first scan: the gather / resolve half works without a GPU
        self._L = capi.load()
        # Every rank sends a tensor of the SAME size (dist.gather): capacities come from the largest
        # shard of the plan, which every rank knows.  ~1 frame per 20 k samples at 1 k frames/s and
        # ~1 DF-gate pass per 150 offsets on wide-band noise; pass cand_cap / try_cap for denser streams.
        biggest = max(p["n_samples"] for p in self.plan)
        self.cand_cap = cand_cap or 65536 + biggest // 8000
        self.try_cap = try_cap or ((1 << 16) + biggest // 128 if collect_stats else 0)
        # one flat uint8 payload: [n_cands u64 | n_tries u64 | pad to 32 B] [cand_cap candidates] [try_cap u64]
        self._send = torch.zeros(CAND_BYTES * (1 + self.cand_cap) + 8 * self.try_cap, dtype=torch.uint8)
        base = self._send.data_ptr()
        self._hdr = C.cast(base, C.POINTER(C.c_uint64))
        self._cands = C.cast(base + CAND_BYTES, C.POINTER(capi.Candidate))
        self._tries = C.cast(base + CAND_BYTES * (1 + self.cand_cap), C.POINTER(C.c_uint64))
        self._recv = None
        self._out, self._out_cap = (capi.Frame * 1)(), 1

    @property
    def dec(self):
        if self._dec is None:
            self._dec = capi.Decoder(**self._dec_args)
        return self._dec

    def load_records(self, cands, tries=()):
        """Put records into the send buffer by hand (what scan() does through the C-ABI): lets the
        gather / resolve half be exercised on machines without a GPU.  cands: [(g, pw, frame)]."""
        if len(cands) > self.cand_cap or len(tries) > self.try_cap:
            raise ShardError("records exceed the gather's capacity")
        for i, (g, pw, fr) in enumerate(cands):
            c = self._cands[i]
            c.g, c.pw, c.len, c.reserved = g, pw, len(fr), 0
            for k, b in enumerate(fr):
                c.frame[k] = b
        for i, t in enumerate(tries):
            self._tries[i] = int(t)
        self._hdr[0], self._hdr[1] = len(cands), len(tries)

    def scan(self, device_ptr: int):
        """adsb_scan_shard of this rank's slice into the send buffer.  Returns (n_cands, n_tries)."""
        nc, nt = C.c_size_t(0), C.c_size_t(0)
        rc = self._L.adsb_scan_shard(self.dec._h, device_ptr, self.first_sample, self.n_samples, self.g_begin,
                                     self.g_end, self._cands, self.cand_cap, C.byref(nc),
                                     self._tries if self.try_cap else None, self.try_cap, C.byref(nt))
        if rc != 0:
            self._hdr[0] = _ERR
            why = (f"{nc.value} candidates / {nt.value} tries exceed the gather's capacity "
                   f"({self.cand_cap} / {self.try_cap}): construct ShardRank with larger cand_cap / try_cap") if rc == -2 \
                else (self._L.adsb_last_error(self.dec._h) or b"").decode()
            raise ShardError(f"rank {self.rank}: adsb_scan_shard failed: {why}")
        self._hdr[0], self._hdr[1] = nc.value, nt.value
        return nc.value, nt.value

    def step(self, device_ptr: int):
        """scan + gather + resolve.  Rank `dst` returns (Frame array, count, stats dict | None)."""
        try:
            self.scan(device_ptr)
        except ShardError:
            self._gather()          # the error marker reaches rank dst, which raises too
            raise
        return self.exchange()

    def exchange(self):
        """Gather every rank's send buffer on rank `dst` and resolve there."""
        parts = self._gather()
        if self.rank != self.dst:
            return None
        return self.resolve(parts)

    def _gather(self):
        if self._dist is None or self.world == 1:
            return [self._send]
        if self.rank == self.dst and self._recv is None:
            self._recv = [self._torch.zeros_like(self._send) for _ in range(self.world)]
        self._dist.gather(self._send, self._recv if self.rank == self.dst else None, dst=self.dst, group=self.group)
        return self._recv

    def resolve(self, parts):
        """Feed the shards' records, in rank order (== ascending g: shards are contiguous and ordered),
        to ONE resolver and replay the sequential rules.  parts: flat uint8 payload tensors."""
        L = self._L
        r = L.adsb_resolver_create()
        try:
            n_all = 0
            for k, t in enumerate(parts):
                hdr = C.cast(t.data_ptr(), C.POINTER(C.c_uint64))
                if hdr[0] == _ERR:
                    raise ShardError(f"rank {k} reported a failed scan")
                nc, nt = int(hdr[0]), int(hdr[1])
                L.adsb_resolver_feed(r, C.cast(t.data_ptr() + CAND_BYTES, C.POINTER(capi.Candidate)), nc,
                                     C.cast(t.data_ptr() + CAND_BYTES * (1 + self.cand_cap), C.POINTER(C.c_uint64)) if nt else None, nt)
                n_all += nc
            m_real = 2 * (self.total // 4)
            L.adsb_resolver_advance(r, 2 * ((self.total + 3) // 4), max(0, m_real - WINDOW + 1))
            if self._out_cap < n_all:
                self._out_cap = n_all + n_all // 4 + 1
                self._out = (capi.Frame * self._out_cap)()
            got = int(L.adsb_resolver_drain(r, self._out, self._out_cap))
            stats = None
            if self.collect_stats:
                st = capi.Stats()
                L.adsb_resolver_stats(r, C.byref(st))
                stats = capi._stats_to_dict(st)
            return self._out, got, stats
        finally:
            L.adsb_resolver_destroy(r)

    def close(self):
        if self._dec is not None:
            self._dec.close()
            self._dec = None


def gather_and_resolve(cands, tries, total_samples: int, dst: int = 0):
    """Object-based variant kept for hosts that hold candidates as Python lists (CPU tests).
    cands: [(g, pw, frame_bytes)] ascending; tries: uint64 ndarray ((g<<2)|code).
    Returns (frames, stats) on rank `dst`, (None, None) elsewhere."""
    import torch.distributed as dist
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
    r.advance(2 * ((total_samples + 3) // 4), max(0, m_real - WINDOW + 1))
    return r.drain(), r.stats()
