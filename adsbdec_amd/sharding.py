"""Multi-rank glue for ONE stream time-sharded over several GPUs (SURVEY.md 8e,
BASELINE configs[4]).

One process per GPU (torch.distributed).  The data path needs NO collective: every
rank scans the preamble offsets it owns on its own halo'd slice of the stream
(adsb_plan_shards: 8 pairs before, one 1196-sample window after).  Two ways to join the
shards:

ResolvedShard (the scalable one).  Every rank resolves its own shard WHILE its kernel runs
(adsb_scan_shard_resolved: the greedy rule of demod.c:89,128,134,141 started at the shard's
first offset) and writes frames + a few head candidates straight into its region of a
shared-memory board (one node: a file in /dev/shm mapped by every rank; the records are
host-resident, tens of bytes per frame, so neither RCCL nor a gloo gather would do anything
but copy them again).  Rank 0 then runs adsb_stitch_shards -- seam repair, per-shard ts
offsets (demod.c:86,99), end-of-file horizon (air.c:94-99) -- each rank applies its ts
offset to its own frames, and the stream's frames lie on the board in shard order.
Synchronisation is three sequence numbers per step on the board itself; torch.distributed
(gloo) is only used to hand out the board's name.  When a seam cannot be decided from the
head candidates (stitcher returns -3) the step falls back to the path below.

ShardRank (the checker, and statistics runs).  adsb_scan_shard hands every CRC-valid
candidate (and try) of the shard to the host; the fixed-layout arrays are gathered on one
rank (one gloo tensor gather) where ONE resolver replays the sequential rules.

    sr = ResolvedShard(total_samples, df18=True, device=local_rank, group=gloo_group)
    x = <device tensor holding stream samples sr.first_sample .. +sr.n_samples>
    res = sr.step(x.data_ptr())         # rank 0: a ShardResult (frames in shard order); others: None

Errors: a rank whose scan fails marks its region and raises; rank 0 raises ShardError naming
that rank and tells the others through the board.  Every wait has a deadline.
"""
from __future__ import annotations

import ctypes as C
import datetime
import mmap
import os
import time

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
        self._need = (0, 0)
        self._L = capi.load()
        # Every rank sends a tensor of the SAME size (dist.gather): capacities come from the largest
        # shard of the plan, which every rank knows.  ~1 frame per 20 k samples at 1 k frames/s and
        # ~1 DF-gate pass per 150 offsets on wide-band noise; pass cand_cap / try_cap for denser streams.
        biggest = max(p["n_samples"] for p in self.plan)
        self.cand_cap = cand_cap or 65536 + biggest // 8000
        self.try_cap = try_cap or ((1 << 16) + biggest // 128 if collect_stats else 0)
        self._alloc()
        self._out, self._out_cap = (capi.Frame * 1)(), 1

    def _alloc(self):
        # one flat uint8 payload: [n_cands u64 | n_tries u64 | pad to 32 B] [cand_cap candidates] [try_cap u64]
        self._send = self._torch.zeros(CAND_BYTES * (1 + self.cand_cap) + 8 * self.try_cap, dtype=self._torch.uint8)
        base = self._send.data_ptr()
        self._hdr = C.cast(base, C.POINTER(C.c_uint64))
        self._cands = C.cast(base + CAND_BYTES, C.POINTER(capi.Candidate))
        self._tries = C.cast(base + CAND_BYTES * (1 + self.cand_cap), C.POINTER(C.c_uint64))
        self._recv = None

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
        try:
            dec = self.dec             # created here on first use: no device, wrong architecture, out of memory ...
        except Exception as e:         # ... must reach the gather as an error marker too, or the other ranks hang in it
            self._hdr[0] = _ERR
            raise ShardError(f"rank {self.rank}: cannot create the decoder: {e}") from e
        rc = self._L.adsb_scan_shard(dec._h, device_ptr, self.first_sample, self.n_samples, self.g_begin,
                                     self.g_end, self._cands, self.cand_cap, C.byref(nc),
                                     self._tries if self.try_cap else None, self.try_cap, C.byref(nt))
        self._need = (0, 0)
        if rc == -2:                   # the density guess was too low: step() regrows every rank's payload and rescans
            self._need = (nc.value, nt.value)
            self._hdr[0], self._hdr[1] = 0, 0
            return None
        if rc != 0:
            self._hdr[0] = _ERR
            raise ShardError(f"rank {self.rank}: adsb_scan_shard failed: " + (self._L.adsb_last_error(self.dec._h) or b"").decode())
        self._hdr[0], self._hdr[1] = nc.value, nt.value
        return nc.value, nt.value

    def _agree_on_capacity(self):
        """Every rank sends a payload of the SAME size (dist.gather), so a shard that needs more room makes all of
        them regrow: one all-reduce (max) of two numbers per step.  Returns True when the buffers were regrown."""
        need = self._torch.tensor([self._need[0], self._need[1]], dtype=self._torch.int64)
        if self._dist is not None and self.world > 1:
            self._dist.all_reduce(need, op=self._dist.ReduceOp.MAX, group=self.group)
        nc, nt = int(need[0]), int(need[1])
        if nc <= self.cand_cap and nt <= self.try_cap:
            return False
        self.cand_cap = max(self.cand_cap, nc + nc // 4 + 64)
        self.try_cap = max(self.try_cap, nt + nt // 4 + 64) if (self.try_cap or nt) else 0
        self._alloc()
        return True

    def step(self, device_ptr: int):
        """scan + gather + resolve.  Rank `dst` returns (Frame array, count, stats dict | None)."""
        err = None
        try:
            self.scan(device_ptr)
        except ShardError as e:
            err = e
        if self._agree_on_capacity() and err is None:   # (a failed rank still takes part in the collectives)
            try:
                self.scan(device_ptr)      # every rank: the payload buffers are new
                if self._need != (0, 0):
                    raise ShardError(f"rank {self.rank}: records exceed the regrown payload")
            except ShardError as e:
                err = e
        if err is not None:
            self._hdr[0] = _ERR
            self._gather()          # the error marker reaches rank dst, which raises too
            raise err
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


# ------------------------------------------------------------------ shards resolved where the records are
FRAME_BYTES = C.sizeof(capi.Frame)
_LINE = 64


class ShardBoard:
    """The exchange area of one time-sharded stream: per shard {adsb_shard_head | seq, done | frames | head
    candidates | call bases}, then what rank 0 hands back {status, fix_seq | adsb_shard_fix per shard | frames
    accepted by seam repairs}.  Lives in any writable buffer: a bytearray (one process driving every shard) or an
    mmap of a /dev/shm file (one process per GPU)."""

    def __init__(self, buf, world: int, frame_cap: int, head_cap: int, new_cap: int = 1024, bases_cap: int = 0):
        self.buf, self.world, self.frame_cap, self.head_cap, self.new_cap = buf, world, frame_cap, head_cap, new_cap
        self.bases_cap = bases_cap
        self.region_bytes = self.region_size(frame_cap, head_cap, bases_cap)
        need = self.size(world, frame_cap, head_cap, new_cap, bases_cap)
        assert len(buf) >= need
        self._base = C.addressof(C.c_char.from_buffer(buf))
        ctl = self._base + world * self.region_bytes
        self.ctl = np.frombuffer(buf, dtype=np.uint64, count=8, offset=world * self.region_bytes)  # [0] status, [1] fix_seq
        self.fix = (capi.ShardFix * world).from_address(ctl + _LINE)
        fix_bytes = -(-C.sizeof(capi.ShardFix) * world // _LINE) * _LINE
        self.new_frames = (capi.Frame * new_cap).from_address(ctl + _LINE + fix_bytes)

    @staticmethod
    def region_size(frame_cap, head_cap, bases_cap=0):
        return 3 * _LINE + FRAME_BYTES * frame_cap + CAND_BYTES * head_cap + -(-8 * bases_cap // _LINE) * _LINE

    @classmethod
    def size(cls, world, frame_cap, head_cap, new_cap=1024, bases_cap=0):
        fix_bytes = -(-C.sizeof(capi.ShardFix) * world // _LINE) * _LINE
        return world * cls.region_size(frame_cap, head_cap, bases_cap) + _LINE + fix_bytes + FRAME_BYTES * new_cap

    def head(self, i):
        return capi.ShardHead.from_address(self._base + i * self.region_bytes)

    def flags(self, i):  # [0] seq: the shard's result of step `seq` is in; [1] done: its ts offset has been applied
        return np.frombuffer(self.buf, dtype=np.uint64, count=8, offset=i * self.region_bytes + 2 * _LINE)

    def frames(self, i):
        return (capi.Frame * self.frame_cap).from_address(self._base + i * self.region_bytes + 3 * _LINE)

    def heads(self, i):
        return (capi.Candidate * self.head_cap).from_address(
            self._base + i * self.region_bytes + 3 * _LINE + FRAME_BYTES * self.frame_cap)

    def bases(self, i):
        return (C.c_uint64 * max(1, self.bases_cap)).from_address(
            self._base + i * self.region_bytes + 3 * _LINE + FRAME_BYTES * self.frame_cap + CAND_BYTES * self.head_cap)

    def parts(self):
        parts = (capi.ShardPart * self.world)()
        for i in range(self.world):
            parts[i].head = C.pointer(self.head(i))
            parts[i].frames = self.frames(i)
            parts[i].head_cands = self.heads(i)
            parts[i].bases = self.bases(i) if self.bases_cap else None
        return parts


def bases_capacity(plan):
    """Calls of the deqframe chain a shard can see: one per 39 780 offsets, and a few."""
    return max((p["g_end"] - p["g_begin"]) // 39780 for p in plan) + 8


class ShardResult:
    """The stream's frames as they lie on the board: per shard, the seam repair's frames, then the kept
    speculative ones (ts final).  collect() copies them into one array (reference order)."""

    def __init__(self, board: ShardBoard, serial_us: float, walk=(0, 0)):
        self.board, self.serial_us = board, serial_us
        self.calls_walked, self.calls_jumped = walk   # of the deqframe call chain: walked by the stitcher / skipped
        self.count = sum(int(f.n_new) + int(f.keep) for f in board.fix)

    def segments(self):
        b = self.board
        for i in range(b.world):
            fx = b.fix[i]
            if fx.n_new:
                yield C.addressof(b.new_frames) + int(fx.new_first) * FRAME_BYTES, int(fx.n_new)
            if fx.keep:
                yield C.addressof(b.frames(i)) + int(fx.drop_front) * FRAME_BYTES, int(fx.keep)

    def collect(self):
        out = (capi.Frame * max(1, self.count))()
        at = C.addressof(out)
        for addr, n in self.segments():
            C.memmove(at, addr, n * FRAME_BYTES)
            at += n * FRAME_BYTES
        return out, self.count


class ResolvedShard:
    """This rank's share of one time-sharded stream, resolved locally and stitched by rank `dst` (module docstring)."""

    def __init__(self, total_samples: int, df18: bool = False, device: int = -1, group=None, fix_1bit: bool = False,
                 dst: int = 0, profile: bool = False, rank: int | None = None, world: int | None = None,
                 frame_cap: int = 0, head_cap: int = 4096, board: ShardBoard | None = None, timeout_s: float = 120.0):
        if world is None:
            import torch.distributed as dist
            self._dist = dist
            rank, world = dist.get_rank(), dist.get_world_size()
        else:
            self._dist = None      # a single process driving the shards itself (tests), on a board it passes in
        self.rank, self.world, self.dst, self.group, self.timeout_s = rank, world, dst, group, timeout_s
        self.total = total_samples
        self.plan = capi.plan_shards(total_samples, world)
        me = self.plan[rank]
        self.g_begin, self.g_end = me["g_begin"], me["g_end"]
        self.first_sample, self.n_samples = me["first_sample"], me["n_samples"]
        self._dec_args = dict(df18=df18, device=device, fix_1bit=fix_1bit, profile=profile)
        self._dec = None
        self._L = capi.load()
        biggest = max(p["n_samples"] for p in self.plan)
        self.frame_cap = frame_cap or 65536 + biggest // 8000   # ~1 frame per 20 k samples at 1 k frames/s
        self.head_cap = head_cap
        self._map = self._path = None
        if board is None:
            bc = bases_capacity(self.plan)
            nbytes = ShardBoard.size(world, self.frame_cap, head_cap, bases_cap=bc)
            if self._dist is not None and world > 1:
                buf = self._shared_buffer(nbytes)
            else:
                buf = bytearray(nbytes)
            board = ShardBoard(buf, world, self.frame_cap, head_cap, bases_cap=bc)
        self.board = board
        self._step = 0
        self._fallback = None
        self.fallbacks = 0
        self.serial_us = 0.0
        self.walk = (0, 0)
        self._parts = None

    # -- setup: one file in /dev/shm, created by rank dst, mapped by everyone, unlinked once everyone has it
    def _shared_buffer(self, nbytes):
        dist = self._dist
        name = [None]
        if self.rank == self.dst:
            base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp"
            name[0] = f"{base}/adsb_shards_{os.getpid()}_{int(time.time() * 1e6) & 0xFFFFFFFF:x}"
            fd = os.open(name[0], os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            os.ftruncate(fd, nbytes)
        dist.broadcast_object_list(name, src=self.dst, group=self.group)
        if self.rank != self.dst:
            fd = os.open(name[0], os.O_RDWR)
        m = mmap.mmap(fd, nbytes)
        os.close(fd)
        dist.barrier(group=self.group)
        if self.rank == self.dst:
            os.unlink(name[0])     # the mappings keep it alive; nothing is left behind if a rank dies later
        self._map = m
        return m

    @property
    def dec(self):
        if self._dec is None:
            self._dec = capi.Decoder(**self._dec_args)
        return self._dec

    def _wait(self, cond, what):
        deadline = time.monotonic() + self.timeout_s
        spins = 0
        while not cond():
            spins += 1
            if (spins & 0x3FF) == 0 and time.monotonic() > deadline:
                raise ShardError(f"rank {self.rank}: timed out waiting for {what}")

    # -- the three phases of a step
    def scan(self, device_ptr: int):
        """adsb_scan_shard_resolved of this rank's slice, straight into its region of the board."""
        b, i = self.board, self.rank
        hd = b.head(i)
        try:
            rc = self._L.adsb_scan_shard_resolved_walk(self.dec._h, device_ptr, self.first_sample, self.n_samples, self.g_begin,
                                                       self.g_end, self.total, C.byref(hd), b.frames(i), b.frame_cap, b.heads(i),
                                                       b.head_cap, b.bases(i) if b.bases_cap else None, b.bases_cap)
            why = "" if rc == 0 else (
                f"{hd.n_frames} frames / {hd.n_head} head candidates exceed the board's capacity ({b.frame_cap} / {b.head_cap})"
                if rc == -2 else (self._L.adsb_last_error(self.dec._h) or b"").decode())
        except Exception as e:     # no device, wrong architecture, out of memory ...
            rc, why = -1, str(e)
        if rc != 0:
            hd.status = 1
            raise ShardError(f"rank {self.rank}: adsb_scan_shard_resolved failed: {why}")

    def _walk_calls(self):
        """This shard's own walk of the deqframe call chain (the stitcher jumps onto it): in parallel on every rank."""
        b, i = self.board, self.rank
        if b.bases_cap:
            self._L.adsb_shard_walk(C.byref(b.head(i)), b.frames(i), self.total, b.bases(i), b.bases_cap)

    def load_candidates(self, cands):
        """The same through the HOST resolver in chain mode, from a list [(g, pw, frame)] of this shard's candidates:
        lets everything behind the scan be exercised on machines without a GPU."""
        b, i = self.board, self.rank
        r = capi.Resolver()
        head_end = min(self.g_end, self.g_begin + 16384)
        self._L.adsb_resolver_start_chain(r._h, self.g_begin, head_end)
        r.feed(list(cands))
        r.advance(0, self.g_end)
        nf = int(self._L.adsb_resolver_drain(r._h, b.frames(i), b.frame_cap))
        nh = int(self._L.adsb_resolver_head(r._h, b.heads(i), b.head_cap))
        hd = b.head(i)
        hd.g_begin, hd.g_end, hd.n_frames, hd.n_head = self.g_begin, self.g_end, nf, min(nh, b.head_cap)
        hd.head_end, hd.skipped, hd.status = head_end, int(self._L.adsb_resolver_skipped(r._h)), 0 if nh <= b.head_cap else 1
        hd.n_bases = hd.walk_final = 0
        r.close()
        self._walk_calls()

    def stitch(self):
        """Rank dst: the serial part.  Returns the stitcher's code (0, or -3: undecidable seam)."""
        b = self.board
        t0 = time.perf_counter()
        for i in range(b.world):
            if b.head(i).status != 0:
                b.ctl[0] = 1
                raise ShardError(f"rank {i} reported a failed scan")
        if self._parts is None:
            self._parts = b.parts()
        n_new = C.c_size_t(0)
        ws = (C.c_uint64 * 2)()
        rc = self._L.adsb_stitch_shards_ex(self._parts, b.world, self.total, b.fix, b.new_frames, b.new_cap, C.byref(n_new), ws)
        self.walk = (int(ws[0]), int(ws[1]))
        self.serial_us = (time.perf_counter() - t0) * 1e6
        if rc not in (0, -3):
            b.ctl[0] = 1
            raise ShardError(f"adsb_stitch_shards failed ({rc})")
        b.ctl[0] = 0 if rc == 0 else 3
        return rc

    def apply_fix(self):
        b, i = self.board, self.rank
        fx = b.fix[i]
        if fx.keep and fx.ts_sub:
            seg = (capi.Frame * int(fx.keep)).from_address(C.addressof(b.frames(i)) + int(fx.drop_front) * FRAME_BYTES)
            self._L.adsb_shard_apply_fix(seg, int(fx.keep), int(fx.ts_sub))

    def step(self, device_ptr: int | None = None, cands=None):
        """scan -> publish -> (rank dst: stitch) -> apply own ts offset -> (rank dst: result).  One process per shard.
        cands: this shard's candidates by hand instead of a device scan (load_candidates; tests without a GPU)."""
        self._step += 1
        k, b = self._step, self.board
        err = None
        try:
            if cands is not None:
                self.load_candidates(cands)
                if b.head(self.rank).status != 0:
                    raise ShardError(f"rank {self.rank}: records exceed the board's capacity")
            else:
                self.scan(device_ptr)
        except ShardError as e:
            err = e
        b.flags(self.rank)[0] = k
        if self.rank == self.dst:
            self._wait(lambda: all(int(b.flags(i)[0]) == k for i in range(b.world)), "the other ranks' scans")
            try:
                self.stitch()
            finally:
                b.ctl[1] = k
        else:
            self._wait(lambda: int(b.ctl[1]) == k, "the stitcher")
        if err is not None:
            raise err
        status = int(b.ctl[0])
        if status == 3:
            return self._step_fallback(device_ptr, cands)
        if status != 0:
            raise ShardError(f"rank {self.rank}: the step failed on another rank")
        self.apply_fix()
        b.flags(self.rank)[1] = k
        if self.rank != self.dst:
            return None
        self._wait(lambda: all(int(b.flags(i)[1]) == k for i in range(b.world)), "the other ranks' ts offsets")
        return ShardResult(b, self.serial_us, self.walk)

    def _step_fallback(self, device_ptr, cands=None):
        """A seam the head candidates cannot decide: this step goes through every-candidate-to-one-resolver."""
        self.fallbacks += 1
        if self._fallback is None:
            self._fallback = ShardRank(self.total, group=self.group, dst=self.dst,
                                       rank=None if self._dist is not None else self.rank,
                                       world=None if self._dist is not None else self.world, **self._dec_args)
        fb = self._fallback
        if cands is not None:
            fb.load_records(list(cands))
            return fb.exchange()
        fb._dec = self.dec
        return fb.step(device_ptr)

    def close(self):
        if self._dec is not None:
            self._dec.close()
            self._dec = None
        if self._map is not None:
            try:
                self.board = None
                self._map.close()
            except BufferError:
                pass                # ctypes views still alive: the mapping goes with the process
            self._map = None


def decode_sharded(dec, device_ptr: int, total_samples: int, n_shards: int, frame_cap: int = 0, head_cap: int = 4096):
    """ONE process, one decoder handle, every shard in turn (tests; a host with a single GPU): resolved shards on a
    private board, stitched.  device_ptr addresses stream sample 0.  Returns (ShardResult | None, stitcher code)."""
    L = capi.load()
    plan = capi.plan_shards(total_samples, n_shards)
    fc = frame_cap or 65536 + max(p["n_samples"] for p in plan) // 8000
    bc = bases_capacity(plan)
    board = ShardBoard(bytearray(ShardBoard.size(n_shards, fc, head_cap, bases_cap=bc)), n_shards, fc, head_cap, bases_cap=bc)
    for i, p in enumerate(plan):
        hd = board.head(i)
        rc = L.adsb_scan_shard_resolved_walk(dec._h, device_ptr + 2 * p["first_sample"], p["first_sample"], p["n_samples"], p["g_begin"],
                                             p["g_end"], total_samples, C.byref(hd), board.frames(i), fc, board.heads(i), head_cap,
                                             board.bases(i), bc)
        if rc != 0:
            raise ShardError(f"shard {i}: adsb_scan_shard_resolved failed ({rc}): " + (L.adsb_last_error(dec._h) or b"").decode())
    parts = board.parts()
    n_new = C.c_size_t(0)
    ws = (C.c_uint64 * 2)()
    t0 = time.perf_counter()
    rc = L.adsb_stitch_shards_ex(parts, n_shards, total_samples, board.fix, board.new_frames, board.new_cap, C.byref(n_new), ws)
    us = (time.perf_counter() - t0) * 1e6
    if rc != 0:
        return None, rc
    for i in range(n_shards):
        fx = board.fix[i]
        if fx.keep and fx.ts_sub:
            seg = (capi.Frame * int(fx.keep)).from_address(C.addressof(board.frames(i)) + int(fx.drop_front) * FRAME_BYTES)
            L.adsb_shard_apply_fix(seg, int(fx.keep), int(fx.ts_sub))
    return ShardResult(board, us, (int(ws[0]), int(ws[1]))), 0
