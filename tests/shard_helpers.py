"""Test-side plumbing for the shard PRIMITIVES of the C-ABI (adsb_plan_shards, adsb_scan_shard_resolved_walk /
adsb_resolver_start_chain, adsb_stitch_shards[_stats], adsb_shard_apply_fix) with any number of shards of any size.
The product's own driver of these primitives is csrc/multi.cpp (adsb_multi_*); these helpers exist so that tests can
cut a stream where that driver never would (13 shards of a 3 Mi-sample capture, head windows of 300 offsets ...)."""
import ctypes as C

import numpy as np

TAIL = 42181           # ADSB_TAIL_OFFSETS
HEAD_TRY_REACH = 1200


class ShardSet:
    """Per-shard results as the stitcher sees them; every array is kept alive here."""

    def __init__(self, capi, total, plan):
        self.capi, self.L, self.total, self.plan = capi, capi.load(), total, plan
        n = len(plan)
        self.heads = [capi.ShardHead() for _ in range(n)]
        self.frames = [None] * n
        self.cands = [None] * n
        self.bases = [None] * n
        self.head_tries = [None] * n
        self.tail_tries = [None] * n
        self.head_tries_end = [0] * n
        self.tail_from = [(1 << 64) - 1] * n

    def tail_first_offset(self):
        m_ref = 2 * ((self.total + 3) // 4)
        return m_ref - TAIL if m_ref > TAIL else 0

    def parts(self, with_bases=True, with_stats=False):
        capi = self.capi
        parts = (capi.ShardPart * len(self.plan))()
        for i in range(len(self.plan)):
            parts[i].head = C.pointer(self.heads[i])
            parts[i].frames = self.frames[i]
            parts[i].head_cands = self.cands[i]
            parts[i].bases = self.bases[i] if (with_bases and self.bases[i] is not None and self.heads[i].n_bases) else None
            parts[i].tail_from = (1 << 64) - 1
            if with_stats:
                ht, tt = self.head_tries[i], self.tail_tries[i]
                parts[i].head_tries = ht.ctypes.data_as(C.POINTER(C.c_uint64)) if ht is not None and ht.size else None
                parts[i].n_head_tries = 0 if ht is None else ht.size
                parts[i].head_tries_end = self.head_tries_end[i]
                parts[i].tail_tries = tt.ctypes.data_as(C.POINTER(C.c_uint64)) if tt is not None and tt.size else None
                parts[i].n_tail_tries = 0 if tt is None else tt.size
                parts[i].tail_from = self.tail_from[i]
        return parts

    def stitch(self, with_bases=True, with_stats=False, new_cap=4096):
        """-> (rc, [(g, ts, pw, frame bytes)], stats dict | None, (calls walked, calls jumped), fixes)"""
        capi, L, n = self.capi, self.L, len(self.plan)
        parts = self.parts(with_bases, with_stats)
        fix = (capi.ShardFix * n)()
        new = (capi.Frame * new_cap)()
        n_new = C.c_size_t(0)
        ws = (C.c_uint64 * 2)()
        st = capi.Stats()
        if with_stats:
            rc = L.adsb_stitch_shards_stats(parts, n, self.total, fix, new, new_cap, C.byref(n_new), ws, C.byref(st))
        else:
            rc = L.adsb_stitch_shards_ex(parts, n, self.total, fix, new, new_cap, C.byref(n_new), ws)
        out = []
        if rc == 0:
            for i in range(n):
                fx = fix[i]
                for q in range(fx.n_new):
                    f = new[fx.new_first + q]
                    out.append((int(f.g), int(f.ts), int(f.pw), bytes(f.frame[: f.len])))
                if fx.keep:
                    seg = (capi.Frame * int(fx.keep))()
                    C.memmove(seg, C.addressof(self.frames[i]) + int(fx.drop_front) * C.sizeof(capi.Frame),
                              int(fx.keep) * C.sizeof(capi.Frame))
                    L.adsb_shard_apply_fix(seg, int(fx.keep), int(fx.ts_sub))
                    out += [(int(f.g), int(f.ts), int(f.pw), bytes(f.frame[: f.len])) for f in seg]
        stats = capi._stats_to_dict(st) if (with_stats and rc == 0) else None
        return rc, out, stats, (int(ws[0]), int(ws[1])), [(int(f.drop_front), int(f.n_new), int(f.keep)) for f in fix]


def from_candidates(capi, cands, total, n_shards, head_span=16384, tries=None, walk=True):
    """Every shard resolved by the HOST resolver in chain mode from the exhaustive candidate list (and, for statistics,
    the exhaustive list of tries: uint64 (g << 2) | code, ascending): what the device path produces, without a device."""
    L = capi.load()
    plan = capi.plan_shards(total, n_shards)
    ss = ShardSet(capi, total, plan)
    t0 = ss.tail_first_offset()
    k = 0
    tr = None if tries is None else np.ascontiguousarray(tries, dtype=np.uint64)
    for i, sh in enumerate(plan):
        r = capi.Resolver()
        head_end = min(sh["g_end"], sh["g_begin"] + head_span)
        assert L.adsb_resolver_start_chain(r._h, sh["g_begin"], head_end) == 0
        cap = (sh["g_end"] - sh["g_begin"]) // 39780 + 8
        ss.bases[i] = (C.c_uint64 * cap)()
        if walk:
            assert L.adsb_resolver_start_walk(r._h, sh["g_begin"], sh["g_end"], total, ss.bases[i], cap) == 0
        mine = []
        while k < len(cands) and cands[k][0] < sh["g_end"]:
            mine.append(cands[k])
            k += 1
        mt = None
        if tr is not None:
            mt = tr[((tr >> np.uint64(2)) >= sh["g_begin"]) & ((tr >> np.uint64(2)) < sh["g_end"])]
        r.feed(mine, mt)
        r.advance(0, sh["g_end"])
        ss.frames[i] = (capi.Frame * max(1, len(mine)))()
        nf = int(L.adsb_resolver_drain(r._h, ss.frames[i], len(ss.frames[i])))
        ss.cands[i] = (capi.Candidate * max(1, len(mine)))()
        nh = int(L.adsb_resolver_head(r._h, ss.cands[i], len(ss.cands[i])))
        hd = ss.heads[i]
        hd.g_begin, hd.g_end, hd.n_frames, hd.n_head, hd.head_end = sh["g_begin"], sh["g_end"], nf, nh, head_end
        hd.skipped, hd.status = int(L.adsb_resolver_skipped(r._h)), 0
        if walk:
            fin = C.c_int(0)
            nb = int(L.adsb_resolver_walk_result(r._h, C.byref(fin)))
            hd.n_bases, hd.walk_final = (nb if nb <= cap else 0), fin.value
        ok = r.stats()["ok"]
        hd.ok[0], hd.ok[1], hd.ok[2] = ok[11], ok[17], ok[18]
        if tr is not None:
            st = r.stats()["try"]
            hd.has_tries = 1
            hd.tries[0], hd.tries[1], hd.tries[2] = st[11], st[17], st[18]
            g = mt >> np.uint64(2)
            ss.head_tries_end[i] = min(sh["g_end"], head_end + HEAD_TRY_REACH)
            ss.head_tries[i] = np.ascontiguousarray(mt[g < ss.head_tries_end[i]])
            if sh["g_end"] > t0:
                ss.tail_from[i] = max(sh["g_begin"], t0 - t0 % 28)
                ss.tail_tries[i] = np.ascontiguousarray(mt[g >= ss.tail_from[i]])
        r.close()
    return ss


def from_device(capi, dec, device_ptr, total, n_shards, stats=False, frame_cap=0, head_cap=4096):
    """Every shard scanned and resolved on the device through ONE handle, in turn (adsb_scan_shard_resolved_walk; with
    `stats` also the two windows of tries through adsb_scan_shard).  device_ptr addresses stream sample 0."""
    L = capi.load()
    plan = capi.plan_shards(total, n_shards)
    ss = ShardSet(capi, total, plan)
    t0 = ss.tail_first_offset()
    fc = frame_cap or 65536 + max(p["n_samples"] for p in plan) // 8000
    head_span = 16384
    for i, p in enumerate(plan):
        ss.frames[i] = (capi.Frame * fc)()
        ss.cands[i] = (capi.Candidate * head_cap)()
        cap = (p["g_end"] - p["g_begin"]) // 39780 + 8
        ss.bases[i] = (C.c_uint64 * cap)()
        if p["g_end"] <= p["g_begin"]:
            ss.heads[i].g_begin, ss.heads[i].g_end, ss.heads[i].has_tries = p["g_begin"], p["g_end"], int(stats)
            continue
        rc = L.adsb_scan_shard_resolved_walk(dec._h, device_ptr + 2 * p["first_sample"], p["first_sample"], p["n_samples"],
                                             p["g_begin"], p["g_end"], total, C.byref(ss.heads[i]), ss.frames[i], fc, ss.cands[i],
                                             head_cap, ss.bases[i], cap)
        if rc != 0:
            raise capi.AdsbError(f"shard {i}: adsb_scan_shard_resolved_walk failed ({rc}): " + (L.adsb_last_error(dec._h) or b"").decode())
        if stats:
            assert ss.heads[i].has_tries == 1
            he = min(p["g_end"], int(ss.heads[i].head_end) + HEAD_TRY_REACH)
            ss.head_tries_end[i] = he
            s0 = max(0, 2 * (p["g_begin"] - 8))
            s1 = min(total, 2 * (he - 1 + 1196))
            _, _, ss.head_tries[i] = dec.scan_shard(device_ptr + 2 * s0, s0, s1 - s0, p["g_begin"], he)
            if p["g_end"] > t0:
                tf = max(p["g_begin"], t0 - t0 % 28)
                ss.tail_from[i] = tf
                s0 = max(0, 2 * (tf - 8))
                s1 = min(total, 2 * (p["g_end"] - 1 + 1196))
                _, _, ss.tail_tries[i] = dec.scan_shard(device_ptr + 2 * s0, s0, s1 - s0, tf, p["g_end"])
    return ss
