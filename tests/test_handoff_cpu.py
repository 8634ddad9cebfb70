"""The host's reading of the device -> host hand-off stream (decoder.hip HandCursor), on CPU.

adsb_handoff_walk runs the streaming collect's own marker / checksum / frontier code over an image of a stream in
ordinary memory.  The images are built here from the documented format (include/adsbdec_amd.h, scan_kernel.h) with an
independent restatement of the check words, so the test pins the format as well as the walk."""
import ctypes as C

import numpy as np
import pytest

M32 = 0xFFFFFFFF
OVER, NOFIT, TRIES, LINES_SHIFT = 0x10000, 0x20000, 0x40000, 19


def rotl(x, k):
    x &= M32
    return ((x << k) | (x >> (32 - k))) & M32


def check_words(tile, nf, gen, recs):
    """scan_kernel_format.h marker_check: the XOR of the record granules and the rank-weighted sum over {g_rel, pw} of the
    records, mixed with the launch's gen, the tile and nf."""
    a = [0, 0, 0, 0]
    grans = np.asarray(recs, np.uint32).reshape(-1, 4)
    for g in grans:
        for k in range(4):
            a[k] ^= int(g[k])
    s = 0
    for r, g in enumerate(grans[0::2]):
        s = (s + (2 * r + 1) * (int(g[0]) ^ rotl(int(g[1]), 13))) & M32
    lo = a[0] ^ rotl(a[2], 16) ^ gen ^ tile ^ rotl(nf, 11) ^ s
    hi = a[1] ^ rotl(a[3], 16) ^ (~gen & M32) ^ rotl(tile, 7) ^ nf ^ rotl(s, 16)
    return lo & M32, hi & M32


def stream_granules(n):
    return (1 + 2 * n + 3) & ~3


class Image:
    """A hand-off stream under construction: tiles are appended in COMPLETION order."""

    def __init__(self, granules, gen, fill=0xDEADBEEF):
        self.w = np.full(granules * 4, fill, np.uint32)
        self.gen, self.pos, self.where = gen, 0, {}

    def tile(self, tile, n, flags=0, lines=0, rng=None, gen=None, write_records=True):
        rng = rng or np.random.default_rng(tile * 7 + n)
        recs = rng.integers(0, 1 << 32, size=(2 * n, 4), dtype=np.uint64).astype(np.uint32)
        if n:
            recs[1::2, 2:] = 0                      # {w2, w3, 0, 0}
        nf = n | flags | (lines << LINES_SHIFT)
        lo, hi = check_words(tile, nf, self.gen if gen is None else gen, recs if not (flags & NOFIT) else [])
        p = self.pos
        self.w[4 * p:4 * p + 4] = (tile, nf, lo, hi)
        if write_records and not (flags & NOFIT):
            self.w[4 * (p + 1):4 * (p + 1 + 2 * n)] = recs.reshape(-1)
        self.where[tile] = (p + 1, n)
        self.pos += max(lines * 4, stream_granules(n))
        return recs


def walk(capi, img, n_tiles, gen=None, granules=None):
    L = capi.load()
    ts = (C.c_uint32 * max(1, n_tiles))()
    tc = (C.c_uint32 * max(1, n_tiles))()
    st = C.c_int(99)
    buf = np.ascontiguousarray(img.w)
    f = L.adsb_handoff_walk(buf.ctypes.data, granules if granules is not None else buf.size // 4, n_tiles,
                            img.gen if gen is None else gen, ts, tc, C.byref(st))
    return f, st.value, list(ts)[:n_tiles], list(tc)[:n_tiles]


def test_tiles_in_completion_order_with_empty_and_full_tiles(capi):
    img = Image(4096, gen=0x1234567)
    order = [2, 0, 1, 5, 3, 4, 6]
    counts = {0: 3, 1: 0, 2: 7, 3: 1, 4: 0, 5: 12, 6: 2}
    for t in order:
        img.tile(t, counts[t])
    f, st, ts, tc = walk(capi, img, 7)
    assert (f, st) == (7, 0)
    assert tc == [counts[t] for t in range(7)]
    assert ts == [img.where[t][0] for t in range(7)]
    # tile ranges are whole 64-byte lines: marker + 2n granules, rounded up to 4
    assert img.where[0][0] - 1 == stream_granules(7) and stream_granules(0) == 4 and stream_granules(2) == 8


def test_frontier_stops_at_the_first_tile_that_is_not_in(capi):
    img = Image(1024, gen=77)
    for t in (0, 1, 3, 4):                       # tile 2 has not completed yet
        img.tile(t, 2)
    f, st, ts, tc = walk(capi, img, 5)
    assert (f, st) == (2, 2)                     # walk ends at bytes that are no marker; tiles 0, 1 are complete in order
    assert tc[:2] == [2, 2] and tc[2] == M32 and tc[3:] == [2, 2]


def test_a_record_granule_that_has_not_landed_keeps_the_tile_out(capi):
    img = Image(1024, gen=5)
    img.tile(0, 4)
    p = img.pos
    img.tile(1, 6)
    good = img.w.copy()
    img.w[4 * (p + 3) + 1] ^= 0x00010000         # one bit of one record granule of tile 1 still stale
    f, st, _, tc = walk(capi, img, 2)
    assert (f, st) == (1, 2) and tc[1] == M32
    img.w = good
    assert walk(capi, img, 2)[:2] == (2, 0)


def test_markers_of_another_launch_never_validate(capi):
    img = Image(1024, gen=1000)
    img.tile(0, 3)
    img.tile(1, 3, gen=999)                      # left over from the previous launch at the same place
    f, st, _, tc = walk(capi, img, 2)
    assert (f, st) == (1, 2) and tc[1] == M32
    assert walk(capi, img, 2, gen=999)[:2] == (0, 2)   # and the whole image read with the other launch's tag: tile 0 fails


@pytest.mark.parametrize("flag", [OVER, NOFIT])
def test_tiles_that_ask_to_be_finished_after_completion(capi, flag):
    img = Image(1024, gen=31)
    img.tile(0, 2)
    img.tile(1, 5, flags=flag)
    img.tile(2, 1)
    f, st, _, tc = walk(capi, img, 3)
    assert (f, st) == (1, 1)                     # tiles from that one on wait for the launch's end ...
    if flag == OVER:
        assert tc == [2, 5, 1]                   # ... but a tile with loose records, and the tiles behind it, are read meanwhile
    else:
        assert tc[1] == M32 and tc[2] == M32     # a range past the array ends the stream: nothing behind it can be in it


def test_a_tile_whose_tries_went_through_the_launch_wide_list_holds_nothing_up(capi):
    """kMarkTries (statistics runs, a tile that overflowed its survivor queue): the tile's records are all in the stream; only
    the count pass has to wait for the launch's counters.  Every tile may be handed on, status 0; the flag is covered by the
    marker's check words like everything else in nf."""
    img = Image(1024, gen=77)
    img.tile(0, 2)
    img.tile(1, 5, flags=TRIES)
    img.tile(2, 1, flags=TRIES, lines=3)
    f, st, ts, tc = walk(capi, img, 3)
    assert (f, st) == (3, 0) and tc == [2, 5, 1]
    assert [ts[t] for t in range(3)] == [img.where[t][0] for t in range(3)]
    img.w[4 * (img.where[1][0] - 1) + 1] ^= TRIES            # the flag flipped on its way: not the marker that was written
    f, st, _, tc = walk(capi, img, 3)
    assert st == 2 and f == 1


def test_a_tile_that_reserved_more_lines_than_it_kept_records_for(capi):
    img = Image(1024, gen=8)
    img.tile(0, 2, lines=5)                      # reserved 5 lines (20 granules) for its staged list, kept 2 records
    img.tile(1, 3)
    f, st, ts, tc = walk(capi, img, 2)
    assert (f, st) == (2, 0) and tc == [2, 3]
    assert ts[1] == 20 + 1


def test_a_tile_twice_is_corruption_and_foreign_tile_numbers_are_not_markers(capi):
    img = Image(1024, gen=3)
    img.tile(0, 1)
    img.tile(0, 1, rng=np.random.default_rng(99))
    assert walk(capi, img, 3)[:2] == (1, -1)
    img = Image(1024, gen=3)
    img.tile(0, 1)
    img.tile(9, 1)                               # a valid-looking marker of a tile this launch does not have
    assert walk(capi, img, 3)[:2] == (1, 2)


def test_a_full_stream_and_a_range_that_runs_past_it(capi):
    img = Image(16, gen=4)
    img.tile(0, 1)                               # 4 granules
    img.tile(1, 5)                               # 12 granules: the stream is exactly full
    assert walk(capi, img, 3)[:2] == (2, 1)      # tile 2 can only be elsewhere
    img = Image(64, gen=4)
    img.tile(0, 1)
    img.tile(1, 5)
    assert walk(capi, img, 2, granules=12)[:2] == (1, 2)   # seen through a 12-granule window the second range does not fit


def test_many_random_streams(capi):
    rng = np.random.default_rng(2024)
    for _ in range(200):
        n_tiles = int(rng.integers(1, 40))
        counts = rng.integers(0, 9, n_tiles)
        img = Image(4 * int(sum(stream_granules(int(c)) + 8 for c in counts)) // 4 + 64, gen=int(rng.integers(1, 1 << 32)))
        order = rng.permutation(n_tiles)
        upto = int(rng.integers(0, n_tiles + 1))           # only the first `upto` completions have happened
        for t in order[:upto]:
            img.tile(int(t), int(counts[t]), lines=int(rng.integers(0, 4)) if rng.random() < 0.3 else 0, rng=rng)
        done = set(int(t) for t in order[:upto])
        want_front = next((t for t in range(n_tiles) if t not in done), n_tiles)
        f, st, ts, tc = walk(capi, img, n_tiles)
        assert f == want_front and st == (0 if upto == n_tiles else 2)
        for t in range(n_tiles):
            if t in done:
                assert (ts[t], tc[t]) == img.where[t]
            else:
                assert tc[t] == M32


def test_stale_records_whose_differences_cancel_in_the_xor_do_not_validate(capi):
    """The marker carries two summaries of the records: their XOR, and a rank-weighted sum over {g_rel, pw}.  The case the
    second one exists for: the bytes behind a (new) marker are still the previous launch's records of the SAME frames at
    other offsets -- same frame words, g_rel differing alike in two records.  In the XOR the two differences cancel; the tile
    must stay out all the same (a round-4 harness with regular records found the XOR alone accepting such ranges)."""
    gen = 0x1234ABCD
    for d in (28, 0x100, 0x55AA, 1 << 29):
        img = Image(64, gen)
        recs = img.tile(0, 4)
        assert walk(capi, img, 1)[:2] == (1, 0)
        p = img.where[0][0]
        for r in (1, 3):                       # records 1 and 3: g_rel ^= d  ->  the XOR over the range is unchanged
            img.w[4 * (p + 2 * r)] ^= d
        f, st, _, tc = walk(capi, img, 1)
        assert (f, st) == (0, 2), (d, f, st)
        for r in (1, 3):                       # ... and the same for pw
            img.w[4 * (p + 2 * r)] ^= d
            img.w[4 * (p + 2 * r) + 1] ^= d
        assert walk(capi, img, 1)[:2] == (0, 2), d
        for r in (1, 3):
            img.w[4 * (p + 2 * r) + 1] ^= d
        assert walk(capi, img, 1)[:2] == (1, 0)


# ---- records that stand for a run of copies (scan_kernel_format.h kRecCopiesShift) ----------------------------------------
def _frame_words(frame: bytes):
    """w0..w3 of a record: the 14 bytes, the length in bits 16..23 of w3."""
    b = frame + bytes(14 - len(frame))
    w = [int.from_bytes(b[4 * k:4 * k + 4], "little") for k in range(3)] + [int.from_bytes(b[12:14], "little")]
    w[3] |= len(frame) << 16
    return w


def _stream_of(tiles, gen, collapse, tile_offsets=48_160):
    """tiles: per tile a list of candidates (g_rel, pw, frame) in ascending g_rel.  collapse: runs of the same frame at
    consecutive offsets become one record of up to three (what the kernel writes since round 5); else one record each."""
    img = Image(16 + sum(4 + 2 * len(t) for t in tiles) * 2, gen)
    for ti, cands in enumerate(tiles):
        recs, i = [], 0
        while i < len(cands):
            g, pw, fr = cands[i]
            run = 1
            if collapse:
                while run < 3 and i + run < len(cands) and cands[i + run][0] == g + run and cands[i + run][2] == fr:
                    run += 1
            w = _frame_words(fr)
            pws = [cands[i + k][1] for k in range(run)] + [0, 0]
            recs.append([g, pw, w[0], w[1]])
            recs.append([w[2], w[3] | ((run - 1) << 25), pws[1] if run > 1 else 0, pws[2] if run > 2 else 0])
            i += run
        n = len(recs) // 2
        nf = n | (((len(cands) * 2 + 1 + 3) // 4) << LINES_SHIFT)          # the tile reserved for its candidates, kept fewer records
        lo, hi = check_words(ti, nf, gen, recs)
        p = img.pos
        img.w[4 * p:4 * p + 4] = (ti, nf, lo, hi)
        if n:
            img.w[4 * (p + 1):4 * (p + 1 + 2 * n)] = np.asarray(recs, np.uint32).reshape(-1)
        img.pos += max(((len(cands) * 2 + 1 + 3) // 4) * 4, stream_granules(n))
        img.where[ti] = (p + 1, n)
    return img


@pytest.mark.parametrize("seed", range(6))
def test_records_that_stand_for_copies_resolve_like_their_expansion(capi, seed):
    """A full channel: 112-bit frames back to back, every one decoded at 1-4 neighbouring offsets (which copy the greedy scan
    lands on depends on where the previous frame ended: demod.c:125-141), some short frames and gaps in between.  The stream
    with one record per run of copies must resolve to the frames -- g, ts, pw, bytes -- of the stream with one record per
    candidate, and of the plain candidate list through adsb_resolver_feed; head candidates of a chain likewise."""
    rng = np.random.default_rng(100 + seed)
    tile_offsets, n_tiles = 48_160, 5
    cands, g = [], int(rng.integers(0, 300))
    while g < n_tiles * tile_offsets - 1300:
        long = rng.random() < 0.85
        fr = bytes(rng.integers(0, 256, 14 if long else 7, dtype=np.uint8).tolist())
        ncopy = int(rng.integers(1, 5))
        first = g - int(rng.integers(0, 2))
        for k in range(ncopy):
            if rng.random() < 0.9 and (not cands or first + k > cands[-1][0]):
                cands.append((first + k, int(rng.integers(1, 1 << 20)), fr))
        g += (1200 if long else 640) + (0 if rng.random() < 0.8 else int(rng.integers(1, 3000)))
    tiles = [[(c[0] - t * tile_offsets, c[1], c[2]) for c in cands if t * tile_offsets <= c[0] < (t + 1) * tile_offsets]
             for t in range(n_tiles)]
    # (tile-relative g_rel here: the records carry launch-relative offsets, so give every tile its base back)
    tiles = [[(g + t * tile_offsets, pw, fr) for g, pw, fr in tl] for t, tl in enumerate(tiles)]
    L = capi.load()
    g_base, total = 1_000_000 * 28, 4 * (n_tiles * tile_offsets + 1_000_000 * 28)
    m = 2 * (total // 4)
    results, heads = [], []
    for mode in ("list", "expanded", "collapsed"):
        for chain in (False, True):
            r = capi.Resolver()
            if chain:
                assert L.adsb_resolver_start_chain(r._h, g_base, g_base + 16_384) == 0
            if mode == "list":
                r.feed([(g_base + g, pw, fr) for g, pw, fr in cands])
                r.advance(m, g_base + n_tiles * tile_offsets)
            else:
                img = _stream_of(tiles, 0xABCD + seed, mode == "collapsed")
                buf = np.ascontiguousarray(img.w)
                assert L.adsb_resolver_advance_stream(r._h, buf.ctypes.data, buf.size // 4, n_tiles, img.gen, g_base, m,
                                                      g_base + n_tiles * tile_offsets, 1) == n_tiles
            got = [(f["g"], f["ts"], f["pw"], bytes(f["frame"])) for f in r.drain()]
            results.append((mode, chain, got))
            if chain:
                hb = (capi.Candidate * 4096)()
                nh = L.adsb_resolver_head(r._h, hb, 4096)
                heads.append([(hb[i].g, hb[i].pw, bytes(hb[i].frame[:hb[i].len])) for i in range(nh)])
            r.close()
    n_exp = sum(n for _, n in _stream_of(tiles, 1, False).where.values())
    n_col = sum(n for _, n in _stream_of(tiles, 1, True).where.values())
    assert n_exp == len(cands) and n_col < 0.6 * n_exp          # most frames have two or more copies
    ref = {False: results[0][2], True: results[1][2]}
    assert len(ref[False]) > 100
    for mode, chain, got in results:
        assert got == ref[chain], (mode, chain)
    assert heads[0] == heads[1] == heads[2] and len(heads[0]) > 5
    # the greedy rule really does land on later copies here: not every accepted frame is a run's first offset
    firsts = {c[0] for i, c in enumerate(cands) if i == 0 or not (cands[i - 1][0] == c[0] - 1 and cands[i - 1][2] == c[2])}
    assert any(g - g_base not in firsts for g, *_ in ref[False])
    # more hands (cfg.host_threads >= 3: gang.hpp): the caller decides, three threads write the frames -- the same frames, in the
    # same order, with the same Ok row, in stream and in chain mode; twice through one handle (the decision array starts over)
    img = _stream_of(tiles, 0xABCD + seed, True)
    buf = np.ascontiguousarray(img.w)
    for chain in (False, True):
        alone, gang = capi.Resolver(), capi.Resolver()
        assert L.adsb_resolver_set_threads(gang._h, 3, 1) == 3
        for r in (alone, gang):
            for rep in range(2):
                if chain:
                    assert L.adsb_resolver_start_chain(r._h, g_base, g_base + 16_384) == 0
                elif rep:
                    break
                assert L.adsb_resolver_advance_stream(r._h, buf.ctypes.data, buf.size // 4, n_tiles, img.gen, g_base, m,
                                                      g_base + n_tiles * tile_offsets, 0) == n_tiles
                if rep == 0 and chain:
                    assert r.stats()["ok"] == alone.stats()["ok"]
                    r.drain()
        assert gang.stats() == alone.stats() and sum(alone.stats()["ok"].values()) > 100
        got = [(f["g"], f["ts"], f["pw"], bytes(f["frame"])) for f in gang.drain()]
        assert got == [(f["g"], f["ts"], f["pw"], bytes(f["frame"])) for f in alone.drain()] == ref[chain]
        assert L.adsb_resolver_set_threads(gang._h, 0, 0) == 0
        alone.close(), gang.close()
