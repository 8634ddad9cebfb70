// stitch.hpp -- joins the per-shard results of ONE time-sharded stream (BASELINE configs[4]).
//
// Every shard has been resolved on its own rank, speculatively: the greedy chain (demod.c:89,128,134,141) started
// at the shard's first offset as if no frame of the previous shard reached into it (Resolver::start_chain).  What is
// left is sequential, but small:
//   1. seams.   If the last accepted frame before shard i ends at e > g_begin(i), the offsets [g_begin(i), e) are
//      jumped over (demod.c:128,134): the chain is re-run from e over the shard's HEAD candidates (every CRC-valid
//      candidate with g < head_end) until it accepts a candidate the speculative chain accepted too -- from there on
//      the two are the same chain.  A real frame decodes at a few neighbouring offsets and the chain re-synchronises
//      at the next frame it meets; O(1) per seam.
//   2. ts.      demod.c:86,99: ts = g + 1 - (offsets jumped so far).  A shard's frames carry that count from the
//      shard's start; the stitcher returns, per shard, what to subtract: the offsets jumped before the shard, corrected
//      for the frames the repair dropped and added.  Each rank applies it to its own frames (adsb_shard_apply_fix).
//   3. horizon. air.c:94-99: deqframe only runs when 40980 power samples are buffered, so the stream's last ~41 k
//      power samples are never scanned.  Where the last call ends depends on every call before it
//      (a frame that straddles a call's limit moves the next call's base), so the call chain is walked once over the
//      accepted frames' positions: one compare per frame and a handful of operations per call -- the only part whose
//      cost grows with the stream (~27 k calls and ~107 k frames for 2 Gi samples).
//   4. statistics (optional).  valid.c:46,68 count a Try for every VISITED offset that passes the DF gate.  Every shard has
//      counted its own against its speculative chain, on the device; that count is right except where the true chain
//      differs: between a shard's first offset and the point where the repaired chain is the speculative one again, and
//      beyond the horizon.  For those two windows the shards hand over the DF-gate passes themselves (a few hundred words)
//      and the difference is counted here.
// Host-only code: no HIP in here, tested on CPU (tests/test_host_logic.py).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/adsbdec_amd_diag.h"

namespace adsb {

inline uint64_t frame_span(const adsb_frame &f) { return 80 + 80 * (uint64_t)f.len; } // demod.c:109,120,123
inline uint64_t cand_span(const adsb_candidate &c) { return 80 + 80 * (uint64_t)c.len; }

// One shard's own walk of the deqframe call chain (air.c:94-99, demod.c:89) over its speculative frames, from the guessed
// entry base g_begin: bases[0] = g_begin, then the base of every later call, the last one being the base whose call is not
// the shard's any more (its limit lies beyond g_end) or does not fire at all (*final = 1: the stream ends first).  Returns
// how many bases there are (more than cap: the caller's array was too small and holds the first cap).
inline size_t walk_shard_calls(const adsb_frame *F, uint64_t nF, uint64_t g_begin, uint64_t g_end, uint64_t total_samples,
                               uint64_t *bases, size_t cap, int *final)
{
    const uint64_t m_ref = 2 * ((total_samples + 3) / 4);
    uint64_t base = g_begin, k = 0, last_end = 0;
    size_t n = 0;
    *final = 0;
    for (;;) {
        if (n < cap)
            bases[n] = base;
        n++;
        const uint64_t fire = base + ADSB_APBUFFSZ + (base & 1);
        if (fire > m_ref) {
            *final = 1;
            break;
        }
        const uint64_t limit = fire - ADSB_DECOFFSET;
        if (limit > g_end)
            break;
        while (k < nF && F[k].g < limit) {
            last_end = F[k].g + frame_span(F[k]);
            k++;
        }
        base = last_end > limit ? last_end : limit;
    }
    return n;
}

// Returns 0, -1 (bad arguments / capacity), or -3: a seam cannot be decided from the head candidates alone (the chain did
// not re-synchronise inside the head window); the caller then falls back to gathering every candidate to one resolver.
inline int stitch_shards(const adsb_shard_part *parts, int n_parts, uint64_t total_samples, adsb_shard_fix *fix,
                         adsb_frame *new_frames, size_t new_cap, size_t *n_new_total, uint64_t *walk_stats = nullptr,
                         adsb_stats *stats = nullptr)
{
    if (!parts || n_parts <= 0 || !fix || !n_new_total || (new_cap && !new_frames))
        return -1;
    int64_t tries[3] = {0, 0, 0}, oks[3] = {0, 0, 0}, fixed = 0; // (oks / fixed: valid.c:53,75 over the final frames)
    auto count_ok = [&](const adsb_frame &f, int sign) {
        const unsigned df = f.frame[0] >> 3;
        oks[df == 11 ? 0 : df == 17 ? 1 : 2] += sign;
        fixed += sign * (int)(f.reserved & 1u);
    };
    std::vector<uint64_t> entry(stats ? (size_t)n_parts : 0); // per shard: offsets below are jumped by a frame of an earlier shard
    size_t n_new = 0;
    uint64_t skipped_global = 0; // offsets jumped by all final frames before the current shard
    uint64_t e_prev = 0;         // end of the last final frame so far
    for (int i = 0; i < n_parts; i++) {
        const adsb_shard_head &h = *parts[i].head;
        const adsb_frame *F = parts[i].frames;
        const adsb_candidate *H = parts[i].head_cands;
        const uint64_t nF = h.n_frames, nH = h.n_head;
        adsb_shard_fix &x = fix[i];
        std::memset(&x, 0, sizeof x);
        if (h.status != 0)
            return -1;
        uint64_t drop = 0, dropped_skip = 0, new_skip = 0;
        const size_t new_first = n_new;
        if (e_prev > h.g_begin && (nF || nH)) {
            uint64_t idx = e_prev;
            uint64_t hj = 0, fi = 0;
            for (;;) {
                while (hj < nH && H[hj].g < idx)
                    hj++; // inside an accepted frame: never evaluated
                while (fi < nF && F[fi].g < idx) { // speculative frames the true chain jumps over
                    dropped_skip += frame_span(F[fi]) - 1;
                    fi++;
                }
                if (hj < nH) {
                    const adsb_candidate &c = H[hj];
                    // (speculative frames in [idx, c.g) cannot exist: they would be head candidates themselves)
                    if (fi < nF && F[fi].g == c.g)
                        break; // the speculative chain accepted it as well: same chain from here on
                    if (n_new >= new_cap) {
                        *n_new_total = n_new + 1; // (at least: the caller grows new_frames and calls again)
                        return -2;
                    }
                    adsb_frame &f = new_frames[n_new++];
                    std::memset(&f, 0, sizeof f);
                    f.g = c.g;
                    f.ts = c.g + 1 - (skipped_global + new_skip);
                    f.pw = c.pw;
                    f.len = c.len;
                    std::memcpy(f.frame, c.frame, 14);
                    f.reserved = c.reserved;
                    new_skip += cand_span(c) - 1;
                    idx = c.g + cand_span(c);
                    continue;
                }
                // No head candidate at or behind idx.  Candidates beyond the head window are only known where the
                // speculative chain walked (there every candidate it met became a frame); inside a speculative frame
                // that the true chain does not accept there may be candidates nobody kept.
                const uint64_t known_from = idx > h.head_end ? idx : h.head_end;
                if (fi > 0 && F[fi - 1].g + frame_span(F[fi - 1]) > known_from) // (the last dropped frame ends last)
                    return -3;
                break; // the next candidate the true chain meets is the next speculative frame
            }
            drop = fi;
        }
        if (stats) {
            if (!h.has_tries)
                return -1;
            for (int k = 0; k < 3; k++) {
                tries[k] += (int64_t)h.tries[k];
                oks[k] += (int64_t)h.ok[k];
            }
            fixed += (int64_t)h.fixed;
            for (uint64_t q = 0; q < drop; q++) // the shard counted its speculative frames: the repair's verdict
                count_ok(F[q], -1);
            for (size_t q = new_first; q < n_new; q++)
                count_ok(new_frames[q], +1);
            entry[i] = e_prev > h.g_begin ? e_prev : h.g_begin;
            if (e_prev > h.g_begin) {
                // The shard counted its tries against its speculative frames.  From e_prev on the true chain is the frames
                // accepted above, then the speculative frames that stand; the two are the same chain from X on: behind the
                // last dropped frame, the last new one and e_prev (a candidate both accepted lies at or behind all three).
                uint64_t X = e_prev;
                if (drop)
                    X = std::max(X, F[drop - 1].g + frame_span(F[drop - 1]));
                if (n_new > new_first)
                    X = std::max(X, new_frames[n_new - 1].g + frame_span(new_frames[n_new - 1]));
                if (std::min(X, h.g_end) > parts[i].head_tries_end || (parts[i].n_head_tries && !parts[i].head_tries))
                    return -3; // the window of tries handed over does not reach that far
                uint64_t di = 0;            // cursor into the dropped frames F[0 .. drop)
                size_t ni = new_first;      // ... and into the frames accepted instead
                for (uint64_t q = 0; q < parts[i].n_head_tries; q++) {
                    const uint64_t g = parts[i].head_tries[q] >> 2;
                    if (g >= X)
                        break;
                    while (di < drop && F[di].g + frame_span(F[di]) <= g)
                        di++;
                    while (ni < n_new && new_frames[ni].g + frame_span(new_frames[ni]) <= g)
                        ni++;
                    const bool seen_spec = !(di < drop && F[di].g < g);   // not strictly inside a speculative frame
                    const bool seen_true = g >= e_prev && !(ni < n_new && new_frames[ni].g < g);
                    const unsigned code = (unsigned)(parts[i].head_tries[q] & 3u);
                    tries[code < 3 ? code : 2] += (int)seen_true - (int)seen_spec;
                }
            }
        }
        x.drop_front = drop;
        x.new_first = new_first;
        x.n_new = n_new - new_first;
        x.keep = nF - drop;
        x.ts_sub = (int64_t)(skipped_global + new_skip) - (int64_t)dropped_skip;
        const uint64_t shard_skip = h.skipped - dropped_skip + new_skip;
        if (x.keep)
            e_prev = F[nF - 1].g + frame_span(F[nF - 1]);
        else if (x.n_new)
            e_prev = new_frames[n_new - 1].g + frame_span(new_frames[n_new - 1]);
        skipped_global += shard_skip;
    }

    // The end-of-file horizon: replay the deqframe calls (Resolver::advance) over the final frames' positions.  The chain
    // is sequential -- a frame that straddles a call's limit moves every later call -- but it FORGETS: two chains
    // that reach the same base continue identically, and a chain is re-anchored at a frame's end by every straddle
    // (about one call in eight at 1 k frames/s).  Every rank has therefore walked the calls of its own shard from a
    // guessed entry base (walk_shard_calls: in parallel, ~3 400 calls for an eighth of 2 Gi samples) and left the bases
    // it went through; the true chain is walked here only until it meets one of them, then jumps to that shard's
    // exit.  What is left for this rank is the walk to the first common base of each shard, not the stream.
    const uint64_t m_ref = 2 * ((total_samples + 3) / 4); // a trailing partial quad still produces two power samples
    uint64_t base = 0, horizon = 0;
    int part = 0;
    uint64_t k = 0; // position inside the part's final frames: [new frames][kept speculative frames]
    auto frame_at = [&](int p, uint64_t q) -> const adsb_frame & {
        return q < fix[p].n_new ? new_frames[fix[p].new_first + q] : parts[p].frames[fix[p].drop_front + (q - fix[p].n_new)];
    };
    auto next_limit = [](uint64_t b) { return b + ADSB_APBUFFSZ + (b & 1) - ADSB_DECOFFSET; };
    uint64_t last_end = 0; // end of the last frame that starts below the current limit
    int js = 0;            // shard whose recorded bases the chain is currently compared with
    uint64_t jb = 0;
    uint64_t walked = 0, jumped = 0;
    for (;;) {
        const uint64_t fire = base + ADSB_APBUFFSZ + (base & 1); // air.c:94: tested after every second power sample
        if (fire > m_ref)
            break;
        const uint64_t limit = fire - ADSB_DECOFFSET; // demod.c:89
        while (js < n_parts && limit > parts[js].head->g_end) {
            js++;
            jb = 0;
        }
        if (js < n_parts && parts[js].bases && parts[js].head->n_bases > 1) {
            const adsb_shard_head &h = *parts[js].head;
            // the recorded walk only knew the shard's own speculative frames: it stands for calls whose limit lies where no
            // frame of the previous shard can reach (g_begin + 1200) and behind the part of the shard a seam repair rewrote
            const uint64_t valid_from = fix[js].keep ? parts[js].frames[fix[js].drop_front].g : ~0ull;
            const bool repaired = fix[js].drop_front || fix[js].n_new;
            if (limit >= h.g_begin + ADSB_DECOFFSET && (!repaired || limit > valid_from)) {
                const uint64_t *B = parts[js].bases;
                while (jb + 1 < h.n_bases && B[jb] < base)
                    jb++;
                if (jb + 1 < h.n_bases && B[jb] == base) { // same base: same chain from here to the shard's end
                    jumped += h.n_bases - 1 - jb;
                    horizon = next_limit(B[h.n_bases - 2]);
                    base = B[h.n_bases - 1];
                    // frames that can still straddle a later limit start at or behind base - 1200: re-seat the cursor there
                    part = js;
                    uint64_t lo = 0, hi = fix[js].n_new + fix[js].keep;
                    const uint64_t from = base >= ADSB_DECOFFSET ? base - ADSB_DECOFFSET : 0;
                    while (lo < hi) {
                        const uint64_t mid = (lo + hi) / 2;
                        if (frame_at(js, mid).g < from)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    k = lo;
                    last_end = 0;
                    continue;
                }
            }
        }
        while (part < n_parts) {
            const uint64_t n_here = fix[part].n_new + fix[part].keep;
            if (k >= n_here) {
                part++;
                k = 0;
                continue;
            }
            const adsb_frame &f = frame_at(part, k);
            if (f.g >= limit)
                break;
            last_end = f.g + frame_span(f);
            k++;
        }
        base = last_end > limit ? last_end : limit; // a frame accepted below the limit may jump past it (demod.c:128,134)
        horizon = limit;
        walked++;
    }
    if (walk_stats) {
        walk_stats[0] = walked;
        walk_stats[1] = jumped;
    }
    if (stats) {
        // Nothing at or beyond the horizon is visited (air.c:94-99): the shards counted those tries too -- unless they lie in
        // a frame of the chain as it stands BEFORE the cut below -- so take them out again.
        for (int p = 0; p < n_parts; p++) {
            const adsb_shard_head &h = *parts[p].head;
            if (h.g_end <= horizon || h.g_end <= h.g_begin)
                continue;
            if (parts[p].tail_from == ~0ull || parts[p].tail_from > std::max(horizon, h.g_begin) ||
                (parts[p].n_tail_tries && !parts[p].tail_tries))
                return -3;
            const uint64_t n_here = fix[p].n_new + fix[p].keep;
            // the first frame that can still cover an offset at or beyond the horizon starts above horizon - 1200: seek it
            // (the frames lie in another thread's memory: walking all of them from here is what must not happen)
            uint64_t q = 0, hi_q = n_here;
            const uint64_t from = horizon >= ADSB_DECOFFSET ? horizon - ADSB_DECOFFSET : 0;
            while (q < hi_q) {
                const uint64_t mid = (q + hi_q) / 2;
                if (frame_at(p, mid).g < from)
                    q = mid + 1;
                else
                    hi_q = mid;
            }
            for (uint64_t t = 0; t < parts[p].n_tail_tries; t++) {
                const uint64_t g = parts[p].tail_tries[t] >> 2;
                if (g < horizon)
                    continue;
                while (q < n_here && frame_at(p, q).g + frame_span(frame_at(p, q)) <= g)
                    q++;
                const bool seen = g >= entry[p] && !(q < n_here && frame_at(p, q).g < g);
                const unsigned code = (unsigned)(parts[p].tail_tries[t] & 3u);
                tries[code < 3 ? code : 2] -= (int)seen;
            }
        }
    }
    // frames at or beyond the horizon are never visited: cut them (they can only sit at the very end)
    for (int p = n_parts - 1; p >= 0; p--) {
        adsb_shard_fix &x = fix[p];
        const adsb_frame *F = parts[p].frames + x.drop_front;
        while (x.keep && F[x.keep - 1].g >= horizon) {
            if (stats)
                count_ok(F[x.keep - 1], -1);
            x.keep--;
        }
        if (x.keep)
            break;
        while (x.n_new && new_frames[x.new_first + x.n_new - 1].g >= horizon) {
            if (stats)
                count_ok(new_frames[x.new_first + x.n_new - 1], -1);
            x.n_new--;
        }
        if (x.n_new)
            break;
    }
    *n_new_total = n_new;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (int k = 0; k < 3; k++) {
            if (tries[k] < 0 || oks[k] < 0)
                return -1;
            stats->try_[k] = (uint64_t)tries[k];
            stats->ok[k] = (uint64_t)oks[k];
        }
        stats->fixed = (uint64_t)(fixed < 0 ? 0 : fixed);
    }
    return 0;
}

} // namespace adsb
