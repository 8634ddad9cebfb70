// resolver.hpp -- host-side replay of the reference's SEQUENTIAL rules over the
// sparse candidate records the GPU emits.
//
// The GPU evaluates every preamble offset independently; three things in the
// reference are inherently sequential and are replayed here, once per stream:
//   1. greedy advance  (demod.c:89,128,134,141): an accepted frame makes the scan
//      jump to the end of that frame, everything else advances by one;
//   2. the timestamp   (demod.c:86,99): ts counts loop passes, not samples;
//   3. the call pattern of deqframe (air.c:94-99): it fires when 40980 power
//      samples are buffered, scans [G, T-1200) and carries the rest, so the last
//      ~41k power samples of a stream are never scanned.
// Also valid.c:30-31's Try/Ok counters, which only count VISITED offsets.
//
// Cost matters: at device speed the host sees ~2e8 candidates per second, so the
// queues are flat vectors with a head index, and records are consumed in place -- in two
// tight loops (decide_calls: 16 bytes per accepted frame; FormatGang::format: the frames),
// the second of which, and on a full channel the first as well, other threads can run
// (gang.hpp; speculate_tiles / run_calls_tiles_ahead below).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <memory>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "../../include/adsbdec_amd_diag.h"
#include "scan_kernel_format.h"
#include "gang.hpp"

namespace adsb {

// The accepted frames: a flat array of PODs whose slots are handed out uninitialised (a std::vector value-initialises 40
// bytes per emplace_back and checks its capacity per frame: at the channel's capacity the resolver emits 100 k frames per
// launch and every nanosecond per frame is a tenth of the kernel's time).
class FrameVec {
public:
    FrameVec() = default;
    FrameVec(const FrameVec &) = delete;
    FrameVec &operator=(const FrameVec &) = delete;
    ~FrameVec() { std::free(p_); }
    size_t size() const { return n_; }
    adsb_frame *data() { return p_; }
    const adsb_frame *data() const { return p_; }
    void clear() { n_ = 0; }
    // room for `more` frames behind the ones that are there; the pointer to the first free slot
    adsb_frame *room(size_t more)
    {
        if (n_ + more > cap_) {
            size_t cap = cap_ ? cap_ : 1024;
            while (cap < n_ + more)
                cap *= 2;
            void *q = std::realloc(p_, cap * sizeof(adsb_frame));
            if (!q)
                throw std::bad_alloc();
            p_ = static_cast<adsb_frame *>(q);
            cap_ = cap;
        }
        return p_ + n_;
    }
    void grew(size_t k) { n_ += k; } // k slots behind room()'s pointer have been filled
    bool fits(size_t more) const { return n_ + more <= cap_; } // room(more) would not move the array
    adsb_frame &push()
    {
        adsb_frame *f = room(1);
        n_++;
        return *f;
    }

private:
    adsb_frame *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

class Resolver {
public:
    // More hands (gang.hpp): from now on a batch of tiles that yields kGangMinFrames frames or more is only DECIDED here; the
    // frames are written by the gang's threads.  Every way to the frames or the counters -- drain, take, stats, reset, an
    // output array that has to move -- waits for them (sync); whoever recycles the hand-off stream calls sync() first.
    static constexpr size_t kGangMinFrames = 1024; // (a hand-over has a latency: a small batch is written faster than handed on)
    void set_gang(FormatGang *g, size_t min_frames = kGangMinFrames)
    {
        sync();
        gang_ = g;
        gang_min_frames_ = min_frames;
    }
    void sync()
    {
        if (!gang_busy_)
            return;
        gang_->wait_all();
        const FormatGang::Counts c = gang_->take_counts();
        stats_.ok[0] += c.n11;
        stats_.ok[1] += c.n17;
        stats_.ok[2] += gang_frames_ - c.n11 - c.n17;
        stats_.fixed += c.nfix;
        gang_frames_ = 0;
        gang_busy_ = false;
        if (!arena_pins_)
            arena_.reset();
    }

    void reset()
    {
        sync(); // (also lets every batch that is being decided ahead end)
        for (Ahead &a : ahead_)
            a.posted = false;
        ahead_head_ = ahead_tail_ = 0;
        ahead_now_ = nullptr;
        arena_pins_ = 0;
        arena_.reset();
        base_ = 0;
        chain_ = false;
        w_on_ = false;
        w_n_ = 0;
        w_final_ = w_stop_ = false;
        w_acc_.clear();
        w_acc_head_ = 0;
        head_end_ = 0;
        head_ = nullptr;
        skipped_ = 0;
        cands_.clear();
        chead_ = 0;
        tries_.clear();
        thead_ = 0;
        out_.clear();
        ohead_ = 0;
        log_.clear();
        ext_n_ = 0;
        std::memset(&stats_, 0, sizeof stats_);
    }

    // Records must arrive in ascending g over the life of the stream.
    void feed(const adsb_candidate *c, size_t n, const uint64_t *tries, size_t nt)
    {
        compact();
        if (head_)
            for (size_t i = 0; i < n && c[i].g < head_end_; i++)
                head_->push_back(c[i]);
        if (n)
            cands_.insert(cands_.end(), c, c + n);
        if (nt)
            tries_.insert(tries_.end(), tries, tries + nt);
    }

    // Device records: {g_rel, pw, frame bytes 0..13, len in byte 14} = 6 dwords starting
    // at recs[order[i] * words + off], visited through `order` (ascending g_rel); tries
    // ascending ((g_rel << 2) | code).
    void feed_device(const uint32_t *recs, const uint32_t *order, size_t n, int words, int off, uint64_t g_base,
                     const uint32_t *tries, size_t nt)
    {
        compact();
        const size_t at = cands_.size();
        cands_.resize(at + n);
        for (size_t i = 0; i < n; i++) {
            const uint32_t *r = recs + (size_t)order[i] * words + off;
            adsb_candidate &c = cands_[at + i];
            c.g = g_base + r[0];
            c.pw = r[1];
            std::memcpy(c.frame, &r[2], 14);
            c.len = (uint8_t)((r[5] >> 16) & 0xFF);
            c.reserved = (uint8_t)((r[5] >> 24) & 1u); // bit 0: repaired by the 1-bit extension
        }
        const size_t tat = tries_.size();
        tries_.resize(tat + nt);
        for (size_t i = 0; i < nt; i++)
            tries_[tat + i] = (((uint64_t)(tries[i] >> 2) + g_base) << 2) | (tries[i] & 3u);
    }

    // feed_device + advance in one step, without copying the records into the queue:
    // the device batch is consumed in place; only what lies beyond the last executed
    // deqframe call (normally nothing or a handful) is kept for later.
    void advance_device(const uint32_t *recs, const uint32_t *order, size_t n, int words, int off, uint64_t g_base,
                        uint64_t power_samples, uint64_t g_complete)
    {
        compact();
        batch_ = Batch{};
        batch_.recs = recs + off;
        batch_.order = order;
        batch_.n = n;
        batch_.words = words;
        batch_.g_base = g_base;
        batch_.cur = n ? batch_.recs + (size_t)order[0] * words : nullptr;
        advance(power_samples, g_complete);
        keep_leftovers();
    }

    // The same for tiles [t0, t1) of a hand-off stream (scan_kernel.h): tile u's records
    // are the counts[u] consecutive 8-dword records from granule starts[u] on, each
    // {g_rel, pw, frame | len << 16 | flags << 24}, ascending inside a
    // tile and from tile to tile.
    void advance_tiles(const uint32_t *stream, const uint32_t *starts, const uint32_t *counts, uint32_t t0, uint32_t t1,
                       uint64_t g_base, uint64_t power_samples, uint64_t g_complete)
    {
        compact();
        batch_ = Batch{};
        batch_.recs = stream; // (until round 6 a dword offset `off` could be added here; every caller passed 0, and a batch
                              // decided ahead -- speculate_tiles -- never knew about it: the parameter is gone)
        batch_.starts = starts;
        batch_.counts = counts;
        batch_.u_end = t1;
        batch_.g_base = g_base;
        batch_.seek_tile(t0);
        for (uint32_t u = t0; u < t1; u++)
            batch_.records_left += counts[u];
        Ahead &a = ahead_[ahead_head_ % kAheadSlots]; // the oldest batch that was posted, if any
        if (a.posted) { // (decided ahead: speculate_tiles)
            if (a.stream == stream && a.t0 == t0 && a.t1 == t1 && a.g_base == g_base)
                ahead_now_ = &a;
            else
                for (uint32_t spins = 1; a.n.load(std::memory_order_acquire) < 0; spins++) // not this batch: let it end, and forget it
                    FormatGang::relax(spins);
        }
        advance(power_samples, g_complete);
        if (a.posted) {
            // (a chain whose position already lies beyond this batch never enters the call that would have waited for it:
            // the arena must not be unpinned under a thread that is still writing the batch's decisions)
            for (uint32_t spins = 1; a.n.load(std::memory_order_acquire) < 0; spins++)
                if (!gang_ || !gang_->help())
                    FormatGang::relax(spins);
            ahead_now_ = nullptr;
            a.posted = false;
            arena_pins_--;
            ahead_head_++;
        }
        keep_leftovers();
    }

    // CHAIN mode (time-sharded streams, one resolver per shard): only the greedy rule itself (demod.c:89,128,134,
    // 141), started at offset g_begin as if no earlier frame reached into the shard.  The deqframe call pattern
    // (air.c:94-99) does not change which frames the chain accepts -- a call ends at an offset, the next one starts at
    // that offset -- only where the stream's LAST call ends (the end-of-file horizon), and the stitcher replays that
    // on the accepted frames (stitch.hpp).  ts comes out local: g + 1 - (offsets jumped inside this shard).
    // Every candidate with g < head_end is also copied to `head`: what the stitcher needs to re-run the chain
    // across the seam when a frame of the previous shard does reach in.
    void start_chain(uint64_t g_begin, uint64_t head_end, std::vector<adsb_candidate> *head)
    {
        reset();
        base_ = g_begin;
        chain_ = true;
        head_end_ = head_end;
        head_ = head;
    }
    uint64_t skipped() const { return skipped_; }
    // ... and, beside the chain, the shard's own walk of the deqframe CALLS (air.c:94-99; stitch.hpp walk_shard_calls) from
    // the guessed entry base g_begin, advanced as the frames come in -- a call can be replayed once the chain has passed its
    // limit -- so that it costs the rank nothing behind its scan.  bases[0 .. min(n, cap)) as walk_shard_calls leaves them.
    void start_walk(uint64_t g_begin, uint64_t g_end, uint64_t total_samples, uint64_t *bases, size_t cap)
    {
        w_on_ = true;
        w_base_ = g_begin;
        w_end_ = g_end;
        w_mref_ = 2 * ((total_samples + 3) / 4);
        w_bases_ = bases;
        w_cap_ = cap;
        w_n_ = 0;
        w_last_end_ = 0;
        w_final_ = w_stop_ = false;
        w_acc_.clear();
        w_acc_head_ = 0;
        walk_record();
    }
    size_t walk_bases() const { return w_n_; }
    bool walk_final() const { return w_final_; }
    bool walk_complete() const { return w_stop_; }
    // a device batch about to be consumed: copy its head candidates (ascending; batches arrive in ascending g)
    void capture_head(const uint32_t *recs, const uint32_t *order, size_t n, int words, int off, uint64_t g_base)
    {
        if (!head_)
            return;
        for (size_t i = 0; i < n; i++) {
            const uint32_t *r = recs + (size_t)order[i] * words + off;
            if (g_base + r[0] >= head_end_)
                break;
            push_head(r, g_base);
        }
    }
    void capture_head_tiles(const uint32_t *stream, const uint32_t *starts, const uint32_t *counts, uint32_t t0, uint32_t t1,
                            uint64_t g_base)
    {
        if (!head_)
            return;
        for (uint32_t u = t0; u < t1; u++)
            for (uint32_t i = 0; i < counts[u]; i++) {
                const uint32_t *r = stream + ((size_t)starts[u] + 2 * (size_t)i) * 4;
                for (uint32_t k = 0, nk = rec_copies(r); k < nk; k++) {
                    if (g_base + r[0] + k >= head_end_)
                        return;
                    push_head(r, g_base, k);
                }
            }
    }
    bool head_wanted(uint64_t g_from) const { return head_ && g_from < head_end_; }

    // power_samples: samples the front end has produced so far (air.c `aidx` grows
    // by two per loop pass, so this is even); g_complete: every record with
    // g < g_complete has been fed.
    void advance(uint64_t power_samples, uint64_t g_complete)
    {
        if (chain_) {
            if (g_complete > base_)
                run_call(g_complete);
            if (w_on_)
                walk_advance();
            return;
        }
        for (;;) {
            if (tiles_fast_path()) { // every remaining call of this advance in one tight loop (run_calls_tiles)
                run_calls_tiles(power_samples, g_complete, 0);
                break;
            }
            // air.c:94: the test `aidx >= APBUFFSZ` is made after every second
            // power sample, so the call fires at the first EVEN total T with
            // T - base >= 40980.
            const uint64_t fire = base_ + ADSB_APBUFFSZ + (base_ & 1);
            if (fire > power_samples)
                break; // the reference has not called deqframe yet (never, at EOF)
            const uint64_t limit = fire - ADSB_DECOFFSET; // demod.c:89 `idx < len-DECOFFSET`
            if (limit > g_complete)
                break; // the device has not scanned that far yet
            // Nothing pending before the limit: the call only moves the base, and so
            // does every following call up to the next record (skip them in one step).
            run_call(limit);
        }
    }

    // Statistics runs that keep the try words on the device: log (g, span) of every
    // accepted frame for the device-side visited-try count, and take the counts back.
    void log_accepted(bool on) { log_on_ = on; }
    std::vector<std::pair<uint64_t, uint32_t>> &accepted_log() { return log_; }
    // ... straight into a buffer of the caller's (the pinned array the count pass uploads from: no copy when the pass is
    // prepared); what does not fit goes on to the vector above, behind the buffer's entries
    struct LogEntry {
        uint64_t g;
        uint32_t span, pad;
    };
    void log_into(LogEntry *buf, size_t cap)
    {
        ext_ = buf;
        ext_cap_ = cap;
        ext_n_ = 0;
        log_.clear();
    }
    size_t logged_ext() const { return ext_n_; }
    void log_clear()
    {
        ext_n_ = 0;
        log_.clear();
    }
    void set_tries(uint64_t df11, uint64_t df17, uint64_t df18)
    {
        stats_.try_[0] = df11;
        stats_.try_[1] = df17;
        stats_.try_[2] = df18;
    }

    size_t pending() const { return out_.size() - ohead_; }
    size_t drain(adsb_frame *dst, size_t cap)
    {
        sync();
        size_t n = out_.size() - ohead_;
        if (n > cap)
            n = cap;
        if (n)
            std::memcpy(dst, out_.data() + ohead_, n * sizeof(adsb_frame));
        ohead_ += n;
        if (ohead_ == out_.size()) {
            out_.clear();
            ohead_ = 0;
        }
        return n;
    }
    // drain without the copy: the pending frames where they lie (valid until the next feed /
    // advance / reset, which recycles the storage); they count as drained
    size_t take(const adsb_frame **p)
    {
        sync();
        const size_t n = out_.size() - ohead_;
        *p = n ? out_.data() + ohead_ : nullptr;
        ohead_ = out_.size();
        return n;
    }
    const adsb_stats &stats()
    {
        sync();
        return stats_;
    }
    uint64_t base() const { return base_; }

private:
    static int df_slot(uint8_t byte0)
    {
        switch (byte0 >> 3) {
        case 11: return 0;
        case 17: return 1;
        default: return 2;
        }
    }

    void walk_record()
    {
        if (w_n_ < w_cap_)
            w_bases_[w_n_] = w_base_;
        w_n_++;
    }
    void walk_advance()
    {
        while (!w_stop_) {
            const uint64_t fire = w_base_ + ADSB_APBUFFSZ + (w_base_ & 1); // air.c:94
            if (fire > w_mref_) {
                w_final_ = w_stop_ = true; // the stream ends before this call fires
                break;
            }
            const uint64_t limit = fire - ADSB_DECOFFSET; // demod.c:89
            if (limit > w_end_) {
                w_stop_ = true; // the next shard's call
                break;
            }
            if (limit > base_)
                break; // the chain has not passed the limit yet: a frame below it may still be accepted
            // (the walk has its own record of the accepted frames: the output queue may have been drained meanwhile)
            while (w_acc_head_ < w_acc_.size() && w_acc_[w_acc_head_].first < limit)
                w_last_end_ = w_acc_[w_acc_head_++].second;
            if (w_acc_head_ == w_acc_.size()) {
                w_acc_.clear();
                w_acc_head_ = 0;
            }
            w_base_ = w_last_end_ > limit ? w_last_end_ : limit;
            walk_record();
        }
    }

    void push_head(const uint32_t *r, uint64_t g_base, uint32_t copy = 0)
    {
        head_->emplace_back();
        adsb_candidate &c = head_->back();
        c.g = g_base + r[0] + copy;
        c.pw = copy ? r[5 + copy] : r[1];
        std::memcpy(c.frame, &r[2], 14);
        c.len = (uint8_t)((r[5] >> 16) & 0xFF);
        c.reserved = (uint8_t)((r[5] >> 24) & 1u);
    }

    void compact()
    {
        if (ohead_ && ohead_ == out_.size()) { // everything handed out (take): recycle
            out_.clear();
            ohead_ = 0;
        }
        if (chead_ && chead_ == cands_.size()) {
            cands_.clear();
            chead_ = 0;
        } else if (chead_ > (1u << 16)) {
            cands_.erase(cands_.begin(), cands_.begin() + (ptrdiff_t)chead_);
            chead_ = 0;
        }
        if (thead_ && thead_ == tries_.size()) {
            tries_.clear();
            thead_ = 0;
        } else if (thead_ > (1u << 18)) {
            tries_.erase(tries_.begin(), tries_.begin() + (ptrdiff_t)thead_);
            thead_ = 0;
        }
    }

    // valid.c:46,68: one Try per visited offset that passed the DF gate.
    void count_tries(uint64_t from, uint64_t to_inclusive)
    {
        const size_t n = tries_.size();
        while (thead_ < n && (tries_[thead_] >> 2) < from)
            thead_++; // shadowed by an accepted frame: never visited
        while (thead_ < n && (tries_[thead_] >> 2) <= to_inclusive)
            stats_.try_[tries_[thead_++] & 3]++;
    }

    // A batch of device records being consumed in place (advance_device).  Queue
    // entries (older) always precede batch entries (newer) in g.
    struct Batch {
        const uint32_t *cur = nullptr;  // {g_rel, pw, w0..w3} of the next record; null: exhausted
        const uint32_t *recs = nullptr; // dword 0 of index 0 (the layout's offset already applied)
        uint64_t g_base = 0;
        // either: records visited through an index list
        const uint32_t *order = nullptr;
        size_t n = 0, pos = 0;
        int words = 0;
        // or: tile ranges of a granule stream
        const uint32_t *starts = nullptr, *counts = nullptr;
        uint32_t u = 0, u_end = 0, left = 0;
        // a stream record stands for `copies` candidates at consecutive offsets (scan_kernel_format.h); `sub` is the one `cur`
        // means at the moment.  Index-list records (6 words) are always single.
        uint32_t sub = 0, copies = 1;
        size_t records_left = 0; // tile mode: records of the batch (an upper bound of the frames it can yield)

        uint64_t g() const { return g_base + cur[0] + sub; }
        uint32_t pw() const { return sub ? cur[5 + sub] : cur[1]; }
        void seek_tile(uint32_t t)
        {
            sub = 0;
            for (u = t; u < u_end; u++)
                if (counts[u]) {
                    left = counts[u];
                    cur = recs + (size_t)starts[u] * 4;
                    copies = rec_copies(cur);
                    return;
                }
            cur = nullptr;
        }
        void next_record()
        {
            sub = 0;
            if (order) {
                pos++;
                cur = pos < n ? recs + (size_t)order[pos] * words : nullptr;
            } else if (--left) {
                cur += 8; // two granules per record
                copies = rec_copies(cur);
            } else {
                seek_tile(u + 1);
            }
        }
        void next()
        {
            if (++sub >= copies)
                next_record();
        }
        // the first candidate at or beyond idx
        void skip_below(uint64_t idx)
        {
            while (cur) {
                const uint64_t g0 = g_base + cur[0];
                if (g0 + copies > idx) { // the record's last copy is at g0 + copies - 1
                    if (g0 + sub < idx)
                        sub = (uint32_t)(idx - g0);
                    return;
                }
                next_record();
            }
        }
    };

    void keep_leftovers() // records beyond the last executed call's limit wait in the queue
    {
        for (; batch_.cur; batch_.next()) {
            const uint32_t *r = batch_.cur;
            cands_.emplace_back();
            adsb_candidate &c = cands_.back();
            c.g = batch_.g();
            c.pw = batch_.pw();
            std::memcpy(c.frame, &r[2], 14);
            c.len = (uint8_t)((r[5] >> 16) & 0xFF);
            c.reserved = (uint8_t)((r[5] >> 24) & 1u);
        }
        batch_ = Batch{};
    }

    // run_call for the usual case -- nothing waiting in the queue, the candidates come from the tile ranges of a hand-off
    // stream where they lie -- in two tight loops: DECIDE (which record is accepted, at which of its offsets, with which ts:
    // 16 bytes per accepted frame, gang.hpp Decision), then FORMAT (the 40-byte frames, from the decisions and the records).
    // At the channel's capacity (BASELINE configs[2]: 106 k accepted frames out of 123 k records per 256 Mi-sample launch)
    // the general loop below cost 8 ns per accepted frame, four times the kernel's share; and the second loop is work that
    // other threads can do (set_gang).
    // (host-side try words -- per-shard scans that hand the list back -- are counted between the frames: the general loop)
    bool tiles_fast_path() const { return chead_ == cands_.size() && !batch_.order && batch_.starts && thead_ == tries_.size(); }

    // The deciding loop, written for the register file -- twice.  Round 5's first form was one loop that decided and wrote
    // the frame, inside the function that also walks the tiles and the deqframe calls: some twenty values alive, and both
    // compilers (g++ for the tests, clang for the library) kept the ones that CHANGE per record -- the position idx, the ts
    // count, the output pointer -- on the stack: load-modify-store chains of six cycles each beside a dependent chain of
    // three (idx -> max(idx, g0) -> idx + span).  So: a function of its own (noinline), in which everything that is touched
    // once per tile or once per call stays in memory (this struct, read through the reference) and the seven values of the
    // inner loop are all there is to allocate: cursor, tile end, idx, ts count, output, limit, the stream's base.
    // And no branch on a record's fate: on a full channel one record in seven lies inside the frame before it, at random --
    // as a branch that is a misprediction every few records, 5 ns each, more than the loop's own work.  The slot at `o` is
    // written either way and kept by advancing `o`; the array has a slot to spare.
    struct Walk {
        const uint32_t *cur, *tile_end; // the next record, and the end of its tile's records (records are two granules)
        uint64_t base, tsb;             // where the next call starts; g_base + 1 - (offsets jumped so far): ts = g_rel + tsb
        uint64_t *o;                    // the next decision
        const uint32_t *recs, *starts, *counts;
        uint32_t u, u_end;
        uint64_t g_base, power_samples, g_complete, single_limit;
        // a call can be left and re-entered in its middle (a batch decided ahead is taken over tile by tile, then decision by
        // decision: run_calls_tiles_ahead): the position and the limit, relative to g_base
        int64_t idx = 0, lim = 0;
        bool in_call = false;         // in: resume the call at (idx, lim); out: left inside a call
        bool stop_when_empty = false; // leave when the records are used up, inside the call, instead of finishing the calls
    };
    // single_limit != 0: ONE call with that limit (chain mode: run_call(g_complete)).  Else: the stream's deqframe calls, one
    // after the other while they have fired (air.c:94: at the first EVEN total T with T - base >= 40980) and the device has
    // scanned up to their limit (demod.c:89: T - 1200) -- the loop of advance(), inside: on sparse input a call holds four
    // frames, and entering and leaving this function once per call cost as much as the frames.
    __attribute__((noinline)) static void decide_calls(Walk &w)
    {
        const uint32_t *cur = w.cur, *end = w.tile_end;
        const uint32_t *const recs = w.recs;
        uint64_t base = w.base, tsb = w.tsb;
        uint64_t *o = w.o;
        bool resume = w.in_call;
        w.in_call = false;
        for (;;) { // one deqframe call per round
            // offsets relative to the launch's first from here on (signed: the call may start before the launch)
            int64_t idx, lim;
            if (resume) {
                idx = w.idx, lim = w.lim;
                resume = false;
            } else {
                uint64_t limit = w.single_limit;
                if (!limit) {
                    const uint64_t fire = base + ADSB_APBUFFSZ + (base & 1);
                    if (fire > w.power_samples)
                        break; // the reference has not called deqframe yet (never, at EOF)
                    limit = fire - ADSB_DECOFFSET;
                    if (limit > w.g_complete)
                        break; // the device has not scanned that far yet
                }
                idx = (int64_t)(base - w.g_base);
                lim = (int64_t)(limit - w.g_base);
            }
            while (cur) {
                if (cur == end) { // on to the next tile that has records
                    cur = nullptr;
                    for (uint32_t u = w.u + 1; u < w.u_end; u++)
                        if (w.counts[u]) {
                            cur = recs + (size_t)w.starts[u] * 4;
                            end = cur + 8 * (size_t)w.counts[u];
                            w.u = u;
                            break;
                        }
                    if (!cur)
                        w.u = w.u_end;
                    continue;
                }
                for (;;) {
                    const uint32_t w3 = cur[5];
                    const int64_t g0 = (int64_t)cur[0];
                    const int64_t g = idx > g0 ? idx : g0; // the first of the record's offsets the scan can visit, if any ...
                    if (g >= lim)
                        goto call_over; // (... which lies ahead: an offset inside an accepted frame is below idx, and idx < lim here)
                    const uint64_t take = 0 - (uint64_t)((uint64_t)(g - g0) < 1u + ((w3 >> kRecCopiesShift) & 3u)); // all ones: g is one of its offsets
                    const uint64_t span = 80 + 80 * (uint64_t)((w3 >> 16) & 0xFFu); // demod.c:109,120,123: lidx
                    o[0] = (uint64_t)g + tsb; // demod.c:99: one ts++ per visited offset
                    reinterpret_cast<uint32_t *>(o)[2] = (uint32_t)g;
                    reinterpret_cast<uint32_t *>(o)[3] =
                        (uint32_t)(reinterpret_cast<const char *>(cur) - reinterpret_cast<const char *>(recs)) | ((w3 >> 19) & 1u); // (len 14 = 0b1110, 7 = 0b0111)
                    o += 2 & take;
                    tsb -= (span - 1) & take;
                    idx ^= (idx ^ (g + (int64_t)span)) & (int64_t)take; // demod.c:128,134 -- the record's other offsets lie inside this frame
                    cur += 8;
                    if (idx >= lim)
                        goto call_over;
                    if (cur == end)
                        break;
                }
            }
            if (w.stop_when_empty) { // (no record left, the call not over)
                w.idx = idx, w.lim = lim, w.in_call = true;
                break;
            }
        call_over:
            if (idx < lim)
                idx = lim; // no candidate left below the limit: all remaining offsets advance by one (demod.c:141)
            base = w.g_base + (uint64_t)idx; // deqframe's return value; air.c:96-98 carries the rest
            if (w.single_limit)
                break;
        }
        w.cur = cur, w.tile_end = end, w.base = base, w.tsb = tsb, w.o = o;
    }

    void run_calls_tiles(uint64_t power_samples, uint64_t g_complete, uint64_t single_limit)
    {
        if (ahead_now_) {
            run_calls_tiles_ahead(power_samples, g_complete, single_limit);
            return;
        }
        Batch &b = batch_;
        Decision *const dc = arena_.alloc(b.records_left + 1); // an accepted frame per record at most, and the slot to spare
        Walk w;
        w.cur = b.cur, w.tile_end = b.cur ? b.cur + 8 * (size_t)b.left : nullptr;
        w.base = base_, w.tsb = b.g_base + 1 - skipped_;
        w.o = reinterpret_cast<uint64_t *>(dc);
        w.recs = b.recs, w.starts = b.starts, w.counts = b.counts;
        w.u = b.u, w.u_end = b.u_end;
        w.g_base = b.g_base, w.power_samples = power_samples, w.g_complete = g_complete, w.single_limit = single_limit;
        decide_calls(w);
        const size_t made = (size_t)(reinterpret_cast<Decision *>(w.o) - dc);
        if (w.cur && w.cur == w.tile_end) { // (the cursor never rests behind a tile's last record)
            w.cur = nullptr;
            for (w.u++; w.u < w.u_end; w.u++)
                if (w.counts[w.u]) {
                    w.cur = w.recs + (size_t)w.starts[w.u] * 4;
                    w.tile_end = w.cur + 8 * (size_t)w.counts[w.u];
                    break;
                }
        }
        b.cur = w.cur;
        b.left = w.cur ? (uint32_t)((w.tile_end - w.cur) / 8) : 0;
        b.copies = w.cur ? rec_copies(w.cur) : 1;
        b.u = w.u;
        b.sub = 0;
        skipped_ = b.g_base + 1 - w.tsb;
        base_ = w.base;
        arena_pins_++; // (emit may have to wait for the gang: the arena must not start over under dc)
        const bool handed = emit(dc, made);
        arena_pins_--;
        arena_.shrink_to(dc + (handed ? made : 0));
    }
    // `made` decisions become frames: written by the gang's threads (true: the decisions must stay where they are until
    // sync()) or by this thread, right away (false).
    // (packed: the decisions' offsets as adopt_calls read them -- a batch decided ahead: the decisions themselves lie in another
    // core's cache, and a statistics run's log must not fetch them line by line)
    bool emit(const Decision *dc, size_t made, uint64_t ts_add = 0, const uint32_t *packed = nullptr)
    {
        if (!made)
            return false;
        const Batch &b = batch_;
        if (gang_busy_ && !out_.fits(made))
            sync(); // (the array is about to move under the gang's hands)
        adsb_frame *const dst = out_.room(made);
        out_.grew(made);
        FormatGang::Task t;
        t.stream = b.recs;
        t.g_base = b.g_base;
        t.ts_add = ts_add;
        const bool handed = gang_ && made >= gang_min_frames_;
        if (handed) { // the frames, the Ok row and the repair count are the gang's work
            for (size_t i = 0; i < made; i += FormatGang::kBlock) {
                t.dec = dc + i;
                t.n = (uint32_t)std::min<size_t>(FormatGang::kBlock, made - i);
                t.dst = dst + i;
                gang_->post(t);
            }
            gang_busy_ = true;
            gang_frames_ += made;
        } else { // ... or this thread's
            t.dec = dc;
            t.n = (uint32_t)made;
            t.dst = dst;
            FormatGang::Counts c;
            FormatGang::format(t, c);
            stats_.ok[0] += c.n11;
            stats_.ok[1] += c.n17;
            stats_.ok[2] += made - c.n11 - c.n17;
            stats_.fixed += c.nfix;
        }
        if ((w_on_ && !w_stop_) || log_on_) { // what a shard's call walk or a statistics run wants per frame
            size_t i = 0;
            if (log_on_ && !(w_on_ && !w_stop_) && log_.empty()) {
                // the usual case, with the log's end in locals (through the members every entry is a load-add-store of
                // ext_n_ that the next one waits for -- the entries could alias it: 2.7 ns per frame on a full channel)
                const size_t n = std::min(made, ext_cap_ - std::min(ext_cap_, ext_n_));
                LogEntry *const e = ext_ + ext_n_;
                const uint64_t g_base = b.g_base;
                if (packed)
                    for (; i < n; i++)
                        e[i] = LogEntry{g_base + (packed[i] & 0x7FFFFFFFu), packed[i] >> 31 ? 1200u : 640u, 0};
                else
                    for (; i < n; i++)
                        e[i] = LogEntry{g_base + dc[i].g_rel, dc[i].where & kDecLong ? 1200u : 640u, 0};
                ext_n_ += n;
            }
            for (; i < made; i++)
                if (packed)
                    note_accepted(b.g_base + (packed[i] & 0x7FFFFFFFu), packed[i] >> 31 ? 1200u : 640u);
                else
                    note_accepted(b.g_base + dc[i].g_rel, dc[i].where & kDecLong ? 1200u : 640u);
        }
        return handed;
    }

    // ---- a batch decided AHEAD (speculate_tiles) -----------------------------------------------------------------------------
    // The greedy rule forgets: whatever the position was before, once the scan accepts a frame at offset g the rest follows
    // from g alone.  So a batch of tiles can be decided before the position at its entry is known -- as if no earlier frame
    // reached into it, in ONE call without a limit, ts counted from zero -- by another thread, while this one is busy with the
    // batch before.  Taking such a batch over: decide its first tile for real (from the true position, under the true calls);
    // if the last decision of that tile is one the other thread made too, everything behind it is the same chain -- only the
    // ts count (a constant to add) and the call pattern (air.c:94-99: which decision is the first of a new call, and where
    // the stream's last executed call ends) are missing, and adopt_calls() puts them in at ten instructions per decision
    // instead of forty per record.  No match (a frame of the tile before reaches far in): the next tile for real, and so on.
    struct Ahead {
        const uint32_t *stream = nullptr, *starts = nullptr, *counts = nullptr;
        uint32_t t0 = 0, t1 = 0;
        uint64_t g_base = 0;
        Decision *d = nullptr;
        uint32_t *g = nullptr;      // g_rel | (a 112-bit frame) << 31 of every decision, packed: what adopt_calls reads (a launch has < 2^31 offsets)
        std::atomic<int64_t> n{-1}; // decisions made; -1: not decided yet
        bool posted = false;
    };
    static void decide_ahead(const FormatGang::Task &t, FormatGang::Counts &)
    {
        Ahead &a = *static_cast<Ahead *>(t.ctx);
        Walk w;
        w.cur = w.tile_end = nullptr;
        w.recs = a.stream, w.starts = a.starts, w.counts = a.counts;
        w.u = a.t1, w.u_end = a.t1;
        for (uint32_t u = a.t0; u < a.t1; u++)
            if (a.counts[u]) {
                w.cur = a.stream + (size_t)a.starts[u] * 4;
                w.tile_end = w.cur + 8 * (size_t)a.counts[u];
                w.u = u;
                break;
            }
        w.base = a.g_base, w.tsb = 0; // position 0 of the launch, no offset jumped yet
        w.o = reinterpret_cast<uint64_t *>(a.d);
        w.g_base = a.g_base, w.power_samples = w.g_complete = 0;
        w.single_limit = a.g_base + (1ull << 62);
        decide_calls(w);
        const int64_t n = reinterpret_cast<Decision *>(w.o) - a.d;
        for (int64_t i = 0; i < n; i++)
            a.g[i] = a.d[i].g_rel | (a.d[i].where & kDecLong) << 31;
        a.n.store(n, std::memory_order_release);
    }

public:
    static constexpr size_t kAheadMinRecords = 2048;
    // Tiles [t0, t1) will be handed to advance_tiles() later, behind the batches posted before (at most kAheadSlots wait): have
    // them decided ahead.  Returns false if that is not to be had (no gang, a small batch): advance_tiles() then decides itself.
    bool speculate_tiles(const uint32_t *stream, const uint32_t *starts, const uint32_t *counts, uint32_t t0, uint32_t t1, uint64_t g_base)
    {
        Ahead &a = ahead_[ahead_tail_ % kAheadSlots];
        // (round 6: a chain -- a shard of the multi-GPU driver -- is decided ahead too: it is the simpler case, one call without
        // a pattern of calls; its walk of the deqframe calls and its head candidates ride on emit() and on capture_head_tiles(),
        // not on the deciding loop)
        if (!gang_ || ahead_tail_ - ahead_head_ == kAheadSlots)
            return false;
        size_t records = 0;
        for (uint32_t u = t0; u < t1; u++)
            records += counts[u];
        if (records < ahead_min_records_)
            return false;
        a.stream = stream, a.starts = starts, a.counts = counts, a.t0 = t0, a.t1 = t1, a.g_base = g_base;
        a.d = arena_.alloc(records + 1 + (records + 4) / 4); // (the packed offsets behind the decisions: 4 bytes each)
        a.g = reinterpret_cast<uint32_t *>(a.d + records + 1);
        a.n.store(-1, std::memory_order_relaxed);
        a.posted = true;
        arena_pins_++;
        ahead_tail_++;
        FormatGang::Task t;
        t.fn = decide_ahead;
        t.ctx = &a;
        gang_->post(t);
        gang_busy_ = true;
        return true;
    }
    // has the oldest batch that was posted been decided?  (advance_tiles() for it will not have to wait)
    bool ahead_ready() const
    {
        const Ahead &a = ahead_[ahead_head_ % kAheadSlots];
        return a.posted && a.n.load(std::memory_order_acquire) >= 0;
    }
    void set_ahead_min_records(size_t n) { ahead_min_records_ = n; }
    void set_arena_chunk(size_t n) { arena_.chunk = n; } // (before the first batch)
    uint64_t ahead_taken() const { return ahead_taken_; } // frames whose decision was made ahead and taken over as it was

private:
    // The decisions [j, n) of a batch decided ahead, under the true calls (air.c:94-99: which decision is the first of a new
    // call, where the last executed call ends).  Reads only the packed offsets G -- the decisions themselves stay in the cache
    // of the thread that made them and will write the frames; their ts is off by a constant, which the caller works out.
    // Enters and leaves like decide_calls (Walk::in_call; tsb is not kept here); `j` ends at the first decision that is not
    // taken over (a call that has not fired yet, or will not: the stream's last executed call ends before it).
    static void adopt_calls(Walk &w, const uint32_t *G, size_t &j, size_t n)
    {
        uint64_t base = w.base;
        int64_t idx = w.idx, lim = w.lim;
        size_t i = j;
        bool in_call = w.in_call;
        for (;;) {
            if (!in_call) {
                const uint64_t fire = base + ADSB_APBUFFSZ + (base & 1);
                if (fire > w.power_samples)
                    break;
                const uint64_t limit = fire - ADSB_DECOFFSET;
                if (limit > w.g_complete)
                    break;
                idx = (int64_t)(base - w.g_base);
                lim = (int64_t)(limit - w.g_base);
                in_call = true;
            }
            // the decisions of this call: all those below the limit -- the chain is theirs already; only the last one's end matters
            const size_t i0 = i;
            while (i < n && (int64_t)(G[i] & 0x7FFFFFFFu) < lim)
                i++;
            if (i > i0)
                idx = (int64_t)(G[i - 1] & 0x7FFFFFFFu) + (G[i - 1] >> 31 ? 1200 : 640);
            if (i == n && idx < lim)
                break; // (no decision left, the call not over)
            if (idx < lim)
                idx = lim;
            base = w.g_base + (uint64_t)idx;
            in_call = false;
        }
        w.base = base, w.idx = idx, w.lim = lim, w.in_call = in_call;
        j = i;
    }

    // (single_limit != 0: chain mode -- ONE call with that limit, as in run_calls_tiles; adopt_calls then finds no further call
    // to open: power_samples is 0 there)
    void run_calls_tiles_ahead(uint64_t power_samples, uint64_t g_complete, uint64_t single_limit)
    {
        Batch &b = batch_;
        Ahead &a = *ahead_now_;
        for (uint32_t spins = 1; a.n.load(std::memory_order_acquire) < 0; spins++)
            if (!gang_->help()) // (the batch may still be waiting for a thread: then this one decides it, or writes frames meanwhile)
                FormatGang::relax(spins);
        const size_t n = (size_t)a.n.load(std::memory_order_relaxed);
        Walk w;
        w.recs = b.recs, w.starts = b.starts, w.counts = b.counts;
        w.g_base = b.g_base, w.power_samples = power_samples, w.g_complete = g_complete, w.single_limit = single_limit;
        w.base = base_, w.tsb = b.g_base + 1 - skipped_;
        w.in_call = false;
        uint32_t u = b.u;
        const uint32_t *cur = b.cur, *tile_end = b.cur ? b.cur + 8 * (size_t)b.left : nullptr;
        auto next_tile = [&] {
            cur = nullptr;
            for (u++; u < b.u_end; u++)
                if (b.counts[u]) {
                    cur = b.recs + (size_t)b.starts[u] * 4;
                    tile_end = cur + 8 * (size_t)b.counts[u];
                    break;
                }
        };
        auto park = [&] { // the cursor and the resolver's own state, as run_calls_tiles leaves them
            b.cur = cur;
            b.left = cur ? (uint32_t)((tile_end - cur) / 8) : 0;
            b.copies = cur ? rec_copies(cur) : 1;
            b.u = u;
            b.sub = 0;
            skipped_ = b.g_base + 1 - w.tsb;
            base_ = w.base;
        };
        while (cur) { // a tile for real ...
            Decision *const dc = arena_.alloc((size_t)(tile_end - cur) / 8 + 1);
            w.cur = cur, w.tile_end = tile_end, w.u = u, w.u_end = u + 1;
            w.o = reinterpret_cast<uint64_t *>(dc);
            w.stop_when_empty = true;
            decide_calls(w);
            const size_t made = (size_t)(reinterpret_cast<Decision *>(w.o) - dc);
            arena_.shrink_to(dc + (emit(dc, made) ? made : 0));
            if (w.cur && w.cur != w.tile_end) { // a call that has not fired: the rest waits (keep_leftovers)
                cur = w.cur;
                park();
                return;
            }
            next_tile();
            if (!w.in_call) { // the tile's last record ended a call, and the next one has not fired
                park();
                return;
            }
            if (!made)
                continue;
            // ... and is its last decision one of the batch's?
            const Decision last = dc[made - 1];
            size_t lo = 0, hi = n;
            while (lo < hi) {
                const size_t mid = (lo + hi) / 2;
                if ((a.g[mid] & 0x7FFFFFFFu) < last.g_rel) // (the packed offsets: adopt_calls reads those lines anyway)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo == n || (a.g[lo] & 0x7FFFFFFFu) != last.g_rel || a.d[lo].where != last.where)
                continue;
            size_t j = lo + 1;
            adopt_calls(w, a.g, j, n);
            if (j > lo + 1) {
                // ts = g_rel + tsb on both sides, and tsb moves by the same spans from here on: the difference is a constant
                const uint64_t ts_add = last.ts - a.d[lo].ts; // (the same decision on both sides: the true ts count minus the batch's own)
                emit(a.d + lo + 1, j - (lo + 1), ts_add, a.g + lo + 1); // (the batch's decisions stay where they are: the arena is pinned)
                ahead_taken_ += j - (lo + 1);
                // the ts count behind the last decision taken over: its own ts, its offset, its span
                const Decision &e = a.d[j - 1];
                w.tsb = e.ts + ts_add - e.g_rel - ((e.where & kDecLong ? 1200u : 640u) - 1);
            }
            if (j == n) { // every record of the batch is behind the position
                cur = nullptr;
                u = b.u_end;
                break;
            }
            // the first decision that was not taken over: the cursor goes back to its record (the calls end before it)
            const uint32_t at = (a.d[j].where & ~15u) >> 4; // granule
            for (u = b.u_end; u-- > a.t0;)
                if (b.counts[u] && at >= b.starts[u] && at < b.starts[u] + 2 * b.counts[u])
                    break;
            cur = b.recs + (size_t)at * 4;
            tile_end = b.recs + ((size_t)b.starts[u] + 2 * (size_t)b.counts[u]) * 4;
            park();
            return;
        }
        // no record left: the calls that have fired run out over the rest of the batch's offsets
        w.cur = w.tile_end = nullptr, w.u = w.u_end = b.u_end;
        Decision spare[2];
        w.o = reinterpret_cast<uint64_t *>(spare);
        w.stop_when_empty = false;
        decide_calls(w);
        park();
    }

    void note_accepted(uint64_t g, uint32_t span)
    {
        if (w_on_ && !w_stop_)
            w_acc_.emplace_back(g, g + span);
        if (log_on_) {
            if (ext_n_ < ext_cap_ && log_.empty())
                ext_[ext_n_++] = LogEntry{g, span, 0};
            else
                log_.emplace_back(g, span);
        }
    }

    // One deqframe(ampbuff, len) call: visits offsets from base_ while < limit.
    void run_call(uint64_t limit)
    {
        if (tiles_fast_path()) {
            run_calls_tiles(0, 0, limit);
            return;
        }
        uint64_t idx = base_;
        while (idx < limit) {
            // next candidate at or after idx: the queue first, then the in-place batch
            while (chead_ < cands_.size() && cands_[chead_].g < idx)
                chead_++; // inside an accepted frame: never evaluated
            const bool from_queue = chead_ < cands_.size();
            uint64_t g = ~0ull;
            if (from_queue) {
                g = cands_[chead_].g;
            } else {
                batch_.skip_below(idx);
                if (batch_.cur)
                    g = batch_.g();
            }
            if (g >= limit) { // also: no candidate left
                count_tries(idx, limit - 1);
                idx = limit; // all remaining offsets advance by one (demod.c:141)
                break;
            }
            count_tries(idx, g);
            if (gang_busy_ && !out_.fits(1))
                sync(); // (the array is about to move under the gang's hands)
            adsb_frame &f = out_.push();
            std::memset(reinterpret_cast<char *>(&f) + 32, 0, 8); // (the struct's tail padding: frames are compared and copied as bytes)
            f.g = g;
            f.ts = g + 1 - skipped_; // demod.c:99: one ts++ per visited offset
            if (from_queue) {
                const adsb_candidate &c = cands_[chead_++];
                f.pw = c.pw;
                f.len = c.len;
                std::memcpy(f.frame, c.frame, 14);
                f.reserved = c.reserved;
            } else { // straight from the device record {g_rel, pw, frame[14] | len << 16 | flags << 24}
                const uint32_t *r = batch_.cur;
                f.pw = batch_.pw();
                batch_.next();
                std::memcpy(f.frame, &r[2], 14);
                f.len = (uint8_t)((r[5] >> 16) & 0xFF);
                f.reserved = (uint8_t)((r[5] >> 24) & 1u);
            }
            const uint64_t span = 80 + 80 * (uint64_t)f.len; // demod.c:109,120,123: lidx
            stats_.ok[df_slot(f.frame[0])]++;
            stats_.fixed += f.reserved & 1u;
            skipped_ += span - 1;
            if (w_on_ && !w_stop_)
                w_acc_.emplace_back(g, g + span);
            if (log_on_) {
                if (ext_n_ < ext_cap_ && log_.empty())
                    ext_[ext_n_++] = LogEntry{g, (uint32_t)span, 0};
                else
                    log_.emplace_back(g, (uint32_t)span);
            }
            idx = g + span; // demod.c:128,134
        }
        base_ = idx; // deqframe's return value; air.c:96-98 carries the rest
    }

    bool w_on_ = false, w_final_ = false, w_stop_ = false; // the call walk beside the chain (start_walk)
    uint64_t w_base_ = 0, w_end_ = 0, w_mref_ = 0, w_last_end_ = 0;
    uint64_t *w_bases_ = nullptr;
    size_t w_cap_ = 0, w_n_ = 0;
    std::vector<std::pair<uint64_t, uint64_t>> w_acc_; // (g, end) of accepted frames the walk has not passed yet
    size_t w_acc_head_ = 0;
    bool chain_ = false;   // chain mode (start_chain)
    uint64_t head_end_ = 0;
    std::vector<adsb_candidate> *head_ = nullptr;
    uint64_t base_ = 0;    // global index of ampbuff[0] at the next deqframe call
    uint64_t skipped_ = 0; // offsets jumped over by accepted frames
    std::vector<adsb_candidate> cands_;
    size_t chead_ = 0;
    Batch batch_;
    bool log_on_ = false;
    std::vector<std::pair<uint64_t, uint32_t>> log_;
    LogEntry *ext_ = nullptr;
    size_t ext_cap_ = 0, ext_n_ = 0;
    std::vector<uint64_t> tries_;
    size_t thead_ = 0;
    FrameVec out_;
    size_t ohead_ = 0;
    adsb_stats stats_{};
    FormatGang *gang_ = nullptr;
    size_t gang_min_frames_ = kGangMinFrames;
    bool gang_busy_ = false;     // tasks posted since the last sync()
    uint64_t gang_frames_ = 0;   // ... and the frames they stand for
    // The decisions of the tasks in flight: chunks that never move (a task holds a pointer into them).
    class DecArena {
    public:
        Decision *alloc(size_t n)
        {
            while (cur_ < chunks_.size() && chunks_[cur_].second - pos_ < n)
                cur_++, pos_ = 0;
            if (cur_ == chunks_.size()) {
                const size_t sz = std::max(n, chunk);
                chunks_.emplace_back(std::unique_ptr<Decision[]>(new Decision[sz]), sz);
            }
            Decision *p = chunks_[cur_].first.get() + pos_;
            pos_ += n;
            return p;
        }
        void shrink_to(Decision *end) { pos_ = (size_t)(end - chunks_[cur_].first.get()); } // (of the latest alloc)
        void reset() { cur_ = 0, pos_ = 0; }
        size_t chunk = 1u << 16; // decisions per chunk (tests make it small: every path across a chunk's end)

    private:
        std::vector<std::pair<std::unique_ptr<Decision[]>, size_t>> chunks_;
        size_t cur_ = 0, pos_ = 0;
    };
    DecArena arena_;
    int arena_pins_ = 0;          // batches decided ahead whose decisions are still needed: no reset
    static constexpr unsigned kAheadSlots = 16;
    Ahead ahead_[kAheadSlots];    // the batch being taken over, and the ones being decided behind it (posted in order: a ring)
    unsigned ahead_head_ = 0, ahead_tail_ = 0;
    Ahead *ahead_now_ = nullptr;  // advance_tiles: this batch was decided ahead
    size_t ahead_min_records_ = kAheadMinRecords;
    uint64_t ahead_taken_ = 0;
};

} // namespace adsb
