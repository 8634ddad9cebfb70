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
#pragma once

#include <cstdint>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/adsbdec_amd.h"

namespace adsb {

class Resolver {
public:
    void reset()
    {
        base_ = 0;
        skipped_ = 0;
        cands_.clear();
        tries_.clear();
        out_.clear();
        std::memset(&stats_, 0, sizeof stats_);
    }

    // Records must arrive in ascending g over the life of the stream.
    void feed(const adsb_candidate *c, size_t n, const uint64_t *tries, size_t nt)
    {
        cands_.insert(cands_.end(), c, c + n);
        tries_.insert(tries_.end(), tries, tries + nt);
    }

    // power_samples: samples the front end has produced so far (air.c `aidx` grows
    // by two per loop pass, so this is even); g_complete: every record with
    // g < g_complete has been fed.
    void advance(uint64_t power_samples, uint64_t g_complete)
    {
        for (;;) {
            // air.c:94: the test `aidx >= APBUFFSZ` is made after every second
            // power sample, so the call fires at the first EVEN total T with
            // T - base >= 40980.
            const uint64_t fire = base_ + ADSB_APBUFFSZ + (base_ & 1);
            if (fire > power_samples)
                break; // the reference has not called deqframe yet (never, at EOF)
            const uint64_t limit = fire - ADSB_DECOFFSET; // demod.c:89 `idx < len-DECOFFSET`
            if (limit > g_complete)
                break; // the device has not scanned that far yet
            run_call(limit);
        }
    }

    std::deque<adsb_frame> &out() { return out_; }
    const adsb_stats &stats() const { return stats_; }
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

    // valid.c:46,68: one Try per visited offset that passed the DF gate.
    void count_tries(uint64_t from, uint64_t to_inclusive)
    {
        while (!tries_.empty() && (tries_.front() >> 2) < from)
            tries_.pop_front(); // shadowed by an accepted frame: never visited
        while (!tries_.empty() && (tries_.front() >> 2) <= to_inclusive) {
            stats_.try_[tries_.front() & 3]++;
            tries_.pop_front();
        }
    }

    // One deqframe(ampbuff, len) call: visits offsets from base_ while < limit.
    void run_call(uint64_t limit)
    {
        uint64_t idx = base_;
        while (idx < limit) {
            while (!cands_.empty() && cands_.front().g < idx)
                cands_.pop_front(); // inside an accepted frame: never evaluated
            if (cands_.empty() || cands_.front().g >= limit) {
                count_tries(idx, limit - 1);
                idx = limit; // all remaining offsets advance by one (demod.c:141)
                break;
            }
            const adsb_candidate &c = cands_.front();
            count_tries(idx, c.g);
            const uint64_t span = 80 + 80 * (uint64_t)c.len; // demod.c:109,120,123: lidx
            adsb_frame f;
            std::memset(&f, 0, sizeof f);
            f.g = c.g;
            f.ts = c.g + 1 - skipped_; // demod.c:99: one ts++ per visited offset
            f.pw = c.pw;
            f.len = c.len;
            std::memcpy(f.frame, c.frame, c.len);
            out_.push_back(f);
            stats_.ok[df_slot(c.frame[0])]++;
            skipped_ += span - 1;
            idx = c.g + span; // demod.c:128,134
            cands_.pop_front();
        }
        base_ = idx; // deqframe's return value; air.c:96-98 carries the rest
    }

    uint64_t base_ = 0;    // global index of ampbuff[0] at the next deqframe call
    uint64_t skipped_ = 0; // offsets jumped over by accepted frames
    std::deque<adsb_candidate> cands_;
    std::deque<uint64_t> tries_;
    std::deque<adsb_frame> out_;
    adsb_stats stats_{};
};

} // namespace adsb
