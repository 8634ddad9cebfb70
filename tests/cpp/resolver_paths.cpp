// resolver_paths.cpp -- the three ways records reach adsb::Resolver must agree:
//   (a) adsb_candidate queue   feed() + advance()          (pinned against the oracle and the
//                                                           real deqframe by tests/test_host_logic.py)
//   (b) device records through an index list, in place      advance_device()
//   (c) tile ranges of a hand-off stream, in place          advance_tiles()
//   (d) the same, decided here and written by a gang of     set_gang() + advance_tiles()
//       three threads (gang.hpp), drained every few batches  [also under ThreadSanitizer: tests/test_sanitizers.py]
//   (e) the same, every batch decided AHEAD by the gang     speculate_tiles() + advance_tiles()
//       while the batch before it is taken over
// Random candidate sets with the shifted copies, overlaps and chains the device really
// emits, fed in random batch sizes; frames (g, ts, pw, len, bytes, flag) and Ok counters
// must be identical.  Built and run by tests/test_host_logic.py (g++, no GPU).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../adsbdec_amd/csrc/resolver.hpp"

struct Rec { // a candidate -- or, with copies > 1, the same frame at that many consecutive offsets (one stream record)
    uint64_t g;
    uint32_t pw, w[4];
    uint32_t copies = 1, pw2 = 0, pw3 = 0;
};

static std::vector<adsb_frame> drain_all(adsb::Resolver &r)
{
    std::vector<adsb_frame> out(r.pending());
    if (!out.empty())
        r.drain(out.data(), out.size());
    return out;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    std::mt19937_64 rng(12345);
    adsb::FormatGang gang;
    if (!gang.start(3)) {
        printf("no threads\n");
        return 2;
    }
    // stop() + start() with more tasks than the ring has slots on either side (round 5's advisor finding: start() used to
    // reset the slots and not the counters, and the first post() behind a restart past 256 tasks never returned)
    {
        std::atomic<uint64_t> ran{0};
        adsb::FormatGang::Task t;
        t.fn = [](const adsb::FormatGang::Task &tk, adsb::FormatGang::Counts &) { static_cast<std::atomic<uint64_t> *>(tk.ctx)->fetch_add(1); };
        t.ctx = &ran;
        for (int leg = 0; leg < 3; leg++) {
            for (int i = 0; i < 300 + 17 * leg; i++)
                gang.post(t);
            gang.wait_all();
            gang.end();
            gang.stop();
            if (!gang.start(leg == 1 ? 2 : 3)) {
                printf("no threads after a restart\n");
                return 2;
            }
        }
        if (ran.load() != 300 + 317 + 334) {
            printf("gang restart: %llu tasks ran, %d were posted\n", (unsigned long long)ran.load(), 300 + 317 + 334);
            return 1;
        }
    }
    uint64_t taken = 0, frames_e = 0, taken_chain = 0, frames_g = 0;
    int chain_rounds = 0;
    for (int round = 0; round < rounds; round++) {
        if (round && round % 37 == 0) { // ... and between rounds of real work (nothing is in flight here)
            gang.stop();
            if (!gang.start(3)) {
                printf("no threads after a restart\n");
                return 2;
            }
        }
        // a stream of `total` offsets; candidates in clusters (a frame + shifted copies + overlapping others)
        const uint64_t total = 200000 + rng() % 2000000;
        std::vector<Rec> recs;
        for (uint64_t g = rng() % 5000; g + 1300 < total;) {
            const int copies = 1 + (int)(rng() % 4);
            for (int c = 0; c < copies; c++) {
                Rec r;
                r.g = g + (uint64_t)c * (1 + rng() % 3);
                r.pw = (uint32_t)(rng() % 100000);
                const bool is_short = rng() % 4 == 0;
                for (auto &w : r.w)
                    w = (uint32_t)rng();
                r.w[0] = (r.w[0] & ~0xFFu) | (is_short ? (11u << 3) : ((rng() & 1) ? (17u << 3) : (18u << 3))) | (rng() & 7u);
                r.w[3] = (r.w[3] & 0xFFFFu) | ((is_short ? 7u : 14u) << 16) | ((uint32_t)(rng() & 1) << 24);
                if (is_short)
                    r.w[1] &= 0x00FFFFFFu, r.w[2] = 0, r.w[3] &= 0xFFFF0000u;
                if (rng() % 3 == 0) { // the half-sample shifted copies of one frame, as one record (scan_kernel_format.h)
                    r.copies = 2 + (uint32_t)(rng() % 2);
                    r.pw2 = (uint32_t)(rng() % 100000), r.pw3 = (uint32_t)(rng() % 100000);
                }
                if (recs.empty() || r.g > recs.back().g + recs.back().copies - 1)
                    recs.push_back(r);
            }
            g += (rng() % 3 == 0) ? 200 + rng() % 1200 : 1500 + rng() % 60000; // some overlap chains
        }
        const std::vector<Rec> collapsed = recs; // (c), (d), (e) see these; (a), (b) one candidate per offset
        {
            std::vector<Rec> expanded;
            for (const Rec &r : collapsed)
                for (uint32_t k = 0; k < r.copies; k++) {
                    Rec e = r;
                    e.g = r.g + k, e.pw = k == 0 ? r.pw : k == 1 ? r.pw2 : r.pw3, e.copies = 1;
                    expanded.push_back(e);
                }
            recs = expanded;
        }
        const uint64_t power_samples = total + 1195 + (rng() % 2) * 2; // even
        // (a) queue path
        adsb::Resolver ra, rb, rc, rd, re;
        ra.reset(), rb.reset(), rc.reset(), rd.reset(), re.reset();
        re.set_gang(&gang, round % 2 ? 16 : 1);
        re.set_ahead_min_records(round % 5 == 0 ? 40 : 1);
        rc.log_accepted(true), re.log_accepted(true); // a statistics run's log of accepted frames: the same entries either way
        std::vector<adsb::Resolver::LogEntry> lc(round % 4 == 0 ? 50 : 1 << 16), le(lc.size()); // (the caller's buffer; what does not fit: accepted_log())
        rc.log_into(lc.data(), lc.size()), re.log_into(le.data(), le.size());
        if (round % 2) // decisions in chunks of 48: allocations cross chunk ends, and the arena starts over while tasks are in flight
            rd.set_arena_chunk(48), re.set_arena_chunk(48);
        rd.set_gang(&gang, round % 3 == 0 ? 64 : 1); // (with a threshold, small batches are written by the caller in between)
        std::vector<adsb_frame> fa, fb, fc, fd, fe;
        {
            size_t i = 0;
            uint64_t gc = 0;
            while (gc < total) {
                gc = std::min<uint64_t>(total, gc + 1 + rng() % 300000);
                std::vector<adsb_candidate> batch;
                for (; i < recs.size() && recs[i].g < gc; i++) {
                    adsb_candidate c;
                    std::memset(&c, 0, sizeof c);
                    c.g = recs[i].g, c.pw = recs[i].pw;
                    std::memcpy(c.frame, recs[i].w, 14);
                    c.len = (uint8_t)((recs[i].w[3] >> 16) & 0xFF);
                    c.reserved = (uint8_t)((recs[i].w[3] >> 24) & 1u);
                    batch.push_back(c);
                }
                ra.feed(batch.data(), batch.size(), nullptr, 0);
                ra.advance(power_samples, gc);
                auto f = drain_all(ra);
                fa.insert(fa.end(), f.begin(), f.end());
            }
        }
        // (b) 6-word records through an index list (shuffled storage), with a launch base
        {
            size_t i = 0;
            uint64_t gc = 0;
            while (gc < total) {
                const uint64_t base = gc - gc % 28;
                gc = std::min<uint64_t>(total, gc + 1 + rng() % 300000);
                std::vector<uint32_t> words, order;
                std::vector<size_t> idx;
                for (; i < recs.size() && recs[i].g < gc; i++)
                    idx.push_back(i);
                std::vector<size_t> slot(idx.size());
                for (size_t k = 0; k < slot.size(); k++)
                    slot[k] = k;
                std::shuffle(slot.begin(), slot.end(), rng);
                words.resize(idx.size() * 6);
                order.resize(idx.size());
                for (size_t k = 0; k < idx.size(); k++) {
                    uint32_t *w = &words[slot[k] * 6];
                    w[0] = (uint32_t)(recs[idx[k]].g - base), w[1] = recs[idx[k]].pw;
                    std::memcpy(w + 2, recs[idx[k]].w, 16);
                    order[k] = (uint32_t)slot[k];
                }
                rb.advance_device(words.data(), order.data(), order.size(), 6, 0, base, power_samples, gc);
                auto f = drain_all(rb);
                fb.insert(fb.end(), f.begin(), f.end());
            }
        }
        // (c) tile ranges of a granule stream: tiles of `per` offsets, stored in shuffled tile order
        {
            const uint32_t per = 12880 + 7056 * (uint32_t)(rng() % 4);
            const uint32_t ntiles = (uint32_t)((total + per - 1) / per);
            std::vector<std::vector<size_t>> by_tile(ntiles);
            for (size_t i = 0; i < collapsed.size(); i++)
                by_tile[collapsed[i].g / per].push_back(i);
            std::vector<uint32_t> tile_order(ntiles), starts(ntiles), counts(ntiles), stream;
            for (uint32_t t = 0; t < ntiles; t++)
                tile_order[t] = t;
            for (uint32_t t = 0; t + 1 < ntiles; t += 2) // neighbours complete out of order
                if (rng() & 1)
                    std::swap(tile_order[t], tile_order[t + 1]);
            for (uint32_t t : tile_order) {
                stream.insert(stream.end(), 4, 0xDEADBEEFu); // the marker granule
                starts[t] = (uint32_t)(stream.size() / 4);
                counts[t] = (uint32_t)by_tile[t].size();
                for (size_t i : by_tile[t]) {
                    const Rec &r = collapsed[i];
                    const uint32_t w[8] = {(uint32_t)r.g, r.pw, r.w[0], r.w[1], r.w[2], r.w[3] | (r.copies - 1) << adsb::kRecCopiesShift, r.pw2, r.pw3};
                    stream.insert(stream.end(), w, w + 8);
                }
            }
            for (uint32_t t = 0; t < ntiles;) {
                const uint32_t t1 = std::min<uint32_t>(ntiles, t + 1 + (uint32_t)(rng() % 40));
                rc.advance_tiles(stream.data(), starts.data(), counts.data(), t, t1, 0, power_samples,
                                 std::min<uint64_t>(total, (uint64_t)t1 * per));
                auto f = drain_all(rc);
                fc.insert(fc.end(), f.begin(), f.end());
                rd.advance_tiles(stream.data(), starts.data(), counts.data(), t, t1, 0, power_samples,
                                 std::min<uint64_t>(total, (uint64_t)t1 * per));
                if (rng() % 4 == 0 || t1 == ntiles) {
                    f = drain_all(rd);
                    fd.insert(fd.end(), f.begin(), f.end());
                }
                t = t1;
            }
            rd.sync(); // (the stream goes away with this block)
            // (e): the batches first, so that each can be posted one ahead
            std::vector<std::pair<uint32_t, uint32_t>> bt;
            for (uint32_t t = 0; t < ntiles;) {
                const uint32_t t1 = std::min<uint32_t>(ntiles, t + 1 + (uint32_t)(rng() % 40));
                bt.emplace_back(t, t1);
                t = t1;
            }
            const size_t depth = 1 + round % 3; // batches posted ahead of the one that is handed over
            for (size_t k = 0; k < depth && k < bt.size(); k++)
                re.speculate_tiles(stream.data(), starts.data(), counts.data(), bt[k].first, bt[k].second, 0);
            for (size_t k = 0; k < bt.size(); k++) {
                if (k + depth < bt.size())
                    re.speculate_tiles(stream.data(), starts.data(), counts.data(), bt[k + depth].first, bt[k + depth].second, 0);
                re.advance_tiles(stream.data(), starts.data(), counts.data(), bt[k].first, bt[k].second, 0, power_samples,
                                 std::min<uint64_t>(total, (uint64_t)bt[k].second * per));
                if (rng() % 4 == 0 || k + 1 == bt.size()) {
                    auto f = drain_all(re);
                    fe.insert(fe.end(), f.begin(), f.end());
                }
            }
            re.sync();
            taken += re.ahead_taken(), frames_e += fe.size();
            // (f), (g) CHAIN mode (a shard of the multi-GPU driver: the greedy rule from g_begin on, one call per advance, head
            // candidates and the walk of the deqframe calls beside it): the queue path against tiles whose batches are decided
            // AHEAD by the gang and taken over -- round 6; until then a chain refused to be decided ahead.
            {
                const uint64_t g_begin = 28 * (rng() % 2000), head_end = std::min<uint64_t>(total, g_begin + 3000 + rng() % 60000);
                const uint64_t total_samples = 2 * power_samples;
                std::vector<adsb_candidate> hf, hg;
                std::vector<uint64_t> bf(4096), bg(4096);
                adsb::Resolver rf, rg;
                rf.start_chain(g_begin, head_end, &hf);
                rf.start_walk(g_begin, total, total_samples, bf.data(), bf.size());
                rg.start_chain(g_begin, head_end, &hg);
                rg.start_walk(g_begin, total, total_samples, bg.data(), bg.size());
                rg.set_gang(&gang, round % 2 ? 8 : 1);
                rg.set_ahead_min_records(round % 5 == 0 ? 40 : 1);
                if (round % 2)
                    rg.set_arena_chunk(48);
                std::vector<adsb_frame> ff, fg;
                {   // (f) the queue
                    size_t i = 0;
                    while (i < recs.size() && recs[i].g < g_begin)
                        i++;
                    uint64_t gc = g_begin;
                    while (gc < total) {
                        gc = std::min<uint64_t>(total, gc + 1 + rng() % 300000);
                        std::vector<adsb_candidate> batch;
                        for (; i < recs.size() && recs[i].g < gc; i++) {
                            adsb_candidate c;
                            std::memset(&c, 0, sizeof c);
                            c.g = recs[i].g, c.pw = recs[i].pw;
                            std::memcpy(c.frame, recs[i].w, 14);
                            c.len = (uint8_t)((recs[i].w[3] >> 16) & 0xFF);
                            c.reserved = (uint8_t)((recs[i].w[3] >> 24) & 1u);
                            batch.push_back(c);
                        }
                        rf.feed(batch.data(), batch.size(), nullptr, 0);
                        rf.advance(0, gc);
                        auto f = drain_all(rf);
                        ff.insert(ff.end(), f.begin(), f.end());
                    }
                }
                {   // (g) tiles of a stream whose records start at g_begin (a shard only holds its own offsets), decided ahead
                    const uint32_t t_first = (uint32_t)(g_begin / per);
                    std::vector<uint32_t> cstream, cstarts(ntiles, 0), ccounts(ntiles, 0);
                    for (uint32_t t = 0; t < ntiles; t++) {
                        cstream.insert(cstream.end(), 4, 0xDEADBEEFu);
                        cstarts[t] = (uint32_t)(cstream.size() / 4);
                        for (size_t i : by_tile[t]) {
                            const Rec &r = collapsed[i];
                            if (r.g < g_begin) // (a record that straddles g_begin with its copies is cut: the copies below are not the shard's)
                                continue;
                            const uint32_t w[8] = {(uint32_t)r.g, r.pw, r.w[0], r.w[1], r.w[2], r.w[3] | (r.copies - 1) << adsb::kRecCopiesShift, r.pw2, r.pw3};
                            cstream.insert(cstream.end(), w, w + 8);
                            ccounts[t]++;
                        }
                    }
                    std::vector<std::pair<uint32_t, uint32_t>> cb;
                    for (uint32_t t = t_first; t < ntiles;) {
                        const uint32_t t1 = std::min<uint32_t>(ntiles, t + 1 + (uint32_t)(rng() % 40));
                        cb.emplace_back(t, t1);
                        t = t1;
                    }
                    const size_t cdepth = 1 + round % 3;
                    for (size_t k = 0; k < cdepth && k < cb.size(); k++)
                        rg.speculate_tiles(cstream.data(), cstarts.data(), ccounts.data(), cb[k].first, cb[k].second, 0);
                    for (size_t k = 0; k < cb.size(); k++) {
                        if (k + cdepth < cb.size())
                            rg.speculate_tiles(cstream.data(), cstarts.data(), ccounts.data(), cb[k + cdepth].first, cb[k + cdepth].second, 0);
                        rg.capture_head_tiles(cstream.data(), cstarts.data(), ccounts.data(), cb[k].first, cb[k].second, 0);
                        rg.advance_tiles(cstream.data(), cstarts.data(), ccounts.data(), cb[k].first, cb[k].second, 0, 0,
                                         std::min<uint64_t>(total, (uint64_t)cb[k].second * per));
                        if (rng() % 4 == 0 || k + 1 == cb.size()) {
                            auto f = drain_all(rg);
                            fg.insert(fg.end(), f.begin(), f.end());
                        }
                    }
                    rg.advance(0, total);
                    auto f = drain_all(rg);
                    fg.insert(fg.end(), f.begin(), f.end());
                    rg.sync();
                    taken_chain += rg.ahead_taken(), frames_g += fg.size();
                }
                // a record of copies that straddles g_begin exists in (f) as single candidates from g_begin on and in (g) not at
                // all: such rounds are skipped (the device never makes one: a shard's first tile starts at g_begin)
                bool straddle = false;
                for (const Rec &r : collapsed)
                    straddle |= r.g < g_begin && r.g + r.copies > g_begin;
                auto same_c = [](const std::vector<adsb_candidate> &x, const std::vector<adsb_candidate> &y) {
                    if (x.size() != y.size())
                        return false;
                    for (size_t i = 0; i < x.size(); i++)
                        if (x[i].g != y[i].g || x[i].pw != y[i].pw || x[i].len != y[i].len || std::memcmp(x[i].frame, y[i].frame, 14))
                            return false;
                    return true;
                };
                int fin_f = 0, fin_g = 0;
                if (!straddle) {
                    const bool walk_ok = rf.walk_bases() == rg.walk_bases() && rf.walk_final() == rg.walk_final() &&
                                         !std::memcmp(bf.data(), bg.data(), std::min(rf.walk_bases(), bf.size()) * sizeof(uint64_t));
                    (void)fin_f, (void)fin_g;
                    auto same_f = [](const std::vector<adsb_frame> &x, const std::vector<adsb_frame> &y) {
                        if (x.size() != y.size())
                            return false;
                        for (size_t i = 0; i < x.size(); i++)
                            if (x[i].g != y[i].g || x[i].ts != y[i].ts || x[i].pw != y[i].pw || x[i].len != y[i].len ||
                                std::memcmp(x[i].frame, y[i].frame, x[i].len) || x[i].reserved != y[i].reserved)
                                return false;
                        return true;
                    };
                    if (!same_f(ff, fg) || rf.skipped() != rg.skipped() || !same_c(hf, hg) || !walk_ok || rf.base() != rg.base() ||
                        std::memcmp(&rf.stats(), &rg.stats(), sizeof(adsb_stats))) {
                        printf("round %d: CHAIN MISMATCH (%zu / %zu frames, skipped %llu / %llu, head %zu / %zu, walk %zu / %zu, base %llu / %llu)\n", round,
                               ff.size(), fg.size(), (unsigned long long)rf.skipped(), (unsigned long long)rg.skipped(), hf.size(), hg.size(),
                               rf.walk_bases(), rg.walk_bases(), (unsigned long long)rf.base(), (unsigned long long)rg.base());
                        return 1;
                    }
                    chain_rounds++;
                }
            }
        }
        auto same = [](const std::vector<adsb_frame> &x, const std::vector<adsb_frame> &y) {
            if (x.size() != y.size())
                return false;
            for (size_t i = 0; i < x.size(); i++)
                if (x[i].g != y[i].g || x[i].ts != y[i].ts || x[i].pw != y[i].pw || x[i].len != y[i].len ||
                    std::memcmp(x[i].frame, y[i].frame, x[i].len) || x[i].reserved != y[i].reserved)
                    return false;
            return true;
        };
        const bool ok = same(fa, fb) && same(fa, fc) && same(fa, fd) && same(fa, fe) && !std::memcmp(&ra.stats(), &re.stats(), sizeof(adsb_stats)) && !std::memcmp(&ra.stats(), &rb.stats(), sizeof(adsb_stats)) &&
                        !std::memcmp(&ra.stats(), &rc.stats(), sizeof(adsb_stats)) && !std::memcmp(&ra.stats(), &rd.stats(), sizeof(adsb_stats));
        const bool same_log = rc.accepted_log() == re.accepted_log() && rc.logged_ext() == re.logged_ext() &&
                              rc.logged_ext() + rc.accepted_log().size() == fc.size() &&
                              !std::memcmp(lc.data(), le.data(), rc.logged_ext() * sizeof(adsb::Resolver::LogEntry));
        if (!ok || !same_log || fa.empty()) {
            printf("round %d: MISMATCH (%zu / %zu / %zu / %zu / %zu frames of %zu candidates)\n", round, fa.size(), fb.size(), fc.size(), fd.size(), fe.size(),
                   recs.size());
            return 1;
        }
    }
    if (taken * 2 < frames_e) { // (e) must really take batches over, not decide them all again
        printf("only %llu of %llu frames were taken over from batches decided ahead\n", (unsigned long long)taken, (unsigned long long)frames_e);
        return 1;
    }
    if (chain_rounds * 2 < rounds || taken_chain * 2 < frames_g) { // (g) must really take a chain's batches over
        printf("chain mode: %d of %d rounds compared, %llu of %llu frames taken over\n", chain_rounds, rounds, (unsigned long long)taken_chain,
               (unsigned long long)frames_g);
        return 1;
    }
    printf("ok: %d rounds (%llu of %llu frames of path (e) taken over from batches decided ahead; chain mode: %d rounds, %llu of %llu)\n", rounds,
           (unsigned long long)taken, (unsigned long long)frames_e, chain_rounds, (unsigned long long)taken_chain, (unsigned long long)frames_g);
    return 0;
}
