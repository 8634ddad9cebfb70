// hostpath_bench.cpp -- CPU-only timing of the host side of the streaming hand-off (handoff.hpp's collect_alone: marker
// check + tile bookkeeping; resolver.hpp's advance_tiles: the greedy replay) on two synthetic launches shaped like the
// bench's: sparse (2 786 tiles, ~17 k records, 13 k frames) and dense10 (BASELINE configs[2] at its stated density: ~110
// records per tile, 311 k records, 106 k frames).  The stream is complete before the host starts (what the host sees when it
// is the bottleneck) and lies in ordinary cached memory (the real one is device-written: every line is a miss the first time).
//   g++ -O2 -std=c++17 tools/hostpath_bench.cpp -o tools/bin/hostpath_bench && tools/bin/hostpath_bench
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#ifndef HP_HDR
#define HP_HDR "../adsbdec_amd/csrc/"
#endif
#define HP_STR2(x) #x
#include "../adsbdec_amd/csrc/handoff.hpp"
#include "../adsbdec_amd/csrc/resolver.hpp"
using namespace adsb;
#ifdef HP_R4 // built against round 4's headers (-I a checkout of them): no records of copies there
#define HP_COPIES_SHIFT 25
#else
#define HP_COPIES_SHIFT kRecCopiesShift
#endif
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void run(const char *name, uint32_t ntiles, uint32_t per, uint32_t frame_gap, int copies, bool collapse, bool two_threads = false, bool flush_lines = false, int gang_helpers = 0, bool ahead = false)
{
    const uint32_t gen = 0x1234567u;
    std::mt19937 rng(1);
    std::vector<uint32_t> hv;
    uint64_t nrec = 0;
    uint64_t next_frame = 500;
    for (uint32_t t = 0; t < ntiles; t++) {
        std::vector<uint32_t> gs, cp;
        while (next_frame < (uint64_t)(t + 1) * per) {
            int nc = 0;
            for (int c = 0; c < copies; c++)
                if (next_frame + c < (uint64_t)(t + 1) * per)
                    nc++;
            if (collapse) { // one record per run of copies (what the kernel writes since round 5)
                gs.push_back((uint32_t)next_frame);
                cp.push_back((uint32_t)nc);
            } else {
                for (int c = 0; c < nc; c++) {
                    gs.push_back((uint32_t)(next_frame + c));
                    cp.push_back(1u);
                }
            }
            // sparse: frames at random distances.  dense: back to back, and one record in seven lies INSIDE the frame before it
            // (a second transmitter's frame that the greedy rule never visits: 123 k records for 106 k frames in BASELINE
            // configs[2]'s capture) -- at random, which is what the resolver's branch predictor sees
            next_frame += frame_gap > 2000 ? frame_gap + rng() % frame_gap : rng() % 7 ? frame_gap : 150 + rng() % 900;
        }
        const uint32_t n = (uint32_t)gs.size(), reserve = stream_granules(n);
        uint32_t a[4] = {0, 0, 0, 0}, sum = 0;
        const size_t mpos = hv.size();
        hv.insert(hv.end(), 4 * reserve, 0u);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t r[8] = {gs[i], 1000 + gs[i] % 77, (17u << 3) | (uint32_t)(rng() << 8), (uint32_t)rng(), (uint32_t)rng(),
                             ((uint32_t)rng() & 0xFFFFu) | (14u << 16) | ((cp[i] - 1u) << HP_COPIES_SHIFT), cp[i] > 1 ? 1001 + gs[i] % 77 : 0,
                             cp[i] > 2 ? 1002 + gs[i] % 77 : 0};
            for (int k = 0; k < 8; k++) a[k & 3] ^= r[k];
            sum += record_term(i, r[0], r[1]);
            std::memcpy(hv.data() + mpos + 4 + 8 * i, r, 32);
            nrec++;
        }
        const uint32_t nf = n | ((reserve >> 2) << kMarkLinesShift);
        uint32_t lo, hi;
        marker_check(t, nf, gen, a[0], a[1], a[2], a[3], sum, lo, hi);
        uint32_t m[4] = {t, nf, lo, hi};
        std::memcpy(hv.data() + mpos, m, 16);
    }
    std::vector<__m128i> store(hv.size() / 4 + 1);
    uint32_t *hand = reinterpret_cast<uint32_t *>(store.data());
    std::memcpy(hand, hv.data(), hv.size() * 4);
    if (gang_helpers)
        printf(ahead ? "[%d threads decide each batch ahead and write the frames, the caller takes the decisions over] " : "[the caller decides, %d threads write the frames] ", gang_helpers);
    printf("%s%s%s: %u tiles, %llu records, %zu KiB stream\n", name, two_threads ? " [reader thread + resolver]" : " [one thread]",
           flush_lines ? " [stream flushed from the caches before every pass]" : "", ntiles, (unsigned long long)nrec, hv.size() * 4 / 1024);
    Resolver res;
    FormatGang gang;
    if (gang_helpers && gang.start(gang_helpers))
        res.set_gang(&gang);
#ifdef HP_LOG // a statistics run: every accepted frame is also logged for the device's count pass (Resolver::log_into)
    std::vector<Resolver::LogEntry> logbuf(400000);
#endif
    std::vector<uint32_t> t_start(ntiles), t_count(ntiles);
    StreamReader rd;
    if (two_threads)
        rd.start();
    for (int rep = 0; rep < 6; rep++) {
        res.reset();
#ifdef HP_LOG
        res.log_accepted(true);
        res.log_into(logbuf.data(), logbuf.size());
#endif
        std::fill(t_count.begin(), t_count.end(), ~0u);
        if (flush_lines) { // what the device's writes leave behind: no line of the stream in any cache
            for (size_t b = 0; b < hv.size() * 4; b += 64)
                _mm_clflush(reinterpret_cast<const char *>(hand) + b);
            _mm_mfence();
        }
        HandJob job;
        job.hand = hand, job.ntiles = ntiles, job.gen = gen, job.cap = (uint32_t)(hv.size() / 4);
        double tr = 0, wait_ms = 0;
        auto tl = HandCursor::clk::now();
        uint32_t delivered = 0;
        const double t0 = now();
        uint32_t held[12][2]; // (the streaming collect's own policy: decoder.hip slot_collect_streaming)
        int n_held = 0;
        auto adv = [&](uint32_t from, uint32_t upto) {
            res.advance_tiles(hand, t_start.data(), t_count.data(), from, upto, 0, (uint64_t)ntiles * per + 100000, (uint64_t)upto * per);
        };
        auto deliver_held = [&](int keep, bool only_ready) {
            int k = 0;
            for (; n_held - k > keep && (!only_ready || res.ahead_ready()); k++)
                adv(held[k][0], held[k][1]);
            for (int i = k; i < n_held; i++)
                held[i - k][0] = held[i][0], held[i - k][1] = held[i][1];
            n_held -= k;
        };
        auto flush = [&](uint32_t upto) {
            const double ta = now();
            if (ahead) {
                for (uint32_t from = delivered; from < upto;) {
                    const uint32_t to = std::min(upto, from + 64);
                    if (n_held == 12)
                        deliver_held(11, false);
                    if (res.speculate_tiles(hand, t_start.data(), t_count.data(), from, to, 0)) {
                        held[n_held][0] = from, held[n_held][1] = to;
                        n_held++;
                    } else {
                        deliver_held(0, false);
                        adv(from, to);
                    }
                    from = to;
                }
                deliver_held(0, true);
            } else {
                adv(delivered, upto);
            }
            delivered = upto;
            tr += now() - ta;
        };
        const CollectEnd end = two_threads ? collect_behind_reader(rd, job, t_start.data(), t_count.data(), delivered, flush, wait_ms, tl)
                                           : collect_alone(job, t_start.data(), t_count.data(), delivered, flush, wait_ms, tl);
        if (n_held) {
            const double ta = now();
            deliver_held(0, false);
            tr += now() - ta;
        }
        res.sync(); // (the frames are whole)
        const double t1 = now();
        const adsb_frame *fp;
        const size_t nf = res.take(&fp);
        printf("  rep %d: status %d, check + bookkeeping %.1f us, resolve %.1f us, total %.1f us = %.1f ns per record, %zu frames\n", rep, end.status,
               t1 - t0 - tr, tr, t1 - t0, (t1 - t0) * 1e3 / nrec, nf);
        if (ahead && rep == 5)
            printf("  (%llu frames taken over from batches decided ahead, all passes)\n", (unsigned long long)res.ahead_taken());
    }
    rd.stop();
    res.set_gang(nullptr);
}

int main()
{
    run("sparse", 2786, 48188, 10000, 1, false);
    run("dense10, one record per candidate (round 4)", 2786, 48188, 1200, 3, false);
#ifndef HP_R4
    run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true);
    run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, false, true);
    run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, true, false);
    run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, true, true);
    run("sparse", 2786, 48188, 10000, 1, false, false, true);
    for (int h = 1; h <= 4; h++) {
        run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, false, false, h);
        run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, false, true, h);
    }
    run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, true, true, 3);
    for (int h = 2; h <= 5; h++) {
        run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, false, true, h, true);
        run("dense10, one record per run of copies", 2786, 48188, 1200, 3, true, true, true, h, true);
    }
    run("sparse", 2786, 48188, 10000, 1, false, false, true, 4, true);
#endif
    return 0;
}
