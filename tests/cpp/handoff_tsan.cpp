// handoff_tsan.cpp -- the host side of the device -> host hand-off (adsbdec_amd/csrc/handoff.hpp: HandCursor, StreamReader,
// collect_alone, collect_behind_reader) and the resolver behind it (resolver.hpp advance_tiles), with a THREAD playing the
// device.  Built twice by tests/test_sanitizers.py: -fsanitize=thread, and -fsanitize=address,undefined.  No GPU.
//
// The "device" writes a launch's stream the way the kernel does, and worse: tiles reserve their ranges in a random
// completion order; the stream's memory starts out holding a valid stream of ANOTHER launch (other gen: stale bytes that
// look right); every range is first scribbled over, then its granules land one by one in random order with random pauses,
// the marker first, last or in between; some tiles reserve more lines than they keep records for; some launches overflow
// the stream, flag a tile "finish after completion", publish a tile twice, or never publish one at all.
// The consumers must deliver exactly the records that were written, tile by tile in ascending order, report the right end
// status, and -- what the sanitizers are here for -- do so without a data race or a bad access in their OWN
// synchronisation.  (The device's stores and the cursor's polling loads are the one deliberate exception: see
// HandCursor::tile_in.)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../adsbdec_amd/csrc/handoff.hpp"
#include "../../adsbdec_amd/csrc/resolver.hpp"

using namespace adsb;

static constexpr uint32_t kTileOffsets = 4096; // offsets a tile owns in this model

struct Launch {
    uint32_t ntiles = 0, gen = 0, cap = 0;
    std::vector<uint32_t> n;        // records per tile
    std::vector<uint32_t> reserve;  // granules the tile reserves (>= stream_granules(n))
    std::vector<uint32_t> flags;    // kMarkOver on some
    std::vector<uint32_t> order;    // completion order
    int fault = 0;                  // 0 none; 1 a tile twice; 2 a tile never published
    uint32_t fault_tile = 0;
};

static uint32_t mix32(uint32_t h)
{
    h ^= h >> 15, h *= 2246822519u, h ^= h >> 13, h *= 3266489917u, h ^= h >> 16;
    return h;
}

// A record as the kernel writes it: {g_rel, pw, frame bytes 0..7} {frame bytes 8..13 | len << 16, 0, 0}.  Offsets and
// powers are REGULAR on purpose (the same values come back launch after launch at the same places: stale bytes that
// differ from the new ones in g_rel / pw only, and alike -- what the marker's rank-weighted sum is for); the frame
// bytes are what real frames are to a checksum: arbitrary, and different from launch to launch.
static void record_words(uint32_t gen, uint32_t tile, uint32_t i, uint32_t out[8])
{
    const uint32_t g_rel = tile * kTileOffsets + 7 + 600 * i; // ascending inside a tile and from tile to tile; frames of 640 / 1200 overlap now and then
    const bool lng = ((tile + i) % 3) != 0;
    const uint32_t h = mix32(gen ^ mix32(tile * 8191u + i));
    out[0] = g_rel;
    out[1] = 1000 + tile % 977 + i;
    out[2] = (lng ? 0x8Du : 0x5Du) | (mix32(h + 1) << 8);
    out[3] = mix32(h + 2);
    out[4] = mix32(h + 3);
    out[5] = (mix32(h + 4) & 0xFFFFu) | ((lng ? 14u : 7u) << 16);
    out[6] = out[7] = 0;
}

// the device's stores: deliberately outside the race detector's view, like the real ones
__attribute__((no_sanitize("thread"))) static void dev_store(uint32_t *hand, uint32_t gran, const uint32_t w[4])
{
    _mm_store_si128(reinterpret_cast<__m128i *>(hand) + gran, _mm_set_epi32((int)w[3], (int)w[2], (int)w[1], (int)w[0]));
}

static void write_tile(uint32_t *hand, const Launch &L, uint32_t tile, uint32_t base, std::mt19937 &rng, bool torn)
{
    const uint32_t n = L.n[tile];
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> gr; // (granule index, words)
    uint32_t a[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < n; i++) {
        uint32_t w[8];
        record_words(L.gen, tile, i, w);
        gr.push_back({base + 1 + 2 * i, {w[0], w[1], w[2], w[3]}});
        gr.push_back({base + 2 + 2 * i, {w[4], w[5], w[6], w[7]}});
        for (int k = 0; k < 4; k++)
            a[k] ^= w[k] ^ w[4 + k];
    }
    const uint32_t nf = n | L.flags[tile] | ((L.reserve[tile] >> 2) << kMarkLinesShift);
    uint32_t lo, hi, sum = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t w[8];
        record_words(L.gen, tile, i, w);
        sum += record_term(i, w[0], w[1]);
    }
    marker_check(tile, nf, L.gen, a[0], a[1], a[2], a[3], sum, lo, hi);
    gr.push_back({base, {tile, nf, lo, hi}});
    if (torn) {
        const uint32_t junk[4] = {tile, nf, lo ^ 1u, hi}; // a marker that is ALMOST right, over stale records
        dev_store(hand, base, junk);
        std::shuffle(gr.begin(), gr.end(), rng);
    }
    for (auto &g : gr) {
        dev_store(hand, g.first, g.second.data());
        if (torn && rng() % 4 == 0)
            for (volatile int spin = 0; spin < (int)(rng() % 2000); spin++) {
            }
    }
}

struct Device {
    std::thread th;
    std::atomic<int> done{0};
    static int query(void *ctx) { return static_cast<Device *>(ctx)->done.load(std::memory_order_acquire); }
    void run(uint32_t *hand, const Launch &L, uint32_t seed)
    {
        th = std::thread([=, &L] {
            std::mt19937 rng(seed);
            uint32_t next = 0;
            for (uint32_t k = 0; k < L.ntiles; k++) {
                const uint32_t t = L.order[k];
                if (L.fault == 2 && t == L.fault_tile)
                    continue; // reserves nothing, writes nothing: the host must notice once the launch has ended
                const uint32_t base = next;
                next += L.reserve[t];
                if (base + L.reserve[t] <= L.cap)
                    write_tile(hand, L, t, base, rng, true);
                else if (base < L.cap) { // does not fit: says so in its marker, records would be "loose"
                    const uint32_t nf = L.n[t] | kMarkNoFit | ((L.reserve[t] >> 2) << kMarkLinesShift);
                    uint32_t lo, hi;
                    marker_check(t, nf, L.gen, 0, 0, 0, 0, 0, lo, hi);
                    const uint32_t w[4] = {t, nf, lo, hi};
                    dev_store(hand, base, w);
                }
                if (L.fault == 1 && t == L.fault_tile && next + L.reserve[t] <= L.cap) { // the same tile again
                    write_tile(hand, L, t, next, rng, false);
                    next += L.reserve[t];
                }
                if (rng() % 16 == 0)
                    std::this_thread::sleep_for(std::chrono::microseconds(rng() % 40));
            }
            done.store(1, std::memory_order_release);
        });
    }
};

static Launch make_launch(std::mt19937 &rng, int round)
{
    Launch L;
    L.ntiles = 50 + rng() % 1500;
    L.gen = (uint32_t)rng() | 1u;
    L.n.resize(L.ntiles);
    L.reserve.resize(L.ntiles);
    L.flags.assign(L.ntiles, 0);
    uint32_t total = 0;
    for (uint32_t t = 0; t < L.ntiles; t++) {
        L.n[t] = rng() % 8 == 0 ? rng() % 30 : rng() % 5;
        L.reserve[t] = stream_granules(L.n[t] + (rng() % 4 == 0 ? rng() % 9 : 0)); // sometimes reserved for more than it kept
        total += L.reserve[t];
    }
    L.cap = total + 64;
    L.order.resize(L.ntiles);
    for (uint32_t t = 0; t < L.ntiles; t++)
        L.order[t] = t;
    for (uint32_t t = 0; t + 1 < L.ntiles; t++) // completion order: ascending, locally shuffled (tiles of a resident round)
        if (rng() % 2)
            std::swap(L.order[t], L.order[std::min<uint32_t>(L.ntiles - 1, t + rng() % 24)]);
    switch (round % 8) {
    case 3: L.cap = total / 2 + 4; break;                                                // the stream overflows half-way
    case 5: L.flags[rng() % L.ntiles] = kMarkOver; break;                                // a tile asks to be finished after completion
    case 6: L.fault = 1, L.fault_tile = rng() % L.ntiles; break;                         // a tile twice
    case 7: L.fault = 2, L.fault_tile = rng() % L.ntiles; break;                         // a tile never
    default: break;
    }
    return L;
}

static int run(int rounds, StreamReader &reader);

int main(int argc, char **argv)
{
    StreamReader reader;
    reader.start();
    const int rc = run(argc > 1 ? atoi(argv[1]) : 64, reader);
    reader.stop();
    return rc;
}

static int run(int rounds, StreamReader &reader)
{
    std::mt19937 rng(20260001);
    uint64_t tiles_checked = 0, records_checked = 0, frames = 0;
    int ends[4] = {0, 0, 0, 0};
    uint32_t *hand = nullptr;
    const size_t hand_granules = 1u << 18;
    if (posix_memalign(reinterpret_cast<void **>(&hand), 64, hand_granules * 16) != 0)
        return 2;
    std::memset(hand, 0, hand_granules * 16);
    for (int round = 0; round < rounds; round++) {
        const Launch L = make_launch(rng, round);
        if (L.cap > hand_granules)
            continue;
        // (the memory still holds the previous round's stream, written with another gen: stale bytes that look like a stream)
        std::vector<uint32_t> t_start(L.ntiles, 0), t_count(L.ntiles, ~0u);
        HandJob job;
        job.hand = hand;
        job.ntiles = L.ntiles;
        job.gen = L.gen;
        job.cap = L.cap;
        Device dev;
        job.done = Device::query;
        job.ctx = &dev;
        Resolver res;
        res.reset();
        uint32_t delivered = 0;
        bool bad = false;
        auto flush = [&](uint32_t upto) {
            for (uint32_t u = delivered; u < upto && !bad; u++) {
                if (t_count[u] != L.n[u]) {
                    fprintf(stderr, "round %d: tile %u has %u records, %u were written\n", round, u, t_count[u], L.n[u]);
                    bad = true;
                    break;
                }
                for (uint32_t i = 0; i < L.n[u]; i++) {
                    uint32_t w[8];
                    record_words(L.gen, u, i, w);
                    if (std::memcmp(hand + 4 * (size_t)(t_start[u] + 2 * i), w, 32) != 0) {
                        fprintf(stderr, "round %d: tile %u record %u differs from what the device wrote\n", round, u, i);
                        const uint32_t *m = hand + 4 * (size_t)(t_start[u] - 1);
                        fprintf(stderr, "  marker {%u, %#x, %#x, %#x}, n = %u, gen %#x\n", m[0], m[1], m[2], m[3], L.n[u], L.gen);
                        for (uint32_t k = 0; k < L.n[u]; k++) {
                            uint32_t e[8];
                            record_words(L.gen, u, k, e);
                            const uint32_t *h = hand + 4 * (size_t)(t_start[u] + 2 * k);
                            fprintf(stderr, "  rec %u: have %08x %08x %08x %08x | %08x %08x %08x %08x\n          want %08x %08x %08x %08x | %08x %08x %08x %08x\n", k,
                                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
                        }
                        bad = true;
                        break;
                    }
                    records_checked++;
                }
                tiles_checked++;
            }
            // ... and through the resolver, in place, as the decoder does
            res.advance_tiles(hand, t_start.data(), t_count.data(), delivered, upto, 0, (uint64_t)L.ntiles * kTileOffsets + 100000,
                              (uint64_t)upto * kTileOffsets);
            delivered = upto;
        };
        dev.run(hand, L, (uint32_t)rng());
        double wait_ms = 0;
        auto t_last = HandCursor::clk::now();
        const bool two = round % 2 == 1;
        const CollectEnd end = two ? collect_behind_reader(reader, job, t_start.data(), t_count.data(), delivered, flush, wait_ms, t_last)
                                   : collect_alone(job, t_start.data(), t_count.data(), delivered, flush, wait_ms, t_last);
        dev.th.join();
        if (bad)
            return 1;
        int want = 0;
        if (L.fault == 1)
            want = -1;
        else if (L.fault == 2)
            want = -2;
        else if (round % 8 == 3 || round % 8 == 5)
            want = 1;
        // (a faulty launch may also end on an earlier overflow-free condition only: the fault is what ends it)
        if (end.status != want) {
            fprintf(stderr, "round %d (%s): collect ended with %d, expected %d (pos %u tile %u, delivered %u of %u)\n", round,
                    two ? "reader thread" : "alone", end.status, want, end.pos, end.tile, delivered, L.ntiles);
            return 1;
        }
        if (want == 0 && delivered != L.ntiles) {
            fprintf(stderr, "round %d: %u of %u tiles delivered\n", round, delivered, L.ntiles);
            return 1;
        }
        if (want == -2 && delivered > L.fault_tile) {
            fprintf(stderr, "round %d: tiles beyond the one that never came were delivered\n", round);
            return 1;
        }
        ends[want == 0 ? 0 : want == 1 ? 1 : want == -1 ? 2 : 3]++;
        frames += res.pending();
    }
    free(hand);
    printf("ok: %d launches (%d complete, %d finish-after-completion, %d tile-twice, %d never-published), %llu tiles, %llu records checked, %llu frames resolved\n",
           rounds, ends[0], ends[1], ends[2], ends[3], (unsigned long long)tiles_checked, (unsigned long long)records_checked,
           (unsigned long long)frames);
    return 0;
}
