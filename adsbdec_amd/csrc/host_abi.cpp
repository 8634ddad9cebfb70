// host_abi.cpp -- the part of the C-ABI (include/adsbdec_amd.h) that needs no device: configuration defaults, shard
// planning, the greedy resolver handle, the stitcher, the hand-off stream walk.  Built without HIP, so that the same
// object links into libadsbdec_amd.so and into the sanitizer harnesses under tests/cpp/ (which run where no GPU is).
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/adsbdec_amd_diag.h"
#include "handoff.hpp"
#include "resolver.hpp"
#include "stitch.hpp"

namespace {
inline uint64_t round_down(uint64_t v, uint64_t q) { return v - v % q; }
using adsb::HandCursor;
} // namespace

extern "C" {

int adsb_abi_version(void) { return ADSB_ABI_VERSION; }

// decoder.hip's host side is built with -mavx2 (adsbdec_amd/_build.py says why); this file is not.  adsb_create asks here,
// before anything else, whether the host can run it: a C string to show, or NULL.
const char *adsb_host_cpu_refusal(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") ? nullptr
                                          : "this build of libadsbdec_amd needs a host CPU with AVX2 (rebuild without -Xarch_host -mavx2 for an older one)";
}

void adsb_config_init(adsb_config *cfg, size_t struct_size)
{
    if (!cfg || struct_size < offsetof(adsb_config, device) + sizeof(int32_t))
        return;
    if (struct_size > sizeof *cfg) // a caller from the future: this library fills what it knows, adsb_create refuses the rest
        struct_size = sizeof *cfg;
    std::memset(cfg, 0, struct_size);
    cfg->struct_size = (uint32_t)struct_size;
    cfg->abi = ADSB_ABI_VERSION;
    cfg->device = -1;
}

// The symbol binaries built against ABI <= 4 call (their header had no macro of this name, or one that passed another size).
// Their adsb_config has another layout: what this leaves behind -- 72 zero bytes, the smallest struct there ever was -- carries
// no `abi`, so adsb_create / adsb_multi_create refuse it by name instead of misreading it.
void (adsb_config_default)(adsb_config *cfg) // (the parentheses keep the header's macro of the same name out of the way)
{
    if (cfg) {
        std::memset(cfg, 0, 72);
        cfg->struct_size = 72;
    }
}

int adsb_stitch_shards(const adsb_shard_part *parts, int n_parts, uint64_t total_samples, adsb_shard_fix *fix,
                       adsb_frame *new_frames, size_t new_cap, size_t *n_new_total)
{
    return adsb::stitch_shards(parts, n_parts, total_samples, fix, new_frames, new_cap, n_new_total);
}

int adsb_stitch_shards_ex(const adsb_shard_part *parts, int n_parts, uint64_t total_samples, adsb_shard_fix *fix,
                          adsb_frame *new_frames, size_t new_cap, size_t *n_new_total, uint64_t walk_stats[2])
{
    return adsb::stitch_shards(parts, n_parts, total_samples, fix, new_frames, new_cap, n_new_total, walk_stats);
}

int adsb_stitch_shards_stats(const adsb_shard_part *parts, int n_parts, uint64_t total_samples, adsb_shard_fix *fix,
                             adsb_frame *new_frames, size_t new_cap, size_t *n_new_total, uint64_t walk_stats[2], adsb_stats *stats)
{
    if (!stats)
        return -1;
    return adsb::stitch_shards(parts, n_parts, total_samples, fix, new_frames, new_cap, n_new_total, walk_stats, stats);
}

int adsb_shard_layout_check(size_t sizeof_shard_head, size_t sizeof_shard_part)
{
    return sizeof_shard_head == sizeof(adsb_shard_head) && sizeof_shard_part == sizeof(adsb_shard_part) ? 0 : -1;
}

size_t adsb_shard_walk(adsb_shard_head *head, const adsb_frame *frames, uint64_t total_samples, uint64_t *bases, size_t cap)
{
    if (!head || (head->n_frames && !frames) || (cap && !bases))
        return 0;
    int final = 0;
    const size_t n = adsb::walk_shard_calls(frames, head->n_frames, head->g_begin, head->g_end, total_samples, bases, cap, &final);
    head->n_bases = n <= cap ? n : 0;
    head->walk_final = final;
    return n;
}

void adsb_shard_apply_fix(adsb_frame *frames, size_t n, int64_t ts_sub)
{
    for (size_t i = 0; i < n; i++)
        frames[i].ts = (uint64_t)((int64_t)frames[i].ts - ts_sub);
}

int adsb_plan_shards(uint64_t total_samples, int n_shards, uint64_t *g_begin, uint64_t *g_end,
                     uint64_t *first_sample, uint64_t *n_samples)
{
    if (n_shards <= 0 || !g_begin || !g_end || !first_sample || !n_samples)
        return -1;
    const uint64_t m = 2 * (total_samples / 4);
    const uint64_t n_off = m >= ADSB_WINDOW ? m - ADSB_WINDOW + 1 : 0;
    for (int i = 0; i < n_shards; i++) {
        const uint64_t lo = round_down((uint64_t)((__uint128_t)n_off * (unsigned)i / (unsigned)n_shards), 28);
        const uint64_t hi = (i == n_shards - 1)
                                ? n_off
                                : round_down((uint64_t)((__uint128_t)n_off * (unsigned)(i + 1) / (unsigned)n_shards), 28);
        g_begin[i] = lo;
        g_end[i] = hi;
        // pre-halo: 8 pairs (6 needed; 8 keeps 16-byte alignment); post-halo: one window
        const uint64_t s0 = lo >= 8 ? 2 * (lo - 8) : 0;
        uint64_t s1 = hi > lo ? 2 * (hi - 1 + ADSB_WINDOW) : s0;
        if (s1 > total_samples)
            s1 = total_samples;
        first_sample[i] = s0;
        n_samples[i] = s1 > s0 ? s1 - s0 : 0;
    }
    return 0;
}

// ---- resolver handle ----------------------------------------------------------
struct adsb_resolver {
    adsb::Resolver r;
    std::vector<adsb_candidate> head;
    adsb::FormatGang gang; // adsb_resolver_set_threads
    ~adsb_resolver() { r.set_gang(nullptr); }
};

adsb_resolver *adsb_resolver_create(void)
{
    adsb_resolver *r = new (std::nothrow) adsb_resolver();
    if (r)
        r->r.reset();
    return r;
}

void adsb_resolver_destroy(adsb_resolver *r) { delete r; }

int adsb_resolver_set_threads(adsb_resolver *r, int helpers, size_t min_frames)
{
    if (!r || helpers < 0 || helpers > 15)
        return -1;
    if (helpers == 0) {
        r->r.set_gang(nullptr);
        return 0;
    }
    if (!r->gang.start(helpers))
        return -1;
    r->r.set_gang(&r->gang, min_frames);
    return (int)r->gang.helpers();
}

int adsb_resolver_feed(adsb_resolver *r, const adsb_candidate *cands, size_t n_cands,
                       const uint64_t *tries, size_t n_tries)
{
    if (!r || (n_cands && !cands) || (n_tries && !tries))
        return -1;
    r->r.feed(cands, n_cands, tries, n_tries);
    return 0;
}

int adsb_resolver_advance(adsb_resolver *r, uint64_t power_samples, uint64_t g_complete)
{
    if (!r)
        return -1;
    r->r.advance(power_samples, g_complete);
    return 0;
}

long adsb_resolver_drain(adsb_resolver *r, adsb_frame *out, size_t cap)
{
    if (!r || (!out && cap))
        return -1;
    return (long)r->r.drain(out, cap);
}

int adsb_resolver_start_chain(adsb_resolver *r, uint64_t g_begin, uint64_t head_end)
{
    if (!r)
        return -1;
    r->head.clear();
    r->r.start_chain(g_begin, head_end, &r->head);
    return 0;
}

int adsb_resolver_start_walk(adsb_resolver *r, uint64_t g_begin, uint64_t g_end, uint64_t total_samples, uint64_t *bases, size_t cap)
{
    if (!r || (cap && !bases))
        return -1;
    r->r.start_walk(g_begin, g_end, total_samples, bases, cap);
    return 0;
}

size_t adsb_resolver_walk_result(const adsb_resolver *r, int *final)
{
    if (!r)
        return 0;
    if (final)
        *final = r->r.walk_final() ? 1 : 0;
    return r->r.walk_bases();
}

long adsb_resolver_head(adsb_resolver *r, adsb_candidate *out, size_t cap)
{
    if (!r || (!out && cap))
        return -1;
    const size_t n = std::min(cap, r->head.size());
    if (n)
        std::memcpy(out, r->head.data(), n * sizeof(adsb_candidate));
    return (long)r->head.size();
}

uint64_t adsb_resolver_skipped(const adsb_resolver *r) { return r ? r->r.skipped() : 0; }

int adsb_resolver_stats(const adsb_resolver *r, adsb_stats *out)
{
    if (!r || !out)
        return -1;
    *out = const_cast<adsb_resolver *>(r)->r.stats(); // (waits for the frames that are still being written)
    return 0;
}

long adsb_handoff_walk(const void *stream, size_t granules, uint32_t n_tiles, uint32_t gen, uint32_t *tile_start,
                       uint32_t *tile_count, int *status)
{
    if (!stream || !tile_start || !tile_count || !status || granules > 0xFFFFFFFFull)
        return -1;
    // the checks read 16-byte granules with aligned loads: walk a 64-byte-aligned copy
    const size_t bytes = granules * adsb::kGranuleWords * sizeof(uint32_t);
    void *copy = nullptr;
    if (posix_memalign(&copy, 64, bytes ? bytes : 64) != 0)
        return -1;
    std::memcpy(copy, stream, bytes);
    for (uint32_t t = 0; t < n_tiles; t++)
        tile_start[t] = 0, tile_count[t] = ~0u;
    adsb::HandJob job;
    job.hand = static_cast<const uint32_t *>(copy);
    job.ntiles = n_tiles;
    job.gen = gen;
    job.cap = (uint32_t)granules;
    HandCursor cur(job, tile_start, tile_count); // (no launch behind these bytes: job.done stays null, every wait is one look)
    *status = 0;
    while (cur.frontier < n_tiles) {
        if (cur.pos >= cur.cap) {
            *status = 1;
            break;
        }
        if (!cur.wait_tile()) { // (no launch behind these bytes: one look)
            *status = 2;
            break;
        }
        const int rc = cur.take();
        if (rc != 0) {
            *status = rc < 0 ? -1 : 1;
            break;
        }
    }
    if (*status == 0 && cur.hold != ~0u)
        *status = 1; // every tile is in, but from the one that holds on they wait for the launch's end
    free(copy);
    return (long)cur.deliverable();
}

// The streaming collect's hand-over to the resolver, over an image of a stream: every tile must be in (adsb_handoff_walk's
// rules); the resolver then walks the tiles' records where they lie -- the same Resolver::advance_tiles the decoder calls --
// so that the RECORD format (a record that stands for the same frame at up to three consecutive offsets) is tested without
// a device.  Returns the tiles handed on (n_tiles), or -1.
long adsb_resolver_advance_stream(adsb_resolver *r, const void *stream, size_t granules, uint32_t n_tiles, uint32_t gen,
                                  uint64_t g_base, uint64_t power_samples, uint64_t g_complete, int with_head)
{
    if (!r || !stream || granules > 0xFFFFFFFFull || n_tiles == 0)
        return -1;
    const size_t bytes = granules * adsb::kGranuleWords * sizeof(uint32_t);
    void *copy = nullptr;
    if (posix_memalign(&copy, 64, bytes ? bytes : 64) != 0)
        return -1;
    std::memcpy(copy, stream, bytes);
    std::vector<uint32_t> ts(n_tiles, 0u), tc(n_tiles, ~0u);
    adsb::HandJob job;
    job.hand = static_cast<const uint32_t *>(copy);
    job.ntiles = n_tiles;
    job.gen = gen;
    job.cap = (uint32_t)granules;
    HandCursor cur(job, ts.data(), tc.data());
    long rc = (long)n_tiles;
    while (cur.frontier < n_tiles)
        if (cur.pos >= cur.cap || !cur.wait_tile() || cur.take() != 0) {
            rc = -1;
            break;
        }
    if (rc >= 0 && cur.hold != ~0u)
        rc = -1;
    if (rc >= 0) {
        try {
            if (with_head)
                r->r.capture_head_tiles(job.hand, ts.data(), tc.data(), 0, n_tiles, g_base);
            r->r.advance_tiles(job.hand, ts.data(), tc.data(), 0, n_tiles, g_base, power_samples, g_complete);
        } catch (const std::exception &) {
            rc = -1;
        }
        r->r.sync(); // (the gang's threads read the records where they lie: in the copy)
    }
    free(copy);
    return rc;
}

} // extern "C"
