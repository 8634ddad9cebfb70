// multi_tsan.cpp -- the multi-GPU driver (adsbdec_amd/csrc/multi.cpp: worker threads, job hand-over, stitch, gather,
// fallback, error paths) built against a FAKE device backend and run under ThreadSanitizer / AddressSanitizer.  No GPU.
//
// multi.cpp calls nothing but the public C-ABI.  The part of that ABI that needs no device is the real thing
// (host_abi.cpp: planner, resolver, stitcher); the part that does -- adsb_create, the pushes, adsb_shard_begin / _end,
// adsb_scan_shard* -- is replaced below by a "decoder" whose scan is a table look-up: in this model a capture is an array
// of uint16 in which the pair at offset g says whether a CRC-valid candidate (and which frame) or a mere DF-gate pass
// sits at g.  Everything behind the scan is the product's own code: adsb::Resolver in stream mode and in chain mode with
// the walk of the deqframe calls, exactly as decoder.hip drives it.  The expected answer is one Resolver over the whole
// capture.  Random captures (sparse, dense, frames packed back to back across the seams), 1..7 workers, host / file /
// "device-resident" sources, statistics on and off, independent streams, failing devices and failing pushes -- with
// random pauses inside the fake device so that the threads interleave differently every time.
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../adsbdec_amd/csrc/host_abi.cpp"
#include "../../adsbdec_amd/csrc/multi.cpp"
#include "../../adsbdec_amd/csrc/numa.cpp"

// ---------------------------------------------------------------- the model: what a "scan" finds in a capture
namespace fake {

constexpr uint16_t kCand = 0xC000, kTry = 0x7000; // x[2g] & 0xF000: candidate / try marker; low bits: payload

inline bool cand_at(const uint16_t *x, uint64_t first, uint64_t g, adsb_candidate *c)
{
    const uint16_t a = x[2 * g - first], b = x[2 * g + 1 - first];
    if ((a & 0xF000) != kCand)
        return false;
    std::memset(c, 0, sizeof *c);
    c->g = g;
    c->pw = 100u + (a & 0x3FF);
    const bool lng = (a & 0x400) != 0;
    c->len = lng ? 14 : 7;
    c->frame[0] = lng ? (uint8_t)(0x88 | (b & 7)) : (uint8_t)(0x58 | (b & 7)); // DF17 / DF11
    for (int k = 1; k < c->len; k++)
        c->frame[k] = (uint8_t)((g * 2654435761u + k * 40503u + b) >> 7);
    c->reserved = (uint8_t)((a >> 11) & 1u); // "repaired"
    return true;
}
inline bool try_at(const uint16_t *x, uint64_t first, uint64_t g, unsigned *code)
{
    const uint16_t a = x[2 * g - first];
    if ((a & 0xF000) == kTry) {
        *code = a % 3u;
        return true;
    }
    if ((a & 0xF000) == kCand) { // an accepted frame is a DF-gate pass too (valid.c:46,68 count it)
        *code = (a & 0x400) ? 1u : 0u;
        return true;
    }
    return false;
}

struct Fault { // test hooks: which device cannot be created, how many pushes a handle survives
    std::atomic<int> dead_device{-1};
    std::atomic<int> push_budget{-1};
    std::atomic<int> hang_ms{0}; // a handle's SECOND push sleeps this long: a device that stops answering
};
Fault g_fault;
std::atomic<int> g_gang_streams{0}, g_gang_shards{0}; // decodes whose frames went through a handle's gang (streams: decided ahead too)
thread_local std::string g_err;
thread_local std::mt19937 g_rng{std::random_device{}()};

inline void jitter()
{
    if (g_rng() % 3 == 0)
        std::this_thread::sleep_for(std::chrono::microseconds(g_rng() % 200));
}

} // namespace fake

struct adsb_decoder {
    adsb_config cfg{};
    int device = 0;
    std::string err;
    adsb::Resolver res;
    std::vector<uint16_t> x; // the stream's samples from `first` on
    uint64_t first = 0, total = 0, g_begin = 0, g_end = 0;
    bool shard = false, finished = false;
    size_t bases_cap = 0;
    uint64_t *bases = nullptr;
    std::vector<adsb_candidate> hv;
    std::vector<adsb_candidate> cands;
    std::vector<uint64_t> tries;
    int pushes_left = -1;
    int pushes = 0;
    // cfg.host_threads >= 3: a gang of frame-writing threads of the handle's own (gang.hpp), like the real decoder's; the
    // records then reach the resolver as tiles of a hand-off stream image (Resolver::advance_tiles: the product's dense path)
    adsb::FormatGang *gang = nullptr;
    std::vector<uint32_t> stream, t_start, t_count;
};

// The candidates of [g0, g1) as the image of a hand-off stream (scan_kernel_format.h): tiles of `per` offsets, a marker granule
// and two granules per record.  g_rel is relative to g0.
static uint32_t fake_tiles(adsb_decoder *d, uint64_t g0, uint64_t g1, uint32_t per)
{
    const uint32_t ntiles = (uint32_t)((g1 - g0 + per - 1) / per);
    d->stream.clear();
    d->t_start.assign(ntiles, 0);
    d->t_count.assign(ntiles, 0);
    size_t i = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
        d->stream.insert(d->stream.end(), 4, 0xDEADBEEFu); // the marker granule
        d->t_start[t] = (uint32_t)(d->stream.size() / 4);
        for (; i < d->cands.size() && d->cands[i].g < g0 + (uint64_t)(t + 1) * per; i++) {
            const adsb_candidate &c = d->cands[i];
            uint32_t w[8] = {(uint32_t)(c.g - g0), c.pw, 0, 0, 0, 0, 0, 0};
            std::memcpy(&w[2], c.frame, 14);
            w[5] = (w[5] & 0xFFFFu) | ((uint32_t)c.len << 16) | ((uint32_t)(c.reserved & 1u) << 24);
            d->stream.insert(d->stream.end(), w, w + 8);
            d->t_count[t]++;
        }
    }
    return ntiles;
}

static int dfail(adsb_decoder *d, const char *what)
{
    d->err = what;
    return -1;
}

// scan offsets [g0, g1) of samples x (x[0] = stream sample `first`, n of them) -> candidates / tries, ascending
static int fake_scan(adsb_decoder *d, const uint16_t *x, uint64_t first, uint64_t n, uint64_t g0, uint64_t g1, std::vector<adsb_candidate> &cands,
                     std::vector<uint64_t> *tries)
{
    if (g1 > g0) {
        const uint64_t need_lo = g0 >= 6 ? 2 * (g0 - 6) : 0, need_hi = 2 * (g1 - 1 + ADSB_WINDOW);
        if (first > need_lo || first + n < need_hi)
            return dfail(d, "buffer does not cover the window of the owned offsets"); // (what the real scan checks: the halo must be there)
    }
    for (uint64_t g = g0; g < g1; g++) {
        adsb_candidate c;
        unsigned code;
        if (tries && fake::try_at(x, first, g, &code))
            tries->push_back((g << 2) | code);
        if (fake::cand_at(x, first, g, &c))
            cands.push_back(c);
    }
    return 0;
}

extern "C" {

adsb_decoder *adsb_create(const adsb_config *cfg)
{
    fake::jitter();
    if (cfg->device == fake::g_fault.dead_device.load()) {
        fake::g_err = "hipSetDevice: no such device (fake)";
        return nullptr;
    }
    adsb_decoder *d = new adsb_decoder();
    d->cfg = *cfg;
    d->device = cfg->device;
    d->pushes_left = fake::g_fault.push_budget.load();
    d->res.reset();
    if (cfg->host_threads >= 3) {
        d->gang = new adsb::FormatGang;
        if (!d->gang->start(cfg->host_threads - 2)) {
            delete d->gang;
            d->gang = nullptr;
        }
    }
    return d;
}
void adsb_destroy(adsb_decoder *d)
{
    if (d && d->gang) {
        d->res.set_gang(nullptr);
        d->gang->stop();
        delete d->gang;
    }
    delete d;
}
const char *adsb_last_error(const adsb_decoder *d) { return d ? d->err.c_str() : fake::g_err.c_str(); }
void *adsb_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void adsb_host_free(void *p)
{
    if (!adsb_host_release_mapped(p)) // (numa.cpp's mappings, like the real adsb_host_free)
        free(p);
}
int adsb_host_register(void *, size_t) { return 0; }
int adsb_host_unregister(void *) { return 0; }
int adsb_device_numa_node(int device) { return device % 2; } // a made-up two-socket machine: even devices on node 0, odd ones on node 1
int adsb_device_cpulist(int, char *out, size_t cap)
{
    if (cap)
        out[0] = 0;
    return 0;
}
int adsb_get_profile_sized(const adsb_decoder *, adsb_profile *out, size_t size)
{
    std::memset(out, 0, std::min(size, sizeof *out));
    return 0;
}

int adsb_reset(adsb_decoder *d)
{
    d->x.clear();
    d->first = 0;
    d->shard = d->finished = false;
    d->res.reset();
    d->err.clear();
    return 0;
}

int adsb_push_async(adsb_decoder *d, const uint16_t *samples, size_t n)
{
    fake::jitter();
    if (d->finished)
        return dfail(d, "push after finish");
    if (d->pushes_left == 0)
        return dfail(d, "hipMemcpyAsync failed (fake)");
    if (d->pushes_left > 0)
        d->pushes_left--;
    if (++d->pushes == 2 && fake::g_fault.hang_ms.load() > 0)
        std::this_thread::sleep_for(std::chrono::milliseconds(fake::g_fault.hang_ms.load()));
    d->x.insert(d->x.end(), samples, samples + n);
    return 0;
}

int adsb_finish(adsb_decoder *d)
{
    fake::jitter();
    if (d->shard)
        return dfail(d, "a shard stream ends with adsb_shard_end");
    const uint64_t n = d->x.size(), m = 2 * (n / 4);
    const uint64_t n_off = m >= ADSB_WINDOW ? m - ADSB_WINDOW + 1 : 0;
    d->cands.clear();
    d->tries.clear();
    if (fake_scan(d, d->x.data(), 0, n, 0, n_off, d->cands, d->cfg.collect_stats ? &d->tries : nullptr))
        return -1;
    if (d->gang && !d->cfg.collect_stats && n_off) {
        // the dense path of the real decoder: tiles of a stream image, every batch decided AHEAD by the gang and taken over by
        // this thread, the frames written by the gang (decoder.hip slot_collect_streaming)
        const uint32_t per = 12880, ntiles = fake_tiles(d, 0, n_off, per);
        d->res.set_gang(d->gang, 1);
        d->res.set_ahead_min_records(1);
        d->gang->begin();
        for (uint32_t t = 0; t < ntiles;) {
            const uint32_t t1 = std::min<uint32_t>(ntiles, t + 1 + fake::g_rng() % 9);
            d->res.speculate_tiles(d->stream.data(), d->t_start.data(), d->t_count.data(), t, t1, 0);
            fake::jitter();
            d->res.advance_tiles(d->stream.data(), d->t_start.data(), d->t_count.data(), t, t1, 0, 2 * ((n + 3) / 4),
                                 std::min<uint64_t>(n_off, (uint64_t)t1 * per));
            t = t1;
        }
        d->res.sync();
        d->gang->end();
        fake::g_gang_streams++;
    } else {
        d->res.feed(d->cands.data(), d->cands.size(), d->tries.data(), d->tries.size());
        d->res.advance(2 * ((n + 3) / 4), n_off);
    }
    d->finished = true;
    return 0;
}

long adsb_take(adsb_decoder *d, const adsb_frame **frames) { return (long)d->res.take(frames); }
int adsb_get_stats(const adsb_decoder *d, adsb_stats *out)
{
    *out = const_cast<adsb_decoder *>(d)->res.stats();
    return 0;
}

static void fill_head(adsb_decoder *d, adsb_shard_head *head, size_t nf)
{
    std::memset(head, 0, sizeof *head);
    head->g_begin = d->g_begin;
    head->g_end = d->g_end;
    head->n_frames = nf;
    head->n_head = d->hv.size();
    head->head_end = std::min<uint64_t>(d->g_end, d->g_begin + ADSB_SHARD_HEAD);
    head->skipped = d->res.skipped();
    if (d->bases_cap) {
        head->n_bases = d->res.walk_bases() <= d->bases_cap ? d->res.walk_bases() : 0;
        head->walk_final = d->res.walk_final() ? 1 : 0;
    }
    const adsb_stats &st = d->res.stats();
    for (int k = 0; k < 3; k++)
        head->ok[k] = st.ok[k], head->tries[k] = st.try_[k];
    head->fixed = st.fixed;
    head->has_tries = d->cfg.collect_stats ? 1 : 0;
}

static int resolve_shard(adsb_decoder *d, const uint16_t *x, uint64_t first, uint64_t n, uint64_t *bases, size_t bases_cap,
                         adsb_shard_head *head, const adsb_frame **frames, const adsb_candidate **head_cands)
{
    d->cands.clear();
    d->tries.clear();
    if (fake_scan(d, x, first, n, d->g_begin, d->g_end, d->cands, d->cfg.collect_stats ? &d->tries : nullptr))
        return -1;
    d->hv.clear();
    d->res.start_chain(d->g_begin, std::min<uint64_t>(d->g_end, d->g_begin + ADSB_SHARD_HEAD), &d->hv);
    d->bases_cap = (bases && bases_cap) ? bases_cap : 0;
    if (d->bases_cap)
        d->res.start_walk(d->g_begin, d->g_end, d->total, bases, bases_cap);
    if (d->gang && !d->cfg.collect_stats && d->g_end > d->g_begin) {
        // chain mode over tiles: every batch decided ahead by the gang and taken over by this thread, the frames written by the
        // gang (what a worker of the real driver does on a full channel)
        const uint32_t per = 12880, ntiles = fake_tiles(d, d->g_begin, d->g_end, per);
        d->res.set_gang(d->gang, 1);
        d->res.set_ahead_min_records(1);
        d->gang->begin();
        for (uint32_t t = 0; t < ntiles;) {
            const uint32_t t1 = std::min<uint32_t>(ntiles, t + 1 + fake::g_rng() % 9);
            d->res.speculate_tiles(d->stream.data(), d->t_start.data(), d->t_count.data(), t, t1, d->g_begin); // (round 6: a chain's batches too)
            d->res.capture_head_tiles(d->stream.data(), d->t_start.data(), d->t_count.data(), t, t1, d->g_begin);
            d->res.advance_tiles(d->stream.data(), d->t_start.data(), d->t_count.data(), t, t1, d->g_begin, 0,
                                 std::min<uint64_t>(d->g_end, d->g_begin + (uint64_t)t1 * per));
            fake::jitter();
            t = t1;
        }
        d->res.advance(0, d->g_end);
        d->res.sync();
        d->gang->end();
        fake::g_gang_shards++;
        const size_t nf = d->res.take(frames);
        *head_cands = d->hv.empty() ? nullptr : d->hv.data();
        fill_head(d, head, nf);
        return 0;
    }
    // records arrive in batches, as from a kernel that is still running
    size_t ci = 0, ti = 0;
    for (uint64_t g = d->g_begin; g < d->g_end;) {
        const uint64_t upto = std::min<uint64_t>(d->g_end, g + 20000 + fake::g_rng() % 90000);
        size_t cj = ci, tj = ti;
        while (cj < d->cands.size() && d->cands[cj].g < upto)
            cj++;
        while (tj < d->tries.size() && (d->tries[tj] >> 2) < upto)
            tj++;
        d->res.feed(d->cands.data() + ci, cj - ci, d->tries.data() + ti, tj - ti);
        d->res.advance(0, upto);
        ci = cj, ti = tj, g = upto;
    }
    d->res.advance(0, d->g_end);
    const size_t nf = d->res.take(frames);
    *head_cands = d->hv.empty() ? nullptr : d->hv.data();
    fill_head(d, head, nf);
    return 0;
}

int adsb_shard_begin(adsb_decoder *d, uint64_t first_sample, uint64_t g_begin, uint64_t g_end, uint64_t total_samples, uint64_t *bases,
                     size_t bases_cap)
{
    fake::jitter();
    adsb_reset(d);
    d->shard = true;
    d->first = first_sample;
    d->g_begin = g_begin;
    d->g_end = g_end;
    d->total = total_samples;
    d->bases = bases; // (the fake resolves when the shard ends: the walk's array is remembered until then)
    d->bases_cap = bases_cap;
    return 0;
}

int adsb_shard_end(adsb_decoder *d, adsb_shard_head *head, const adsb_frame **frames, const adsb_candidate **head_cands)
{
    fake::jitter();
    if (!d->shard)
        return dfail(d, "adsb_shard_end without adsb_shard_begin");
    d->finished = true;
    return resolve_shard(d, d->x.data(), d->first, d->x.size(), d->bases, d->bases_cap, head, frames, head_cands);
}

int adsb_scan_shard_resolved_take(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n, uint64_t g_begin,
                                  uint64_t g_end, uint64_t total_samples, adsb_shard_head *head, const adsb_frame **frames,
                                  const adsb_candidate **head_cands, uint64_t *bases, size_t bases_cap)
{
    fake::jitter();
    adsb_reset(d);
    d->g_begin = g_begin;
    d->g_end = g_end;
    d->total = total_samples;
    return resolve_shard(d, static_cast<const uint16_t *>(device_samples), first_sample, n, bases, bases_cap, head, frames, head_cands);
}

int adsb_scan_shard(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n, uint64_t g_begin, uint64_t g_end,
                    adsb_candidate *cands, size_t cand_cap, size_t *n_cands, uint64_t *tries, size_t try_cap, size_t *n_tries)
{
    fake::jitter();
    std::vector<adsb_candidate> cv;
    std::vector<uint64_t> tv;
    if (fake_scan(d, static_cast<const uint16_t *>(device_samples), first_sample, n, g_begin, g_end, cv, &tv))
        return -1;
    *n_cands = cv.size();
    *n_tries = tv.size();
    if (cv.size() > cand_cap || tv.size() > try_cap)
        return -2;
    if (!cv.empty())
        std::memcpy(cands, cv.data(), cv.size() * sizeof(adsb_candidate));
    if (!tv.empty())
        std::memcpy(tries, tv.data(), tv.size() * sizeof(uint64_t));
    return 0;
}

int adsb_scan_shard_host(adsb_decoder *d, const uint16_t *host_samples, uint64_t first_sample, size_t n, uint64_t g_begin, uint64_t g_end,
                         adsb_candidate *cands, size_t cand_cap, size_t *n_cands, uint64_t *tries, size_t try_cap, size_t *n_tries)
{
    return adsb_scan_shard(d, host_samples, first_sample, n, g_begin, g_end, cands, cand_cap, n_cands, tries, try_cap, n_tries);
}

} // extern "C"

// ---------------------------------------------------------------- the test
static std::vector<uint16_t> make_capture(std::mt19937 &rng, int kind)
{
    const uint64_t n = kind == 3 ? 2000 + rng() % 300000 : 400000 + rng() % 3000000;
    std::vector<uint16_t> x(n, 0);
    const uint64_t m = n / 2;
    auto put_cand = [&](uint64_t g, bool lng) {
        if (g + 1 >= m)
            return;
        x[2 * g] = (uint16_t)(fake::kCand | (lng ? 0x400 : 0) | (rng() % 16 == 0 ? 0x800 : 0) | (rng() & 0x3FF));
        x[2 * g + 1] = (uint16_t)rng();
    };
    if (kind == 1) { // frames packed back to back: every seam cuts through one, the horizon lands among them
        for (uint64_t g = 3000; g + 1300 < m; g += (rng() % 8 == 0 ? 640 : 1200)) {
            const bool lng = rng() % 4 != 0;
            put_cand(g, lng);
            if (rng() % 2)
                put_cand(g + 1, lng); // the shifted copy a real frame decodes at
        }
    } else {
        const uint64_t gap = kind == 2 ? 300 : 9000;
        for (uint64_t g = rng() % gap; g < m; g += 1 + rng() % (2 * gap)) {
            const bool lng = rng() % 5 != 0;
            for (int d = 0; d < 1 + (int)(rng() % 3); d++)
                put_cand(g + d, lng);
        }
    }
    const uint64_t try_gap = kind == 2 ? 40 : 700;
    for (uint64_t g = rng() % try_gap; g < m; g += 1 + rng() % (2 * try_gap))
        if ((x[2 * g] & 0xF000) == 0)
            x[2 * g] = (uint16_t)(fake::kTry | (rng() & 0xFFF));
    return x;
}

struct Expected {
    std::vector<adsb_frame> frames;
    adsb_stats stats{};
};

static Expected sequential(const std::vector<uint16_t> &x, bool stats)
{
    adsb_decoder d;
    d.cfg.collect_stats = stats;
    d.res.reset();
    d.x = x;
    Expected e;
    if (adsb_finish(&d) != 0) {
        fprintf(stderr, "sequential decode failed: %s\n", d.err.c_str());
        exit(2);
    }
    const adsb_frame *fp = nullptr;
    const size_t n = d.res.take(&fp);
    e.frames.assign(fp, fp + n);
    e.stats = d.res.stats();
    return e;
}

static bool same_frames(const adsb_frame *a, size_t na, const std::vector<adsb_frame> &b, const char *what)
{
    if (na != b.size()) {
        fprintf(stderr, "%s: %zu frames, expected %zu\n", what, na, b.size());
        return false;
    }
    for (size_t i = 0; i < na; i++)
        if (a[i].g != b[i].g || a[i].ts != b[i].ts || a[i].pw != b[i].pw || a[i].len != b[i].len ||
            std::memcmp(a[i].frame, b[i].frame, 14) != 0 || a[i].reserved != b[i].reserved) {
            fprintf(stderr, "%s: frame %zu differs (g %llu ts %llu vs g %llu ts %llu)\n", what, i, (unsigned long long)a[i].g,
                    (unsigned long long)a[i].ts, (unsigned long long)b[i].g, (unsigned long long)b[i].ts);
            return false;
        }
    return true;
}

static bool same_stats(const adsb_stats &a, const adsb_stats &b, const char *what)
{
    if (std::memcmp(&a, &b, sizeof a) == 0)
        return true;
    fprintf(stderr, "%s: Try %llu/%llu/%llu Ok %llu/%llu/%llu fixed %llu, expected Try %llu/%llu/%llu Ok %llu/%llu/%llu fixed %llu\n", what,
            (unsigned long long)a.try_[0], (unsigned long long)a.try_[1], (unsigned long long)a.try_[2], (unsigned long long)a.ok[0],
            (unsigned long long)a.ok[1], (unsigned long long)a.ok[2], (unsigned long long)a.fixed, (unsigned long long)b.try_[0],
            (unsigned long long)b.try_[1], (unsigned long long)b.try_[2], (unsigned long long)b.ok[0], (unsigned long long)b.ok[1],
            (unsigned long long)b.ok[2], (unsigned long long)b.fixed);
    return false;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 24;
    std::mt19937 rng(777);
    char path[] = "/tmp/adsb_multi_tsan_XXXXXX";
    const int tfd = mkstemp(path);
    if (tfd < 0)
        return 2;
    close(tfd);
    int decodes = 0, shards = 0, fallbacks = 0, placements = 0;
    for (int round = 0; round < rounds; round++) {
        const bool stats = round % 2 == 0;
        const int workers = 1 + (int)(rng() % 7);
        adsb_config cfg;
        adsb_config_default(&cfg);
        cfg.collect_stats = stats;
        cfg.host_threads = round % 4 == 1 ? 5 : 0; // a gang of three frame-writing threads per worker's handle now and then (never with the table: the fake counts tries in the resolver)
        cfg.stage_samples = round % 3 == 0 ? 1u << 17 : 0; // small pieces now and then
        std::vector<int> devs(workers);
        for (int i = 0; i < workers; i++)
            devs[i] = i % 3;
        adsb_multi *m = adsb_multi_create(&cfg, workers, devs.data());
        if (!m) {
            fprintf(stderr, "round %d: adsb_multi_create failed: %s\n", round, adsb_multi_last_error(nullptr));
            return 1;
        }
        for (int rep = 0; rep < 3; rep++) {
            const std::vector<uint16_t> x = make_capture(rng, (round + rep) % 4);
            const Expected want = sequential(x, stats);
            const adsb_frame *fp = nullptr;
            long n = -1;
            const int src = (round + rep) % 3;
            uint16_t *placed = nullptr;
            if (src == 0 && rep == 0) {
                // the capture in an array laid out shard by shard on the nodes of the devices that pull it (numa.cpp); this
                // machine has one node, the made-up map two: mbind for node 1 is refused or not, the pages live on node 0
                // either way, and the placement query must say so -- workers on even devices local, on odd devices remote
                placed = adsb_multi_host_alloc(m, x.size());
                if (!placed) {
                    fprintf(stderr, "adsb_multi_host_alloc failed: %s\n", adsb_multi_last_error(m));
                    return 1;
                }
                std::memcpy(placed, x.data(), x.size() * sizeof(uint16_t));
                n = adsb_multi_decode_host(m, placed, x.size(), &fp);
                adsb_multi_info inf;
                adsb_multi_get_info(m, &inf);
                for (int i = 0; n >= 0 && !inf.fallback && i < inf.shards; i++) {
                    adsb_worker_placement pl;
                    if (adsb_multi_worker_placement(m, i, &pl) != 0 || pl.device != devs[i] || pl.device_node != devs[i] % 2) {
                        fprintf(stderr, "worker %d: placement of the wrong device\n", i);
                        return 1;
                    }
                    if (pl.slice_node >= 0 && (pl.slice_node != 0 || pl.local_fraction != (pl.device_node == 0 ? 1.0 : 0.0))) {
                        fprintf(stderr, "worker %d (device %d, node %d): slice on node %d, local fraction %.2f\n", i, pl.device, pl.device_node,
                                pl.slice_node, pl.local_fraction);
                        return 1;
                    }
                    placements += pl.slice_node >= 0;
                }
            } else if (src == 0) {
                n = adsb_multi_decode_host(m, x.data(), x.size(), &fp);
            } else if (src == 1) {
                FILE *f = fopen(path, "wb");
                fwrite(x.data(), 2, x.size(), f);
                fclose(f);
                n = adsb_multi_decode_file(m, path, &fp);
            } else {
                std::vector<uint64_t> p[4];
                for (auto &v : p)
                    v.resize(workers);
                const int k = adsb_multi_plan(m, x.size(), p[0].data(), p[1].data(), p[2].data(), p[3].data());
                std::vector<const void *> slices(k);
                for (int i = 0; i < k; i++)
                    slices[i] = x.data() + p[2][i];
                n = adsb_multi_decode_device(m, x.size(), slices.data(), k, &fp);
            }
            char what[96];
            snprintf(what, sizeof what, "round %d rep %d (%d workers, source %d, stats %d)", round, rep, workers, src, (int)stats);
            if (n < 0) {
                fprintf(stderr, "%s: decode failed: %s\n", what, adsb_multi_last_error(m));
                return 1;
            }
            if (!same_frames(fp, (size_t)n, want.frames, what))
                return 1;
            adsb_stats st;
            if (stats && (adsb_multi_get_stats(m, &st) != 0 || !same_stats(st, want.stats, what)))
                return 1;
            if (placed)
                adsb_host_free(placed); // (the workers are done with it: the call has returned)
            adsb_multi_info inf;
            adsb_multi_get_info(m, &inf);
            decodes++;
            shards += inf.shards;
            fallbacks += inf.fallback;
        }
        // independent streams, more of them than workers
        {
            std::vector<std::vector<uint16_t>> xs;
            std::vector<const uint16_t *> ptrs;
            std::vector<size_t> lens;
            for (int s = 0; s < workers + 2; s++) {
                xs.push_back(make_capture(rng, 3));
                ptrs.push_back(xs.back().data());
                lens.push_back(xs.back().size());
            }
            if (adsb_multi_decode_streams_host(m, (int)xs.size(), ptrs.data(), lens.data()) != 0) {
                fprintf(stderr, "round %d: streams failed: %s\n", round, adsb_multi_last_error(m));
                return 1;
            }
            for (size_t s = 0; s < xs.size(); s++) {
                const Expected want = sequential(xs[s], stats);
                const adsb_frame *fp = nullptr;
                const long n = adsb_multi_stream_frames(m, (int)s, &fp);
                adsb_stats st;
                if (n < 0 || !same_frames(fp, (size_t)n, want.frames, "stream") ||
                    (stats && (adsb_multi_stream_stats(m, (int)s, &st) != 0 || !same_stats(st, want.stats, "stream"))))
                    return 1;
            }
        }
        adsb_multi_destroy(m);
    }
    // ---- error paths: a device that cannot be created; pushes that fail half-way through a shard
    {
        adsb_config cfg;
        adsb_config_default(&cfg);
        fake::g_fault.dead_device.store(2);
        const int devs[4] = {0, 1, 2, 3};
        if (adsb_multi_create(&cfg, 4, devs) != nullptr || !strstr(adsb_multi_last_error(nullptr), "device 2 (worker 2)")) {
            fprintf(stderr, "a dead device must fail adsb_multi_create and be named: '%s'\n", adsb_multi_last_error(nullptr));
            return 1;
        }
        fake::g_fault.dead_device.store(-1);
        fake::g_fault.push_budget.store(1);
        adsb_multi *m = adsb_multi_create(&cfg, 3, devs);
        fake::g_fault.push_budget.store(-1);
        if (!m)
            return 1;
        cfg.stage_samples = 0;
        std::vector<uint16_t> x = make_capture(rng, 0);
        x.resize(std::max<size_t>(x.size(), 120u << 20 >> 1)); // > 32 MiB per shard: more than one piece each
        const adsb_frame *fp = nullptr;
        if (adsb_multi_decode_host(m, x.data(), x.size(), &fp) >= 0 || !strstr(adsb_multi_last_error(m), "adsb_push_async failed")) {
            fprintf(stderr, "a failing push must fail the decode and be named: '%s'\n", adsb_multi_last_error(m));
            return 1;
        }
        if (adsb_multi_decode_file(m, "/nonexistent/capture.u16", &fp) >= 0)
            return 1;
        adsb_multi_destroy(m);
        // ---- a device that stops answering in the middle of a shard: the driver gives up after cfg.wait_timeout_s + margin,
        // names the worker, stays broken, and adsb_multi_destroy returns at once (the worker cleans up after itself later)
        cfg.wait_timeout_s = 1; // + 5 s of margin
        fake::g_fault.hang_ms.store(9000);
        m = adsb_multi_create(&cfg, 2, devs);
        if (!m)
            return 1;
        // (the capture of a decode that was given up stays borrowed by the workers that may still be inside it: it is the
        // caller's to keep alive -- here for the rest of the process, reachable from a global so that no leak is reported)
        static uint16_t *hang_x = nullptr;
        hang_x = static_cast<uint16_t *>(malloc(x.size() * sizeof(uint16_t)));
        std::memcpy(hang_x, x.data(), x.size() * sizeof(uint16_t));
        const auto t_h = std::chrono::steady_clock::now();
        const long rc_h = adsb_multi_decode_host(m, hang_x, x.size(), &fp);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_h).count();
        if (rc_h >= 0 || !strstr(adsb_multi_last_error(m), "no sign of life") || !strstr(adsb_multi_last_error(m), "worker") || waited < 5.5 || waited > 20.0) {
            fprintf(stderr, "a worker that stops answering must end the decode after the limit and be named: rc %ld after %.1f s, '%s'\n", rc_h, waited, adsb_multi_last_error(m));
            return 1;
        }
        if (adsb_multi_decode_host(m, hang_x, x.size(), &fp) >= 0 || !strstr(adsb_multi_last_error(m), "unusable"))
            return 1;
        const auto t_d = std::chrono::steady_clock::now();
        adsb_multi_destroy(m);
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_d).count() > 1.0) {
            fprintf(stderr, "destroying a broken driver must not wait for its workers\n");
            return 1;
        }
        fake::g_fault.hang_ms.store(0);
        std::this_thread::sleep_for(std::chrono::milliseconds(4500)); // the orphaned workers come back, free their handles and end
    }
    unlink(path);
    printf("ok: %d sharded decodes (%d shards, %d fell back to one stream), streams and error paths; %d slices asked where they live; "
           "%d shards and %d streams through a handle's gang\n", decodes, shards, fallbacks, placements, fake::g_gang_shards.load(),
           fake::g_gang_streams.load());
    return 0;
}
