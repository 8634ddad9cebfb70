// decoder.hip -- host side of libadsbdec_amd.so: stream state, device staging,
// kernel launches, record gather and the C-ABI of include/adsbdec_amd.h.
//
// Data layout in HBM
//   stage[2]   uint16 samples; the current one holds stream samples
//              [stage_first, stage_first+stage_fill): what has not been dropped yet plus the
//              newest push (pushes are appended; the scanned part is dropped -- the tail moved to
//              the other buffer -- when the buffer is half full).  stage_first is a multiple of
//              8 samples so that pair index/4 alignment and 16-byte loads line up with the stream.
//              adsb_push_async copies into it on two copy streams of its own (push_copy).
//              Ordering rule of the copies into it: two writers that are not ordered never share a
//              cache line (process_stage, push_copy).
//   d_tries    one dword per DF-gate pass (collect_stats of a stream: counted on the device, on a
//              count stream of its own): a region of kTryRegion words per tile + a launch-wide list
//   counters   adsb::kDevCounterWords dwords per launch slot, every counter on a cache line of its own
// and in pinned host memory, written by the kernel
//   hand       the hand-off stream: a marker + the kept records of every tile (scan_kernel.h),
//              consumed while the kernel runs
//   cands      "loose" list, kCandWords dwords per record, appended with one atomic: records
//              that could not go through the stream; collected after completion
//   tries      try words of per-shard scans, which hand the list back to the caller
// A buffer pushed with adsb_push_device() at a stream position that is a multiple
// of 8 samples and a 16-byte aligned address is scanned IN PLACE: only the ~4 KiB
// seam with the previous push and the ~5 KiB tail go through the staging buffer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include <emmintrin.h>
#include <pthread.h>
#include <sched.h>

#include "../../include/adsbdec_amd_diag.h"
#include "config_abi.hpp"
#include "handoff.hpp"
#include "resolver.hpp"
#include "scan_kernel.h"
#include "stitch.hpp"

namespace {

thread_local std::string g_create_error;
const char *g_cpu_refusal = nullptr; // set by adsb_create on a host without AVX2 (a plain pointer store: nothing of this file's
                                     // vector code has run by then); adsb_last_error(NULL) shows it

// The shipped library reads NO environment variable: what a test must be able to force is a member of adsb_config
// (debug_*).  Builds with -DADSB_TUNING (tools/build_variant.sh; never the one in adsbdec_amd/lib) keep a few knobs for
// A/B runs and diagnosis: ADSB_CHUNK_MI, ADSB_ALT_STREAMS, ADSB_DEBUG_HOST, ADSB_DEBUG_TIMELINE, ADSB_DEBUG_ASYNC.
#ifdef ADSB_TUNING
inline const char *tuning_env(const char *name) { return getenv(name); }
#else
inline const char *tuning_env(const char *) { return nullptr; }
#endif

constexpr uint64_t kDefaultStageSamples = 32ull << 20; // 64 MiB per staging buffer
constexpr uint64_t kStageSlack = 4096;                 // samples kept free for alignment padding
constexpr size_t kInPlaceMinSamples = 1u << 16;
constexpr size_t kSeamSamples = 4096; // > 2*(28+8+1196): enough for the first in-place tile's pre-halo

inline uint64_t round_down(uint64_t v, uint64_t q) { return v - v % q; }

} // namespace

// One scan in flight: the kernel of a chunk of offsets writes its records straight into
// this slot's PINNED HOST buffers (the records are tens of bytes per frame; PCIe writes
// are free next to the sample traffic) -- no device-to-host copy of records, ever.
struct ScanSlot {
    uint32_t *d_counters = nullptr; // device: adsb::kDevCounterWords (ScanArgs::counters)
    uint32_t *h_counters = nullptr; // pinned, written by the launch's report kernel (ScanArgs::report): two copies
                                    // used in turn (ev_cur), so that a launch's kernel time can be read
                                    // behind the slot's NEXT launch instead of in front of it
    uint32_t *cands = nullptr;      // pinned, written by the kernel
    uint32_t *tries = nullptr;      // pinned, written by the kernel (per-shard scans that return the list)
    uint32_t *d_tries = nullptr;    // device: statistics runs of a stream count tries on the device
    size_t cand_cap = 0, try_cap = 0, d_try_cap = 0;
    // d_tries = [d_try_tiles regions of adsb::kTryRegion words][launch-wide list of d_try_cap words]; a tile's
    // whole-tile round writes its region and d_try_counts[tile] (scan_kernel.h), the list takes the rest
    uint32_t *d_try_counts = nullptr;
    size_t d_try_tiles = 0;
    bool try_regions = false;       // the launch in flight uses the regions
    bool tries_on_device = false;   // which of the two the launch in flight uses
    int ev_cur = 0;                  // copy of the launch in flight
    uint64_t ev_offsets[2] = {0, 0}; // offsets of the launch each copy belongs to
    hipEvent_t ev_ready[2] = {nullptr, nullptr}; // the kernel has completed (report and loose list are in)
    uint32_t *hc() { return h_counters + ev_cur * adsb::kCounterWords; }
    // streaming hand-off (scan_kernel.h): one stream of self-validating granules, pinned
    uint32_t *hand = nullptr;
    size_t hand_cap = 0;   // granules hand can hold
    uint32_t ntiles = 0;   // tiles of the launch in flight
    bool streaming = false;
    adsb::ScanArgs args{};
    bool busy = false;
    uint64_t piece = 0; // adsb_push_async: the launch belongs to this push piece (collected one piece later)
    hipStream_t launch_stream = nullptr; // where the launch in flight (or the slot's last one) was enqueued
    bool prof_pending[2] = {false, false}; // kernel time of a collected launch not read yet
    hipEvent_t ev_count = nullptr; // statistics runs: the count pass over d_tries (count stream) has ended;
    bool count_pending = false;    // the slot's next scan waits for it before it overwrites the list
};

constexpr int kSlots = 4;
// Offsets per launch.  With the streaming
// hand-off the host already overlaps a launch while it runs, so launches are as large
// as the record buffers sensibly allow (each launch carries ~20 us of ramp and tail);
// the collect-after-completion path needs several launches in flight to overlap at all.
static uint64_t chunk_offsets(bool streaming)
{
    static const uint64_t forced = [] {
        const char *e = tuning_env("ADSB_CHUNK_MI");
        const uint64_t mi = e ? strtoull(e, nullptr, 10) : 0;
        return (mi >= 1 && mi <= 512) ? mi : 0;
    }();
    const uint64_t mi = forced ? forced : (streaming ? 128 : 64);
    return 28ull * ((mi << 20) / 28);
}

struct ScanSink { // where collected records go: a caller's vectors, or (null) the stream's resolver
    std::vector<adsb_candidate> *cands = nullptr;
    std::vector<uint64_t> *tries = nullptr;
};


struct adsb_decoder {
    adsb_config cfg{};
    adsb_debug_config dbg{}; // the test knobs, copied at adsb_create (adsb_config.debug)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // A second compute stream: the launches of a multi-launch IN-PLACE scan (adsb_push_device*, adsb_scan_shard*)
    // alternate between the two, so that launch k+1's first tiles fill the slots launch k's last tiles leave empty
    // (a launch drains for about one tile life, ~40 us of falling occupancy; on one stream the next launch cannot
    // start before the previous one -- and the report kernel behind it -- has ended).  Staged scans stay on `stream`,
    // behind their copies.  A caller-supplied cfg.stream turns it off.
    hipStream_t stream2 = nullptr;
    bool alt_next = false; // scan_submit: the launches being submitted may alternate
    std::string err;

    // stream position
    uint64_t n_samples = 0; // samples accepted
    uint64_t g_scanned = 0; // every offset below has been submitted to the device
    bool finished = false;

    // staging
    uint16_t *stage[2] = {nullptr, nullptr};
    int cur = 0;
    uint64_t stage_cap = 0;   // samples per staging buffer
    uint64_t stage_first = 0; // stream index of stage[cur][0]
    uint64_t stage_fill = 0;  // samples held
    bool copy_unconfirmed = false; // a copy of caller's samples has been enqueued on the scan stream and no scan
                                   // launched behind it has been collected yet (the caller's buffer is still in use)

    uint32_t *d_synd = nullptr; // 14 x 256 CRC-24 syndrome table (scan_kernel.h)
    uint32_t *d_fix = nullptr;  // single-bit syndrome hash (extension, cfg.fix_1bit)
    uint32_t fix_mul = 0;
    int n_cus = 256;
    ScanSlot slots[kSlots];
    int slot_head = 0, slot_count = 0; // FIFO of busy slots
    ScanSink sink;                     // sink of the scans in flight

    adsb_profile prof{};
    adsb::Resolver res;
    std::vector<uint32_t> order, scratch_a, scratch_b, gather, tile_start, tile_count;
    adsb::StreamReader *reader = nullptr;    // the thread that reads the hand-off stream (slot_collect_streaming): cfg.host_threads = 2
                                             // from the start, 0 (auto) from the first launch that follows a dense one
    bool reader_failed = false;              // no thread could be had: do not try again
    adsb::FormatGang *gang = nullptr;        // the threads that write the frames of dense launches (gang.hpp): cfg.host_threads >= 3, or auto
    bool gang_failed = false;
    int gang_l3 = -1;
    uint32_t reader_min_tiles = 1024; // launches below this many tiles are collected by the calling thread alone
    uint64_t last_launch_records = 0; // records the previous launch handed over (auto: the thread pays from kAutoReaderRecords on)
    uint64_t last_launch_offsets = 0; // ... out of this many offsets
    bool no_streaming = false; // dbg.no_streaming: always collect after completion
    uint64_t shard_head = ADSB_SHARD_HEAD; // offsets of a resolved shard whose candidates are ALL kept for the stitcher (dbg.shard_head)
    int dbg_async = 0;         // tuning builds only (ADSB_DEBUG_ASYNC, tools/async_race.py): 1 = wait for every async copy,
                               // 2 = copies on the scan stream, 4 = tail copies not ordered before the next copy (the old race)
    // device-side visited-try count (scan_kernel.h TryCountArgs)
    uint64_t *d_carry[2] = {nullptr, nullptr};
    uint32_t *d_carry_n = nullptr; // device: three counts in rotation (in, out, next: TryCountArgs)
    int carry_n_cur = 0;
    bool carry_maybe = false;      // a non-final pass has run since the last final one: its carry may be non-empty
    int carry_cur = 0;
    adsb::TryFrame *d_frames = nullptr;
    size_t frames_cap = 0;
    unsigned long long *d_try_acc = nullptr; // device: visited tries per DF code since reset + overflow flag
    bool tries_unread = false;               // passes have been enqueued since the statistics were last read
    bool acc_dirty = false;                  // ... since d_try_acc was last zeroed
    // pinned upload buffers, three in rotation: the resolver logs the frames it accepts straight into one ([0] is kept
    // for the last frame of the pass before), the pass being prepared uploads from the second, the third may still be
    // in flight -- so preparing a pass copies nothing and never waits for an upload
    static constexpr int kFrameBufs = 3;
    adsb::TryFrame *h_frames[kFrameBufs] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_frames[kFrameBufs] = {nullptr, nullptr, nullptr};
    int log_buf = 0; // the buffer the resolver is logging into
    hipStream_t count_stream = nullptr; // statistics runs: frame uploads + count kernels (count_tries_pass)
    bool frames_pending[kFrameBufs] = {false, false, false};
    bool final_follows = false;   // adsb_push_device_final: the end-of-stream count pass comes next
    uint32_t deferred_n = 0;
    ScanSlot *deferred_slot = nullptr;
    // A count pass that has been prepared (frames in h_frames[b], arguments fixed) but not enqueued yet: its HIP
    // calls (~14 us of host time) are made right BEHIND the next scan launch instead of in front of it
    // (count_flush), or when the statistics are asked for.  An adsb_reset in between queues its clearing of the
    // accumulators behind it.
    struct PendingCount {
        bool valid = false, clear_after = false;
        adsb::TryCountArgs a{};
        size_t nf = 0;
        int b = 0;
        const adsb::TryFrame *src = nullptr; // first frame to upload (h_frames[b], or one further without a previous frame)
        ScanSlot *slot = nullptr; // records its ev_count
        hipEvent_t after = nullptr; // the scan (and the report kernel behind it) whose tries the pass reads
    } pending;
    uint64_t deferred_base = 0;
    bool have_prev_frame = false; // last accepted frame of earlier passes (its span may cover later tries)
    uint64_t prev_frame_g = 0;
    uint32_t prev_frame_span = 0;
    uint32_t launch_gen = 0;   // makes every launch's hand-off tags distinct
    // adsb_push_async: host-to-device copies run on streams of their own, used in turn (measured
    // with rocprofv3 --memory-copy-trace: two copies queued on ONE stream start ~15 us apart,
    // whatever their size -- at the reference's 2 MiB per call that is a quarter of the link;
    // on alternating streams the next copy starts while the previous one is still running)
    static constexpr int kCopyStreams = 2;
    hipStream_t copy_stream[kCopyStreams] = {nullptr, nullptr};
    hipEvent_t ev_copy[kCopyStreams] = {nullptr, nullptr};
    hipEvent_t ev_tail = nullptr; // behind a staging compaction's tail copy (process_stage): the copy streams wait for it
    hipEvent_t ev_wait = nullptr; // wait_stream's marker (created at its first use)
    uint64_t piece = 0;        // pieces pushed asynchronously so far

    // Shard-stream mode (adsb_shard_begin .. adsb_shard_end): the stream starts at sample shard_first instead of 0, ends
    // behind offset shard_g_end instead of at the end-of-file horizon, and the resolver runs in chain mode.
    bool shard_on = false;
    uint64_t shard_g_begin = 0, shard_g_end = 0;
    size_t shard_bases_cap = 0;
    std::vector<adsb_candidate> shard_hv;
    uint16_t *win_buf = nullptr; // adsb_scan_shard_host: device copy of the caller's window
    size_t win_cap = 0; // head candidates (handed out in place by adsb_shard_end)

    int fail(const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return -1;
    }
};

#define HIP_TRY(d, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return (d)->fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                             __LINE__);                                                       \
    } while (0)

namespace {

// Every wait for the device inside the library has a deadline (cfg.wait_timeout_s, default 120 s): a kernel or copy that never
// completes -- a wedged queue, a device that has gone away -- ends the call with -1 and a message that says what was waited
// for, instead of a host thread that never returns (the reference's counterpart is a read() that cannot hang).  Polls with
// pauses for the first ~200 us (the usual case: the work is microseconds from its end), then sleeps between looks.
template <class Query>
int wait_until_done(adsb_decoder *d, Query &&query, const char *what)
{
    using clk = std::chrono::steady_clock;
    hipError_t q;
    for (int spin = 0; spin < 4096; spin++) {
        if ((q = query()) != hipErrorNotReady)
            goto out;
        _mm_pause(); // (x86-64 only by handoff.hpp's #error: this file includes it)
    }
    {
        const auto t0 = clk::now();
        const auto limit = std::chrono::seconds(d->cfg.wait_timeout_s > 0 ? d->cfg.wait_timeout_s : 120);
        unsigned nap_us = 20;
        for (;;) {
            for (int spin = 0; spin < 256; spin++) {
                if ((q = query()) != hipErrorNotReady)
                    goto out;
                _mm_pause(); // (x86-64 only by handoff.hpp's #error: this file includes it)
            }
            if (clk::now() - t0 > limit)
                return d->fail("the device did not finish %s within %lld s (wedged queue or lost device?): giving up", what,
                               (long long)limit.count());
            if (clk::now() - t0 > std::chrono::milliseconds(2)) { // long waits (copies of GiBs, first-touch page faults) sleep between looks
                usleep(nap_us);
                nap_us = std::min(nap_us * 2, 200u);
            }
        }
    }
out:
    if (q != hipSuccess)
        return d->fail("waiting for %s failed: %s", what, hipGetErrorString(q));
    return 0;
}
inline int wait_event(adsb_decoder *d, hipEvent_t ev, const char *what)
{
#ifdef ADSB_BLOCKING_WAITS // (A/B builds: the runtime's own blocking waits, as until round 4)
    const hipError_t e = hipEventSynchronize(ev);
    return e == hipSuccess ? 0 : d->fail("waiting for %s failed: %s", what, hipGetErrorString(e));
#endif
    return wait_until_done(d, [ev] { return hipEventQuery(ev); }, what);
}
// "Everything enqueued on st so far": ONE marker (an event recorded behind it) and polls of that event.  Not polls of
// hipStreamQuery: each of those has the runtime enqueue a marker of its own while the stream is busy, and a statistics step,
// which waits for its count pass this way once per call, got 6 % slower for it (profiles/r5_ab_runs.txt section 5).
inline int wait_stream(adsb_decoder *d, hipStream_t st, const char *what)
{
#ifdef ADSB_BLOCKING_WAITS
    const hipError_t es = hipStreamSynchronize(st);
    return es == hipSuccess ? 0 : d->fail("waiting for %s failed: %s", what, hipGetErrorString(es));
#endif
    if (!d->ev_wait && hipEventCreateWithFlags(&d->ev_wait, hipEventDisableTiming) != hipSuccess) {
        d->ev_wait = nullptr;
        (void)hipGetLastError();
        return wait_until_done(d, [st] { return hipStreamQuery(st); }, what);
    }
    // (an idle stream -- most of the streams adsb_reset and adsb_finish wait for -- answers the first question; a marker
    // recorded on an idle stream would cost a round trip to the device, five of them per adsb_reset: measured, +9 % on the
    // statistics step)
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess)
        return 0;
    if (q != hipErrorNotReady)
        return d->fail("waiting for %s failed: %s", what, hipGetErrorString(q));
    const hipError_t e = hipEventRecord(d->ev_wait, st);
    if (e != hipSuccess)
        return d->fail("hipEventRecord (waiting for %s) failed: %s", what, hipGetErrorString(e));
    return wait_event(d, d->ev_wait, what);
}
#define WAIT_EVENT(d, ev, what)   do { if (wait_event((d), (ev), (what))) return -1; } while (0)
#define WAIT_STREAM(d, st, what)  do { if (wait_stream((d), (st), (what))) return -1; } while (0)

inline uint64_t power_samples_produced(uint64_t n_samples)
{
    return 2 * (n_samples / 4); // air.c:59-92: two power samples per four input samples
}

// One past the last offset that can be scanned once `n_samples` samples of the stream are in: the whole 1196-sample
// window of an offset must have been produced.  A stream produces power samples in twos (air.c:59-92); a shard's samples
// end where its last owned window does (adsb_plan_shards), which need not be a whole quad, and its scan never goes
// beyond the offsets it owns.
uint64_t scannable_end(const adsb_decoder *d, uint64_t n_samples, bool final)
{
    uint64_t m = power_samples_produced(n_samples);
    if (d->shard_on)
        m = n_samples / 2; // every complete pair
    uint64_t g_end = m >= ADSB_WINDOW ? m - ADSB_WINDOW + 1 : 0;
    if (!final)
        g_end = round_down(g_end, 28);
    if (d->shard_on && g_end > d->shard_g_end)
        g_end = d->shard_g_end;
    return g_end;
}

constexpr size_t kTryStateBytes = 4 * sizeof(unsigned long long) + 4 * sizeof(uint32_t); // d_try_acc + d_carry_n
constexpr uint32_t kCarryCap = 1u << 20; // undecided tries carried between count passes (a few hundred in practice)

int count_flush(adsb_decoder *d);

// adsb_push_async: wait for the copy of the last piece (a no-op when a collected scan has implied it).
int wait_last_copy(adsb_decoder *d)
{
    if (d->piece == 0 || d->dbg_async == 2)
        return 0;
    HIP_TRY(d, hipSetDevice(d->device));
    WAIT_EVENT(d, d->ev_copy[d->piece % adsb_decoder::kCopyStreams], "the host-to-device copy of the previous piece");
    return 0;
}

// The reference's ring index `fidx` is a uint32_t that counts input samples (air.c:34): at 2^32 samples it
// wraps, 2^32 mod 14 = 4, and the ring phase jumps.  No parity is defined beyond that point, so a
// stream is refused there instead of being decoded differently from the reference.
bool stream_too_long(adsb_decoder *d, size_t n)
{
    if (d->n_samples + (uint64_t)n < (1ull << 32))
        return false;
    d->fail("stream would reach 2^32 samples: the reference's sample counter wraps there (air.c:34) and no parity is defined beyond");
    return true;
}

int slot_reserve_device_tries(adsb_decoder *d, ScanSlot &s, size_t want_list, size_t want_tiles)
{
    if (want_list > s.d_try_cap || want_tiles > s.d_try_tiles) {
        want_list = std::max(want_list, s.d_try_cap);
        want_tiles = std::max(want_tiles, s.d_try_tiles);
        if (d->count_stream) { // a count pass may still be reading the old arrays
            if (count_flush(d))
                return -1;
            WAIT_STREAM(d, d->count_stream, "the try-count stream");
        }
        if (s.d_tries)
            HIP_TRY(d, hipFree(s.d_tries));
        if (s.d_try_counts)
            HIP_TRY(d, hipFree(s.d_try_counts));
        s.d_tries = s.d_try_counts = nullptr;
        s.d_try_cap = s.d_try_tiles = 0;
        HIP_TRY(d, hipMalloc(&s.d_tries, (want_tiles * adsb::kTryRegion + want_list) * sizeof(uint32_t)));
        HIP_TRY(d, hipMalloc(&s.d_try_counts, std::max<size_t>(want_tiles, 1) * sizeof(uint32_t)));
        s.d_try_cap = want_list;
        s.d_try_tiles = want_tiles;
    }
    return 0;
}

int slot_reserve(adsb_decoder *d, ScanSlot &s, size_t want_cands, size_t want_tries)
{
    if (want_cands > s.cand_cap) {
        if (s.cands)
            HIP_TRY(d, hipHostFree(s.cands));
        s.cands = nullptr;
        s.cand_cap = 0;
        HIP_TRY(d, hipHostMalloc(&s.cands, want_cands * adsb::kCandWords * sizeof(uint32_t), hipHostMallocDefault));
        s.cand_cap = want_cands;
    }
    if (want_tries > s.try_cap) {
        if (s.tries)
            HIP_TRY(d, hipHostFree(s.tries));
        s.tries = nullptr;
        s.try_cap = 0;
        HIP_TRY(d, hipHostMalloc(&s.tries, want_tries * sizeof(uint32_t), hipHostMallocDefault));
        s.try_cap = want_tries;
    }
    return 0;
}

int slot_reserve_hand(adsb_decoder *d, ScanSlot &s, size_t want_granules)
{
    // fine-grained (coherent) so that the host sees the device's stores while the
    // kernel is still running
    constexpr unsigned mem_flags = hipHostMallocCoherent;
    if (want_granules > s.hand_cap) {
        if (s.hand)
            HIP_TRY(d, hipHostFree(s.hand));
        s.hand = nullptr;
        s.hand_cap = 0;
        HIP_TRY(d, hipHostMalloc(&s.hand, want_granules * adsb::kGranuleWords * sizeof(uint32_t), mem_flags));
        s.hand_cap = want_granules;
    }
    return 0;
}

// Kernel time of a collected launch (cfg.profile): the tiles leave the earliest start and
// the latest end of the device's 100 MHz clock in the counters -- no events, no extended
// launch.  Read lazily (behind the slot's next launch, or in adsb_get_profile): waiting for
// the counters right after the last tile has been consumed would put a device round trip on
// the critical path of every push.
int slot_settle_profile(adsb_decoder *d, ScanSlot &s, int copy)
{
    if (!s.prof_pending[copy])
        return 0;
    s.prof_pending[copy] = false;
    WAIT_EVENT(d, s.ev_ready[copy], "a scan launch");
    const uint32_t *c = s.h_counters + copy * adsb::kCounterWords;
    const uint64_t t_begin = ~((uint64_t)c[5] << 32 | c[4]), t_end = (uint64_t)c[7] << 32 | c[6];
    const double ms = t_end > t_begin ? (double)(t_end - t_begin) * 1e-5 : 0.0; // 10 ns ticks
    d->prof.kernel_ms += ms;
    d->prof.last_kernel_ms = ms;
    const uint64_t no = s.ev_offsets[copy];
    if (no > d->prof.big_offsets) {
        d->prof.big_offsets = no;
        d->prof.big_launches = 0;
        d->prof.big_ms = 0;
    }
    if (no == d->prof.big_offsets) {
        d->prof.big_launches++;
        d->prof.big_ms += ms;
    }
    return 0;
}

int slot_launch(adsb_decoder *d, ScanSlot &s)
{
    const bool stats = d->cfg.collect_stats != 0;
    s.ev_cur ^= 1;
    if (slot_settle_profile(d, s, s.ev_cur)) // two launches old: normally read long ago
        return -1;
    s.ev_offsets[s.ev_cur] = s.args.g_end - s.args.g_begin;
    s.ntiles = adsb::tile_count(s.args.g_end - s.args.g_begin, s.args.big_tiles, s.args.passes);
    // A stream's statistics run keeps the try words on the device (counted there after
    // resolution); a per-shard scan hands the list back, sorted, so it needs the list
    // complete on the host: collect after completion.
    s.tries_on_device = stats && !d->sink.cands;
    s.streaming = !d->no_streaming && (!stats || s.tries_on_device);
    s.args.gen = ++d->launch_gen * 0x9E3779B9u + 0x7F4A7C15u;
    if (s.streaming) {
        // a line per tile (marker + padding) + two granules per record (sized like the loose list)
        if (slot_reserve_hand(d, s, std::max<size_t>(s.hand_cap, 2 * s.cand_cap + 4 * (size_t)s.ntiles + 64)))
            return -1;
        s.args.hand = s.hand;
        s.args.hand_cap = (uint32_t)std::min<size_t>(s.hand_cap, 0xFFFFFFFFu);
    } else {
        s.args.hand = nullptr;
        s.args.hand_cap = 0;
    }
    s.args.counters = s.d_counters;
    s.args.cands = s.cands;
    s.args.cand_cap = (uint32_t)std::min<size_t>(s.cand_cap, 0xFFFFFFFFu);
    if (d->pending.valid && d->pending.slot == &s && count_flush(d)) // (the pass that reads this slot's list is still to come)
        return -1;
    const int slot_index = (int)(&s - d->slots);
    hipStream_t ls = (d->alt_next && d->stream2 && (slot_index & 1)) ? d->stream2 : d->stream;
    if (s.launch_stream && s.launch_stream != ls) // the slot's previous launch (its report kernel zeroes the counters) ran on the other stream
        HIP_TRY(d, hipStreamWaitEvent(ls, s.ev_ready[s.ev_cur ^ 1], 0));
    s.launch_stream = ls;
    if (s.count_pending) { // the count pass over this slot's previous try list (count stream) must be over
        HIP_TRY(d, hipStreamWaitEvent(ls, s.ev_count, 0));
        s.count_pending = false;
    }
    // debug_try_cap (tests of the relaunch path) wants every try on the launch-wide list
    s.try_regions = s.tries_on_device && d->dbg.try_cap <= 0;
    if (s.try_regions && slot_reserve_device_tries(d, s, s.d_try_cap, s.ntiles))
        return -1;
    s.args.tries = s.tries_on_device ? s.d_tries : s.tries;
    s.args.try_cap = (uint32_t)std::min<size_t>(s.tries_on_device ? s.d_try_cap : s.try_cap, 0xFFFFFFFFu);
    s.args.try_counts = s.try_regions ? s.d_try_counts : nullptr;
    s.args.try_list_first = s.try_regions ? (uint32_t)(s.d_try_tiles * adsb::kTryRegion) : 0u;
    // d_counters are zero here: cleared at creation, and the report kernel behind every scan leaves them so
    s.args.profile = d->cfg.profile ? 1 : 0;
    s.args.report = s.hc();
    HIP_TRY(d, adsb::launch_scan(s.args, stats, ls));
    HIP_TRY(d, hipEventRecord(s.ev_ready[s.ev_cur], ls));
    if (stats && count_flush(d)) // the previous pass's calls are made now, while this scan runs
        return -1;
    for (ScanSlot &o : d->slots) // kernel times of earlier launches: read now, behind this launch
        for (int pair = 0; pair < 2; pair++)
            if (!(&o == &s && pair == s.ev_cur) && slot_settle_profile(d, o, pair))
                return -1;
    s.busy = true;
    return 0;
}

// LSD radix sort of the record indices by their 30-bit g_rel (3 x 10 bits).
void sort_order(adsb_decoder *d, const uint32_t *recs, size_t n)
{
    d->order.resize(n);
    d->scratch_a.resize(n);
    d->scratch_b.resize(n);
    uint32_t *idx = d->order.data(), *tmp = d->scratch_a.data(), *key = d->scratch_b.data();
    bool sorted = true;
    for (size_t i = 0; i < n; i++) {
        idx[i] = (uint32_t)i;
        key[i] = recs[i * adsb::kCandWords];
        if (i && key[i] < key[i - 1])
            sorted = false;
    }
    if (sorted)
        return;
    for (int shift = 0; shift < 30; shift += 10) {
        uint32_t hist[1025] = {0};
        for (size_t i = 0; i < n; i++)
            hist[((key[idx[i]] >> shift) & 1023u) + 1]++;
        for (int b = 0; b < 1024; b++)
            hist[b + 1] += hist[b];
        for (size_t i = 0; i < n; i++)
            tmp[hist[(key[idx[i]] >> shift) & 1023u]++] = idx[i];
        std::swap(idx, tmp);
    }
    if (idx != d->order.data())
        std::memcpy(d->order.data(), idx, n * sizeof(uint32_t)); // odd number of passes
}

void sort_tries(adsb_decoder *d, uint32_t *t, size_t n)
{
    bool sorted = true;
    for (size_t i = 1; i < n && sorted; i++)
        sorted = t[i] >= t[i - 1];
    if (sorted)
        return;
    d->scratch_a.resize(n);
    uint32_t *src = t, *dst = d->scratch_a.data();
    for (int shift = 0; shift < 32; shift += 11) {
        uint32_t hist[2049] = {0};
        for (size_t i = 0; i < n; i++)
            hist[((src[i] >> shift) & 2047u) + 1]++;
        for (int b = 0; b < 2048; b++)
            hist[b + 1] += hist[b];
        for (size_t i = 0; i < n; i++)
            dst[hist[(src[i] >> shift) & 2047u]++] = src[i];
        std::swap(src, dst);
    }
    if (src != t)
        std::memcpy(t, src, n * sizeof(uint32_t));
}

// Hand sorted records to the sink (a caller's vectors or the stream's resolver).
// Record i is the 6 dwords {g_rel, pw, frame | len << 16 | flags << 24} at
// recs[order[i] * words + off] (loose list / gathered copy: words 6, off 0; hand-off
// stream consumed in place: words 4 = granule index, off 1).
void deliver(adsb_decoder *d, const ScanSlot &s, const uint32_t *recs, const uint32_t *order, size_t nc, int words,
             int off, const uint32_t *tries, size_t nt, uint64_t g_complete)
{
    d->prof.candidates += nc;
    d->prof.tries += nt;
    if (d->sink.cands) {
        for (size_t i = 0; i < nc; i++) {
            const uint32_t *r = recs + (size_t)order[i] * words + off;
            adsb_candidate c;
            std::memset(&c, 0, sizeof c);
            std::memcpy(c.frame, &r[2], 14);
            c.len = (uint8_t)((r[5] >> 16) & 0xFF);
            c.reserved = (uint8_t)((r[5] >> 24) & 1u);
            // (a stream record may stand for the same frame at up to three consecutive offsets: scan_kernel_format.h)
            for (uint32_t k = 0, nk = words == adsb::kGranuleWords ? adsb::rec_copies(r) : 1u; k < nk; k++) {
                c.g = s.args.g_begin + r[0] + k;
                c.pw = k ? r[5 + k] : r[1];
                d->sink.cands->push_back(c);
            }
        }
        for (size_t i = 0; i < nt; i++)
            d->sink.tries->push_back((((uint64_t)(tries[i] >> 2) + s.args.g_begin) << 2) | (tries[i] & 3u));
    } else if (nt == 0) {
        d->res.capture_head(recs, order, nc, words, off, s.args.g_begin);
        d->res.advance_device(recs, order, nc, words, off, s.args.g_begin, power_samples_produced(d->n_samples),
                              g_complete);
    } else {
        d->res.feed_device(recs, order, nc, words, off, s.args.g_begin, tries, nt);
        d->res.advance(power_samples_produced(d->n_samples), g_complete);
    }
}

// (the reading side of the hand-off stream -- HandCursor, StreamReader, the two consumer loops -- is host-only code:
// handoff.hpp)

// "has the launch behind these bytes ended?" for handoff.hpp: ctx is the launch's completion event
int launch_done(void *ctx)
{
    const hipError_t q = hipEventQuery(static_cast<hipEvent_t>(ctx));
    return q == hipErrorNotReady ? 0 : q == hipSuccess ? 1 : -1;
}

adsb::HandJob hand_job(const ScanSlot &s)
{
    adsb::HandJob j;
    j.hand = s.hand;
    j.ntiles = s.ntiles;
    j.gen = s.args.gen;
    j.cap = s.args.hand_cap;
    j.done = launch_done;
    j.ctx = s.ev_ready[s.ev_cur];
    return j;
}

// Tiles [from, upto) of the launch's hand-off stream (d->tile_start / d->tile_count say where each one's records lie) go to
// the sink: a caller's vectors, or the stream's resolver, which walks the ranges where they lie.  Returns the records handed on.
size_t deliver_tiles(adsb_decoder *d, ScanSlot &s, uint32_t from, uint32_t upto)
{
    const uint32_t *t_start = d->tile_start.data(), *t_count = d->tile_count.data();
    const uint64_t g_complete = std::min<uint64_t>(
        s.args.g_end, s.args.g_begin + adsb::kRun * adsb::tile_first_run(upto, s.args.big_tiles, s.args.passes));
    size_t nc = 0;
    if (d->sink.cands) { // per-shard scan: the caller's vectors
        std::vector<uint32_t> &order = d->order; // the records in ascending g (granule indices)
        order.clear();
        for (uint32_t u = from; u < upto; u++)
            for (uint32_t i = 0, b = t_start[u], n = t_count[u]; i < n; i++)
                order.push_back(b + 2 * i);
        nc = order.size();
        deliver(d, s, s.hand, order.data(), nc, adsb::kGranuleWords, 0, nullptr, 0, g_complete);
    } else { // the stream's resolver walks the tile ranges where they lie
        for (uint32_t u = from; u < upto; u++)
            nc += t_count[u];
        d->prof.candidates += nc;
        if (d->res.head_wanted(s.args.g_begin + adsb::kRun * adsb::tile_first_run(from, s.args.big_tiles, s.args.passes)))
            d->res.capture_head_tiles(s.hand, t_start, t_count, from, upto, s.args.g_begin);
        d->res.advance_tiles(s.hand, t_start, t_count, from, upto, s.args.g_begin, power_samples_produced(d->n_samples), g_complete);
    }
    return nc;
}

// Streaming collect: consume the oldest scan WHILE its kernel is still running, so
// that resolving overlaps the scan.  The hand-off stream (scan_kernel.h) is read strictly
// sequentially -- one prefetchable stream of device-written lines, no directory to poll:
// a marker says which tile follows, how many records, and what their XOR must be; the
// records are checked where they lie (16-byte loads) and later resolved in place.  Tiles
// reserve their ranges in COMPLETION order, so a tile that finished early waits (start and
// count noted) until every tile before it is in; the resolver is fed whenever the device
// leaves the host nothing to read, or a group of tiles has accumulated.
// Returns 1 if a tile reported records on the loose list (or the stream is full): the
// caller then finishes the launch through the collect-after-completion path, from tile
// *resume_tile on.
// The handle's second host thread (handoff.hpp StreamReader), kept on the caller's L3.  cfg.host_threads = 2 starts it with the
// handle; 0 (auto) the first time a launch follows one that handed over kAutoReaderRecords or more -- at the channel's
// capacity one thread needs four times the kernel's time for a launch's records, and reading + checking on one thread while
// the caller resolves takes a quarter off that; under ordinary traffic the thread never exists.
constexpr uint64_t kAutoReaderRecords = 65536, kAutoReaderMinRecords = 16384;

// Did the previous launch of this handle hand over a record per 2 048 offsets or more (and 16 384 at least)?  The traffic of a
// channel does not change from one launch to the next: the host side starts its helper threads on this (slot_collect), and the
// next launch takes tiles of six passes instead of seven (adsb::choose_passes).
inline bool last_launch_was_dense(const adsb_decoder *d)
{
    const uint64_t dense_from = std::max<uint64_t>(kAutoReaderMinRecords, std::min<uint64_t>(kAutoReaderRecords, d->last_launch_offsets / 2048));
    return d->last_launch_records >= dense_from;
}
void start_reader(adsb_decoder *d)
{
    if (d->reader || d->reader_failed)
        return;
    d->reader = new (std::nothrow) adsb::StreamReader;
    if (d->reader) {
        d->reader->on_start = [](void *ctx) { (void)hipSetDevice(static_cast<adsb_decoder *>(ctx)->device); }; // launch_done()
        d->reader->on_start_ctx = d;
        try {
            d->reader->start();
        } catch (...) { // no thread to be had: the calling thread consumes the stream alone, as without the option
            delete d->reader;
            d->reader = nullptr;
        }
    }
    if (d->reader) {
        d->reader->place = true;
        d->reader->placed_l3 = adsb::place_reader_thread(d->reader->th, sched_getcpu());
    } else {
        d->reader_failed = true;
    }
}

// More hands for a channel at its capacity (gang.hpp): the calling thread decides, `helpers` threads on its L3 write the frames.
void start_gang(adsb_decoder *d, int helpers)
{
    if (d->gang || d->gang_failed)
        return;
    d->gang = new (std::nothrow) adsb::FormatGang;
    if (d->gang && !d->gang->start(helpers)) { // no thread to be had: the calling thread writes its frames itself, as without
        delete d->gang;
        d->gang = nullptr;
    }
    if (!d->gang) {
        d->gang_failed = true;
        return;
    }
    const int cpu = sched_getcpu();
    for (std::thread &t : d->gang->threads())
        d->gang_l3 = adsb::place_reader_thread(t, cpu);
}

constexpr int kAutoGangHelpers = 4; // (measured on the dense capture: profiles/r5_gang_runs.txt)

int slot_collect_streaming(adsb_decoder *d, ScanSlot &s, uint32_t *resume_tile, uint32_t *tiles_in, bool *tries_listed)
{
    using clk = std::chrono::steady_clock;
    const auto t_begin = clk::now();

    std::vector<uint32_t> &t_start = d->tile_start; // per tile: granule index of its first record ...
    std::vector<uint32_t> &t_count = d->tile_count; // ... and its record count (~0u: not in yet)
    t_start.assign(s.ntiles, 0u);
    t_count.assign(s.ntiles, ~0u);
    uint32_t delivered = 0; // every tile below has been handed to the resolver
    uint64_t recs_handed = 0;
    bool overflowed = false;
    double dbg[3] = {0, 0, 0};
    const bool dbg_on = tuning_env("ADSB_DEBUG_HOST") != nullptr;
    double wait_ms = 0;
    auto t_last_wait = t_begin;
    // With more hands (gang.hpp) a batch is handed to the resolver LATE: meanwhile one of the gang's threads decides it ahead
    // (Resolver::speculate_tiles), and the resolver only takes the decisions over.  A flush is cut into batches of kAheadTiles
    // tiles, so that several threads decide side by side and the last batch of a launch is a short one; a batch goes on
    // as soon as it has been decided (looked at with every flush), at the latest when kMaxHeld are waiting.
    constexpr int kMaxHeld = 12;
    constexpr uint32_t kAheadTiles = 64;
    uint32_t held[kMaxHeld][2];
    int n_held = 0;
    bool ahead = false;                    // (set below, once it is known whether this launch goes through the gang)
    auto deliver_held = [&](int keep, bool only_ready) { // the oldest first, until `keep` are left
        size_t nc = 0;
        int k = 0;
        for (; n_held - k > keep && (!only_ready || d->res.ahead_ready()); k++)
            nc += deliver_tiles(d, s, held[k][0], held[k][1]);
        for (int i = k; i < n_held; i++)
            held[i - k][0] = held[i][0], held[i - k][1] = held[i][1];
        n_held -= k;
        return nc;
    };
    auto flush = [&](uint32_t upto) { // tiles [delivered, upto): their ranges, one after the other, are sorted
        clk::time_point tp;
        if (dbg_on)
            tp = clk::now();
        size_t nc = 0;
        if (ahead) {
            for (uint32_t from = delivered; from < upto;) {
                const uint32_t to = std::min(upto, from + kAheadTiles);
                if (n_held == kMaxHeld)
                    nc += deliver_held(kMaxHeld - 1, false);
                if (d->res.speculate_tiles(s.hand, d->tile_start.data(), d->tile_count.data(), from, to, s.args.g_begin)) {
                    held[n_held][0] = from, held[n_held][1] = to;
                    n_held++;
                    d->prof.gang_batches++;
                } else { // (a batch too small to be worth it: in its turn, by this thread)
                    nc += deliver_held(0, false);
                    nc += deliver_tiles(d, s, from, to);
                }
                from = to;
            }
            nc += deliver_held(0, true);
        } else {
            nc += deliver_tiles(d, s, delivered, upto);
        }
        delivered = upto;
        recs_handed += nc;
        if (dbg_on) {
            dbg[1] += std::chrono::duration<double, std::micro>(clk::now() - tp).count();
            dbg[2] += 1;
            if (tuning_env("ADSB_DEBUG_TIMELINE"))
                fprintf(stderr, "  t=%.1f us: tiles < %u resolved (%zu records), waited %.1f us so far\n",
                        std::chrono::duration<double, std::micro>(clk::now() - t_begin).count(), upto, nc, wait_ms * 1e3);
        }
    };
    const adsb::HandJob job = hand_job(s);
    adsb::CollectEnd end;
    // "a channel near its capacity" is a DENSITY: 65 536 records out of a full launch's 128 Mi offsets = one per 2 048 (a full
    // channel has one per 1 090).  Round 6: a shorter launch -- a 128 Mi-sample shard of the multi-GPU driver is 64 Mi offsets,
    // 61 k records on a full channel -- counts by the same density, from 16 384 records on (below that a launch is resolved
    // faster than five threads are woken).
    const bool after_dense = d->cfg.host_threads == 0 && last_launch_was_dense(d);
    if (after_dense) {
        start_reader(d);
        cpu_set_t allowed; // (six threads that poll need cores of their own: on a small or confined host, round 4's pair)
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0 || CPU_COUNT(&allowed) >= 2 * (kAutoGangHelpers + 2))
            start_gang(d, kAutoGangHelpers);
    }
    // (a batch decided ahead packs an offset relative to the launch's first into 31 bits: chunk_offsets() keeps a launch below
    // 2^30 offsets -- kMaxLaunchOffsets -- and this says so where it matters)
    const bool with_gang = d->gang && !d->sink.cands && (uint64_t)s.args.hand_cap * adsb::kGranuleWords * 4 <= adsb::kDecMaxStreamBytes &&
                           s.args.g_end - s.args.g_begin < (1ull << 31) && (d->cfg.host_threads >= 3 || after_dense);
    if (d->dbg.gang_min > 0) {
        d->res.set_gang(with_gang ? d->gang : nullptr, (size_t)d->dbg.gang_min);
        d->res.set_ahead_min_records((size_t)d->dbg.gang_min);
    } else {
        d->res.set_gang(with_gang ? d->gang : nullptr);
    }
    if (with_gang) {
        const int cpu = sched_getcpu(); // the caller may have moved since the threads were placed
        const int l3 = adsb::l3_of_cpu(cpu);
        if (l3 >= 0 && l3 != d->gang_l3)
            for (std::thread &t : d->gang->threads())
                d->gang_l3 = adsb::place_reader_thread(t, cpu);
        d->gang->begin();
        ahead = true;
        d->prof.gang_launches++;
    }
    if (d->reader && s.ntiles >= d->reader_min_tiles && (d->cfg.host_threads >= 2 || after_dense)) {
        adsb::StreamReader &rd = *d->reader;
        if (rd.place) { // the caller may have moved since the thread was placed
            const int cpu = sched_getcpu();
            const int l3 = adsb::l3_of_cpu(cpu);
            if (l3 >= 0 && l3 != rd.placed_l3)
                rd.placed_l3 = adsb::place_reader_thread(rd.th, cpu);
        }
        auto idle = [&]() -> bool { // batches that have been decided meanwhile go on while the device is behind
            if (!n_held || !d->res.ahead_ready())
                return false;
            clk::time_point tp;
            if (dbg_on)
                tp = clk::now();
            recs_handed += deliver_held(0, true);
            if (dbg_on)
                dbg[1] += std::chrono::duration<double, std::micro>(clk::now() - tp).count();
            return true;
        };
        end = adsb::collect_behind_reader(rd, job, t_start.data(), t_count.data(), delivered, flush, wait_ms, t_last_wait, idle);
        if (dbg_on)
            fprintf(stderr, "stream reader thread: busy %.1f us, waits %.1f us\n", rd.busy_ms * 1e3, rd.wait_ms * 1e3);
    } else {
        end = adsb::collect_alone(job, t_start.data(), t_count.data(), delivered, flush, wait_ms, t_last_wait);
    }
    if (n_held) { // the batches that were still waiting for their turn
        const auto tp = clk::now();
        recs_handed += deliver_held(0, false);
        if (dbg_on)
            dbg[1] += std::chrono::duration<double, std::micro>(clk::now() - tp).count();
    }
    if (end.status < 0 && with_gang) {
        d->res.sync();
        d->gang->end();
    }
    if (end.status == -1)
        return d->fail("hand-off stream corrupt at granule %u (tile %u twice)", end.pos, end.tile);
    if (end.status == -2)
        return d->fail("scan kernel finished without publishing granule %u (tile %u of %u pending)", end.pos, end.tile, s.ntiles);
    overflowed = end.status == 1;
    if (with_gang) { // the frames of this launch are whole before its stream is touched again (and before anyone counts the time)
        d->res.sync();
        d->gang->end();
    }
    if (dbg_on)
        fprintf(stderr, "gang: %s, ahead %d, frames taken over so far %llu\n", with_gang ? "on" : "off", (int)ahead,
                (unsigned long long)d->res.ahead_taken());
    if (dbg_on)
        fprintf(stderr,
                "stream collect: %.1f us in all, resolve %.1f us in %d batches, waits %.1f us; %.1f us after the last wait\n",
                std::chrono::duration<double, std::micro>(clk::now() - t_begin).count(), dbg[1], (int)dbg[2], wait_ms * 1e3,
                std::chrono::duration<double, std::micro>(clk::now() - t_last_wait).count());
    const double total_ms = std::chrono::duration<double, std::milli>(clk::now() - t_begin).count();
    d->prof.wait_ms += wait_ms;
    d->prof.host_ms += total_ms - wait_ms;
    *resume_tile = delivered;
    *tiles_in = end.frontier;
    *tries_listed = end.tries_listed;
    d->last_launch_records = recs_handed; // (a launch that is finished after completion adds its part there)
    d->last_launch_offsets = s.args.g_end - s.args.g_begin;
    return overflowed ? 1 : 0;
}

// Device-side visited-try count of a statistics run (scan_kernel.h TryCountArgs):
// decides every try below the resolver's position against the frames it accepted
// since the previous pass (plus the last one before, whose span may reach further),
// adds three counters to the statistics and carries the undecided tries.
// Everything here goes to the decoder's COUNT stream: the upload of the accepted frames and the count kernel
// (~40 us of device time together) run beside the next scan instead of in front of it.  Nothing on the scan
// stream depends on them except the reuse of the slot's try list, four launches later (ev_count); in the other
// direction the count stream waits for the end of the scan that wrote the list (count_flush).
// Enqueue the prepared count pass, if any (and the clearing an adsb_reset has queued behind it).
int count_flush(adsb_decoder *d)
{
    auto &p = d->pending;
    hipStream_t cs = d->count_stream;
    if (p.valid) {
        p.valid = false;
        if (p.nf) {
            HIP_TRY(d, hipMemcpyAsync(d->d_frames, p.src, p.nf * sizeof(adsb::TryFrame), hipMemcpyHostToDevice, cs));
            HIP_TRY(d, hipEventRecord(d->ev_frames[p.b], cs));
            d->frames_pending[p.b] = true;
        }
        // The host may have taken the launch's last tile before the scan kernel has ended (slot_collect does not
        // wait for the launch counters when no tile needed them): the try words become visible to other kernels
        // with the kernel's end, so the count stream waits for it.
        if (p.after)
            HIP_TRY(d, hipStreamWaitEvent(cs, p.after, 0));
        HIP_TRY(d, adsb::launch_count_tries(p.a, cs)); // enqueued and forgotten: read_tries() collects
        if (p.slot) {
            HIP_TRY(d, hipEventRecord(p.slot->ev_count, cs));
            p.slot->count_pending = true;
        }
    }
    if (p.clear_after) {
        p.clear_after = false;
        HIP_TRY(d, hipMemsetAsync(d->d_try_acc, 0, kTryStateBytes, cs));
    }
    return 0;
}

int count_tries_pass(adsb_decoder *d, ScanSlot *slot, uint32_t n_tries, uint64_t g_base, bool final)
{
    hipStream_t cs = d->count_stream;
    if (count_flush(d)) // one pass pending at a time, in order
        return -1;
    const bool regions = slot && slot->try_regions;
    static_assert(sizeof(adsb::Resolver::LogEntry) == sizeof(adsb::TryFrame), "the resolver logs TryFrame records in place");
    auto &over = d->res.accepted_log(); // entries that did not fit the pinned buffer (normally none)
    const size_t n_ext = d->res.logged_ext(), n_log = n_ext + over.size();
    const size_t nf = n_log + (d->have_prev_frame ? 1 : 0);
    int b = d->log_buf;
    const bool had_prev = d->have_prev_frame;
    const uint64_t prev_g = d->prev_frame_g;
    const uint32_t prev_span = d->prev_frame_span;
    if (n_log) { // the last accepted frame: its span may cover tries of the next pass
        d->have_prev_frame = true;
        if (!over.empty()) {
            d->prev_frame_g = over.back().first;
            d->prev_frame_span = over.back().second;
        } else {
            d->prev_frame_g = d->h_frames[b][n_ext].g; // entries sit at [1 .. n_ext]
            d->prev_frame_span = d->h_frames[b][n_ext].span;
        }
    }
    if (n_tries == 0 && !regions && !d->carry_maybe) {
        d->res.log_clear();
        return 0;
    }
    if (!over.empty() || nf > d->frames_cap) { // rare: grow the frame arrays (passes in flight use them: drain the stream first)
        WAIT_STREAM(d, cs, "the try-count stream");
        const size_t cap = std::max<size_t>(nf + nf / 4 + 1, 2 * d->frames_cap);
        adsb::TryFrame *nh[adsb_decoder::kFrameBufs] = {nullptr, nullptr, nullptr};
        for (int i = 0; i < adsb_decoder::kFrameBufs; i++)
            HIP_TRY(d, hipHostMalloc(&nh[i], cap * sizeof(adsb::TryFrame), hipHostMallocDefault));
        if (n_ext)
            std::memcpy(nh[b] + 1, d->h_frames[b] + 1, n_ext * sizeof(adsb::TryFrame));
        size_t k = 1 + n_ext;
        for (const auto &f : over)
            nh[b][k++] = adsb::TryFrame{f.first, f.second, 0};
        for (int i = 0; i < adsb_decoder::kFrameBufs; i++) {
            if (d->h_frames[i]) HIP_TRY(d, hipHostFree(d->h_frames[i]));
            d->h_frames[i] = nh[i];
            d->frames_pending[i] = false;
        }
        if (d->d_frames) HIP_TRY(d, hipFree(d->d_frames));
        d->d_frames = nullptr;
        HIP_TRY(d, hipMalloc(&d->d_frames, cap * sizeof(adsb::TryFrame)));
        d->frames_cap = cap;
    }
    const adsb::TryFrame *src = d->h_frames[b] + 1;
    if (had_prev) {
        d->h_frames[b][0] = adsb::TryFrame{prev_g, prev_span, 0};
        src = d->h_frames[b];
    }
    {   // the resolver goes on logging into the buffer of the pass before last
        const int nb = (b + 1) % adsb_decoder::kFrameBufs;
        if (d->frames_pending[nb]) {
            WAIT_EVENT(d, d->ev_frames[nb], "the upload of the accepted frames");
            d->frames_pending[nb] = false;
        }
        d->log_buf = nb;
        d->res.log_into(reinterpret_cast<adsb::Resolver::LogEntry *>(d->h_frames[nb] + 1), d->frames_cap - 1);
    }
    const int c_in = d->carry_n_cur, c_out = (c_in + 1) % 3, c_next = (c_in + 2) % 3;
    adsb::TryCountArgs a{};
    a.tries = slot ? slot->d_tries + slot->args.try_list_first : nullptr;
    a.n_tries = n_tries;
    a.regions = regions ? slot->d_tries : nullptr;
    a.region_counts = regions ? slot->d_try_counts : nullptr;
    a.n_tiles = regions ? slot->ntiles : 0;
    a.passes = slot ? slot->args.passes : 0;
    a.big_tiles = slot ? slot->args.big_tiles : 0;
    a.g_base = g_base;
    a.carry_in = d->d_carry[d->carry_cur];
    a.n_carry = d->d_carry_n + c_in;
    a.frames = d->d_frames;
    a.n_frames = (uint32_t)nf;
    a.hi = d->res.base(); // every offset below has been visited or jumped over
    a.final = final ? 1 : 0;
    a.carry_out = d->d_carry[d->carry_cur ^ 1];
    a.carry_cap = kCarryCap;
    a.n_carry_out = d->d_carry_n + c_out; // zero: cleared at creation / reset, or by the pass before last
    a.n_carry_next = d->d_carry_n + c_next;
    a.acc = d->d_try_acc;
    d->pending.valid = true;
    d->pending.a = a;
    d->pending.nf = nf;
    d->pending.b = b;
    d->pending.src = src;
    d->pending.slot = (slot && (n_tries || regions)) ? slot : nullptr;
    d->pending.after = d->pending.slot ? slot->ev_ready[slot->ev_cur] : nullptr;
    d->prof.tries += n_tries;
    d->carry_cur ^= 1;
    d->carry_n_cur = c_out;
    d->carry_maybe = !final;
    d->tries_unread = true;
    d->acc_dirty = true;
    return 0;
}

// The statistics are asked for: wait for the count passes and take the device's totals.
int read_tries(adsb_decoder *d)
{
    if (!d->tries_unread)
        return 0;
    if (count_flush(d))
        return -1;
    unsigned long long acc[4];
    HIP_TRY(d, hipMemcpyAsync(acc, d->d_try_acc, sizeof acc, hipMemcpyDeviceToHost, d->count_stream));
    WAIT_STREAM(d, d->count_stream, "the try-count stream");
    d->tries_unread = false;
    if (acc[3])
        return d->fail("undecided tries exceeded the carry buffer (%u entries)", kCarryCap);
    d->res.set_tries(acc[0], acc[1], acc[2]);
    return 0;
}

// Wait for the oldest scan in flight and hand its records on, in ascending g.
int slot_collect(adsb_decoder *d)
{
    ScanSlot &s = d->slots[d->slot_head];
    using clk = std::chrono::steady_clock;
    uint32_t resume_tile = 0, tiles_in = 0;
    bool partial = false; // tiles below resume_tile were already delivered
    bool relaunched = false;
    bool tries_listed = false; // (statistics runs) some tile's tries are on the launch-wide list: its length comes with the counters
    if (s.streaming) {
        const int rc = slot_collect_streaming(d, s, &resume_tile, &tiles_in, &tries_listed);
        if (rc < 0)
            return -1;
        partial = rc == 1;
    }
    if (s.streaming && !partial && !tries_listed && (!s.tries_on_device || s.try_regions)) {
        // Every tile has been published and consumed and none used the loose list -- nor, in a statistics
        // run, the launch-wide try list: a tile that overflows its survivor queue says so in its marker (kMarkTries:
        // its records are in the stream and have been handed on like any other's), and the tries of all others are
        // in their regions.  The launch-wide counters have nothing to add, so
        // do not wait for them (nor for the kernel's end event -- the profile reads that later).
        s.prof_pending[s.ev_cur] = d->cfg.profile != 0;
        d->prof.launches++;
        d->prof.offsets += s.args.g_end - s.args.g_begin;
        d->prof.last_offsets = s.args.g_end - s.args.g_begin;
        if (s.tries_on_device) {
            if (d->final_follows && d->slot_count == 1) { // (see below)
                d->deferred_slot = &s;
                d->deferred_n = 0;
                d->deferred_base = s.args.g_begin;
            } else if (count_tries_pass(d, &s, 0, s.args.g_begin, false)) {
                return -1;
            }
        }
        s.busy = false;
        d->slot_head = (d->slot_head + 1) % kSlots;
        d->slot_count--;
        return 0;
    }
    const auto t_wait = clk::now();
    for (int attempt = 0;; attempt++) {
        if (s.streaming && !partial) {
            // every tile has been consumed: the kernel is ending and its report is microseconds
            // away -- poll for it instead of going to sleep in hipEventSynchronize
            WAIT_EVENT(d, s.ev_ready[s.ev_cur], "the end of a scan launch whose every tile has been consumed");
        } else {
            WAIT_EVENT(d, s.ev_ready[s.ev_cur], "a scan launch");
        }
        s.prof_pending[s.ev_cur] = d->cfg.profile != 0;
        if (slot_settle_profile(d, s, s.ev_cur))
            return -1;
        d->prof.launches++;
        d->prof.offsets += s.args.g_end - s.args.g_begin;
        d->prof.last_offsets = s.args.g_end - s.args.g_begin;
        const size_t nc = s.hc()[0], nt = s.hc()[1];
        if (nc <= s.cand_cap && nt <= (s.tries_on_device ? s.d_try_cap : s.try_cap))
            break;
        // Sparse output sized for far more than noise produces; the counters keep
        // counting past the capacity, so one repeat with exact sizes suffices.
        if (attempt >= 2)
            return d->fail("record buffers overflowed repeatedly (%zu candidates, %zu tries)", nc, nt);
        d->prof.relaunches++;
        relaunched = true;
        WAIT_STREAM(d, s.launch_stream ? s.launch_stream : d->stream, "the launch's stream");
        if (slot_reserve(d, s, std::max(s.cand_cap, nc + nc / 8 + 64),
                         s.tries_on_device ? s.try_cap : std::max(s.try_cap, nt + nt / 8 + 64)))
            return -1;
        if (s.tries_on_device && slot_reserve_device_tries(d, s, std::max(s.d_try_cap, nt + nt / 8 + 64), s.d_try_tiles))
            return -1;
        if (slot_launch(d, s))
            return -1;
        // the repeat is consumed after completion: tiles below resume_tile (if any)
        // were delivered by the first run and are skipped by the gather below
    }
    const auto t_host = clk::now();
    d->prof.wait_ms += std::chrono::duration<double, std::milli>(t_host - t_wait).count();
    const size_t nc = s.hc()[0], nt = s.hc()[1];
    if (!s.streaming) {
        // collect-after-completion: everything is in the launch-wide lists, in arrival order
        sort_order(d, s.cands, nc);
        if (nt && !s.tries_on_device)
            sort_tries(d, s.tries, nt);
        const size_t nt_host = s.tries_on_device ? 0 : nt;
        deliver(d, s, s.cands, d->order.data(), nc, adsb::kCandWords, 0, s.tries, nt_host, s.args.g_end);
    } else if (partial) {
        // Some tile could not put all its records into the hand-off stream (staged list or survivor queue overflowed, its
        // range did not fit): those records are on the loose list, which is only complete now that the kernel has ended.
        // Every granule that was ever written is in: walk the stream again from its start (a missing or non-fitting marker
        // ends it), note where each tile's records lie, sort the LOOSE records (few) and hand the tiles on in order -- runs of
        // tiles that are whole in the stream where they lie, like the streaming collect does; a tile with loose records, or
        // none in the stream at all, merged on the way.  (Round 3 gathered and sorted everything that was left: 4 ms for
        // the 311 k records of a dense launch in which ONE early tile had overflowed.)
        std::vector<uint32_t> &t_start = d->tile_start, &t_count = d->tile_count;
        // (the streaming collect went on reading and checking behind the first tile that held it up: when it got to the end
        // of the launch, where every tile's records lie is known already)
        d->last_launch_records = std::max<uint64_t>(d->last_launch_records, s.hc()[2] / 2); // (an estimate from the granules the stream used)
        const bool walked = tiles_in == s.ntiles && !relaunched;
        if (!walked) {
            t_start.assign(s.ntiles, 0u);
            t_count.assign(s.ntiles, ~0u);
        }
        const uint32_t lim = walked ? 0u : (uint32_t)std::min<size_t>(s.hc()[2], s.args.hand_cap);
        for (uint32_t pos = 0; pos < lim;) {
            const uint32_t *m = s.hand + (size_t)pos * adsb::kGranuleWords;
            const uint32_t tile = m[0], nf = m[1], n = nf & 0xFFFFu;
            if (tile >= s.ntiles || (nf & adsb::kMarkNoFit) || (uint64_t)pos + 1 + 2ull * n > lim || t_count[tile] != ~0u)
                break;
            uint32_t a[4] = {0, 0, 0, 0}, sum = 0, lo, hi;
            for (uint32_t k = 0; k < 8 * n; k++)
                a[k & 3] ^= m[4 + k];
            for (uint32_t r = 0; r < n; r++)
                sum += adsb::record_term(r, m[4 + 8 * r], m[5 + 8 * r]);
            adsb::marker_check(tile, nf, s.args.gen, a[0], a[1], a[2], a[3], sum, lo, hi);
            if (m[2] != lo || m[3] != hi)
                break;
            t_start[tile] = pos + 1;
            t_count[tile] = n;
            pos += std::max(adsb::marker_granules(nf), adsb::stream_granules(n));
        }
        // The loose list may also hold records of tiles the streamed part has already delivered: after a relaunch (record
        // buffers regrown) every tile runs again, and whether a tile's range fits the stream depends on completion order.
        const uint64_t resume_rel = (uint64_t)adsb::kRun * adsb::tile_first_run(resume_tile, s.args.big_tiles, s.args.passes);
        d->gather.clear();
        for (size_t i = 0; i < nc; i++) {
            const uint32_t *w = s.cands + i * adsb::kCandWords;
            if (w[0] >= resume_rel)
                d->gather.insert(d->gather.end(), w, w + adsb::kCandWords);
        }
        const size_t n_loose = d->gather.size() / adsb::kCandWords;
        sort_order(d, d->gather.data(), n_loose);
        const std::vector<uint32_t> loose_order(d->order.begin(), d->order.begin() + (ptrdiff_t)n_loose); // (deliver() reuses d->order)
        std::vector<uint32_t> &merged = d->scratch_b; // one tile's records, kCandWords each, ascending
        std::vector<uint32_t> iota;
        size_t li = 0;
        uint32_t run_from = resume_tile;
        for (uint32_t u = resume_tile; u < s.ntiles; u++) {
            const uint64_t hi_rel = (uint64_t)adsb::kRun * adsb::tile_first_run(u + 1, s.args.big_tiles, s.args.passes);
            size_t lj = li;
            while (lj < n_loose && d->gather[(size_t)loose_order[lj] * adsb::kCandWords] < hi_rel)
                lj++;
            if (t_count[u] != ~0u && lj == li)
                continue; // whole in the stream: part of the current run
            if (u > run_from)
                deliver_tiles(d, s, run_from, u);
            // this tile: its records in the stream (if it got that far) merged with its loose ones, both ascending
            merged.clear();
            const uint32_t ns = t_count[u] == ~0u ? 0u : t_count[u];
            const uint32_t *sr = s.hand + (size_t)t_start[u] * adsb::kGranuleWords;
            uint32_t si = 0;
            while (si < ns || li < lj) {
                const uint32_t *lw = li < lj ? d->gather.data() + (size_t)loose_order[li] * adsb::kCandWords : nullptr;
                const uint32_t *sw = si < ns ? sr + (size_t)si * 2 * adsb::kGranuleWords : nullptr;
                if (sw && (!lw || sw[0] <= lw[0])) {
                    // {g_rel, pw, w0, w1}{w2, w3, pw', pw''}: the first six words, once per offset the record stands for (a run
                    // of copies is never interleaved with a loose record: the offsets are consecutive and every offset yields
                    // at most one candidate -- but a loose one may lie INSIDE the run only if it is one of its offsets, which
                    // the tile would have staged with the others; so the run goes in whole)
                    for (uint32_t k = 0, nk = adsb::rec_copies(sw); k < nk; k++) {
                        const uint32_t one[adsb::kCandWords] = {sw[0] + k, k ? sw[5 + k] : sw[1], sw[2], sw[3], sw[4],
                                                                 sw[5] & ~(3u << adsb::kRecCopiesShift)};
                        merged.insert(merged.end(), one, one + adsb::kCandWords);
                    }
                    si++;
                } else {
                    merged.insert(merged.end(), lw, lw + adsb::kCandWords);
                    li++;
                }
            }
            const size_t nm = merged.size() / adsb::kCandWords;
            iota.resize(nm);
            for (size_t i = 0; i < nm; i++)
                iota[i] = (uint32_t)i;
            const uint64_t g_complete = std::min<uint64_t>(s.args.g_end, s.args.g_begin + hi_rel);
            deliver(d, s, merged.data(), iota.data(), nm, adsb::kCandWords, 0, nullptr, 0, g_complete);
            run_from = u + 1;
        }
        if (s.ntiles > run_from)
            deliver_tiles(d, s, run_from, s.ntiles);
    } else if (nc != 0) {
        return d->fail("internal: %zu loose records without a tile overflow flag", nc);
    }
    if (s.tries_on_device) {
        if (d->final_follows && d->slot_count == 1) {
            // last launch of the stream: its tries are counted by the end-of-stream pass, which
            // runs right after the final resolver step -- one device round trip instead of two
            d->deferred_slot = &s;
            d->deferred_n = (uint32_t)nt;
            d->deferred_base = s.args.g_begin;
        } else if (count_tries_pass(d, &s, (uint32_t)nt, s.args.g_begin, false)) {
            return -1;
        }
    }
    d->res.sync(); // (tiles handed on after completion may have gone to the gang as well)
    if (d->gang)
        d->gang->end(); // ... and FormatGang::post() begins the gang again by itself: without this the helpers would poll on until
                        // the next launch that goes through the gang -- under traffic that has turned sparse, until adsb_destroy
    d->prof.host_ms += std::chrono::duration<double, std::milli>(clk::now() - t_host).count();
    s.busy = false;
    d->slot_head = (d->slot_head + 1) % kSlots;
    d->slot_count--;
    return 0;
}

int scan_drain(adsb_decoder *d)
{
    while (d->slot_count)
        if (slot_collect(d))
            return -1;
    return 0;
}

// Submit offsets [g_begin, g_end) of a device buffer holding stream samples
// [buf_first, buf_first + buf_n) (buf_first % 8 == 0, buf 16-byte aligned) as a
// pipeline of chunked launches: while the device scans chunk k+1 the host sorts
// and resolves chunk k.  Records reach the sink in ascending g.
int scan_submit(adsb_decoder *d, const uint16_t *buf, uint64_t buf_first, uint64_t buf_n, uint64_t g_begin,
                uint64_t g_end)
{
    const bool stats = d->cfg.collect_stats != 0;
    while (g_begin < g_end) {
        // streamed launches: everything but a per-shard scan that hands the try list back
        const uint64_t g_stop = std::min(g_end, g_begin + chunk_offsets(!d->no_streaming && !(stats && d->sink.cands)));
        const uint64_t n_off = g_stop - g_begin;
        if (d->slot_count == kSlots && slot_collect(d))
            return -1;
        ScanSlot &s = d->slots[(d->slot_head + d->slot_count) % kSlots];
        const bool host_tries = stats && d->sink.cands; // per-shard scans return the try list
        // test knobs: start from buffers that are too small, so that the relaunch path runs
        const size_t cand_want = d->dbg.cand_cap > 0 ? (size_t)d->dbg.cand_cap : (size_t)(n_off / 128 + 32768);
        const size_t try_want = d->dbg.try_cap > 0 ? (size_t)d->dbg.try_cap : (size_t)(n_off / 32 + 65536);
        if (slot_reserve(d, s, std::max<size_t>(s.cand_cap, cand_want),
                         host_tries ? std::max<size_t>(s.try_cap, try_want) : s.try_cap))
            return -1;
        if (stats && !host_tries && slot_reserve_device_tries(d, s, std::max<size_t>(s.d_try_cap, try_want), s.d_try_tiles))
            return -1;
        adsb::ScanArgs &a = s.args;
        a = adsb::ScanArgs{};
        a.x = reinterpret_cast<const uint32_t *>(buf);
        a.pbuf0 = (int64_t)(buf_first / 2);
        a.p_lo = a.pbuf0; // stream start: pairs below 0 read as silence (air.c:33)
        a.p_hi = a.pbuf0 + (int64_t)(buf_n / 2);
        a.g_begin = g_begin;
        a.g_end = g_stop;
        a.df18 = d->cfg.df18 ? 1 : 0;
        a.passes = (d->dbg.passes >= 2 && d->dbg.passes <= adsb::kMaxPasses) ? d->dbg.passes
                                                                                         : adsb::choose_passes(n_off, d->n_cus, last_launch_was_dense(d));
        a.big_tiles = adsb::choose_big_tiles(n_off, a.passes, d->n_cus, d->dbg.big_tiles);
        a.synd = d->d_synd;
        a.queue_cap = (d->dbg.queue_cap >= 256 && d->dbg.queue_cap <= adsb::kQueueCap)
                          ? d->dbg.queue_cap
                          : adsb::kQueueCap;
        a.all_candidates = d->cfg.all_candidates ? 1 : 0;
        a.clist_cap = (d->dbg.clist_cap >= 1 && d->dbg.clist_cap <= adsb::kClistCap) ? d->dbg.clist_cap
                                                                                                  : adsb::kClistCap;
        a.fix_tab = d->cfg.fix_1bit ? d->d_fix : nullptr;
        a.fix_mul = d->fix_mul;
        if (slot_launch(d, s))
            return -1;
        s.piece = d->piece;
        d->slot_count++;
        g_begin = g_stop;
    }
    return 0;
}

// Where the kept tail of the staging buffer starts: at or below `want` (a multiple of 8 samples), lowered by up
// to 56 samples so that the tail's END -- where the next push is appended -- falls on a 128-byte line whenever
// the stream position allows it (n_samples % 8 == 0): pieces of adsb_push_async that start on a line run on
// alternating copy streams without waiting for each other (push_copy).
uint64_t line_aligned_keep(uint64_t want, uint64_t n_samples, uint64_t floor_first)
{
    if (n_samples % 8 != 0 || want > n_samples)
        return want;
    const uint64_t extra = (64 - (n_samples - want) % 64) % 64; // samples; a multiple of 8
    return want >= floor_first + extra ? want - extra : want;
}

// Scan what the staged samples allow, resolve, and keep only the unscanned tail.
// in_flight (adsb_push_async): the launches submitted here are left running; only those
// of earlier pieces are collected.
int process_stage(adsb_decoder *d, bool final, bool in_flight = false)
{
    const uint64_t m_real = power_samples_produced(d->n_samples);
    const uint64_t g_end = scannable_end(d, d->n_samples, final);
    bool launched = false;
    if (g_end > d->g_scanned) {
        if (scan_submit(d, d->stage[d->cur], d->stage_first, d->stage_fill, d->g_scanned, g_end))
            return -1;
        d->g_scanned = g_end;
        launched = true;
    }
    if (in_flight) {
        while (d->slot_count && d->slots[d->slot_head].piece < d->piece)
            if (slot_collect(d))
                return -1;
    } else {
        if (scan_drain(d)) // frames become drainable within the call that supplied their samples
            return -1;
        if (launched) // its tiles have all been taken: the scan, and with it every copy queued in front of it, is over
            d->copy_unconfirmed = false;
    }
    // At EOF a trailing partial quad still makes the reference produce two (garbage)
    // power samples (air.c:59 loop bound); they can never be read by a visited
    // offset but they count for the `aidx >= APBUFFSZ` test.
    const uint64_t m_ref = final ? 2 * ((d->n_samples + 3) / 4) : m_real;
    if (!in_flight) // (in flight: the records below g_scanned are not all in yet; slot_collect advanced as far as they are)
        d->res.advance(m_ref, d->g_scanned);
    if (final)
        return 0;

    // Keep samples from pair (g_scanned - 8) on; that index is a multiple of 8 samples.  The
    // scanned part is only dropped (the tail moved to the other buffer) once the buffer is
    // half full: until then the next push is appended behind what is there and the next scan
    // reads [tail | new] where it lies -- small pushes (the reference's 1 Mi-sample calls)
    // then cost no device-to-device copy at all.
    const uint64_t keep_first = line_aligned_keep(d->g_scanned >= 8 ? 2 * (d->g_scanned - 8) : 0, d->n_samples, d->stage_first);
    if (keep_first > d->stage_first && d->stage_fill > (d->stage_cap - kStageSlack) / 2) {
        const uint64_t skip = keep_first - d->stage_first;
        const uint64_t left = d->stage_fill > skip ? d->stage_fill - skip : 0;
        if (left) {
            // The next asynchronous piece is copied right behind this tail by a copy engine on another stream,
            // and the two ranges meet inside a cache line (when `left` is not line-aligned): unordered, the tail
            // copy's write-back of that line and the engine's write to it race, and the loser's bytes are lost
            // (observed: one frame straddling the seam missing in 5-35 % of the runs with a 64 Ki staging buffer,
            // which compacts at every piece; tools/async_race.py).  So every later copy waits for the tail copy.
            // The tail copy itself needs this piece's copy and nothing else -- the buffer it writes was last read
            // by launches that have been collected, and it only reads the current one -- so in asynchronous mode
            // it follows that copy on ITS stream instead of queueing behind this piece's scan: the bubble per
            // compaction is the tail copy (a few KB), not a scan.
            const bool aside = in_flight && d->dbg_async != 4 && d->dbg_async != 2;
            hipStream_t ts = aside ? d->copy_stream[d->piece % adsb_decoder::kCopyStreams] : d->stream;
            // The tail also holds the end of the PREVIOUS piece whenever this piece is shorter than the tail
            // (~2.5 K samples), and that piece was copied on the other copy stream: order behind it too (an event
            // that has already completed costs nothing).
            if (aside && d->piece > 1)
                HIP_TRY(d, hipStreamWaitEvent(ts, d->ev_copy[(d->piece - 1) % adsb_decoder::kCopyStreams], 0));
            HIP_TRY(d, adsb::launch_copy_samples(d->stage[d->cur ^ 1], d->stage[d->cur] + skip, left, ts)); // (the library's own kernel:
            //                                the runtime's first device-to-device copy of a process costs 7 ms, scan_kernel.hip)
            // Behind EVERY tail copy, the one of a synchronous push on the scan stream included: a following
            // adsb_push_async copies right behind this tail on a copy stream that nothing else orders against it.
            if (d->dbg_async != 4) {
                HIP_TRY(d, hipEventRecord(d->ev_tail, ts));
                if (ts != d->stream)
                    HIP_TRY(d, hipStreamWaitEvent(d->stream, d->ev_tail, 0));
                for (hipStream_t cs : d->copy_stream)
                    if (cs != ts)
                        HIP_TRY(d, hipStreamWaitEvent(cs, d->ev_tail, 0));
            }
        }
        d->cur ^= 1;
        d->stage_first = keep_first;
        d->stage_fill = left;
    }
    return 0;
}

// async (adsb_push_async): the copy of a piece goes to the copy stream and the scan stream
// waits for it by event, so that the copy engine moves piece k+1 while piece k is scanned;
// the launches of a piece are collected while the NEXT piece is on its way.  Why no other
// synchronisation is needed: piece k is copied behind the unscanned tail of the staging
// buffer that becomes current after piece k-1's tail copy, [left, left + take) -- a region
// that the tail copy (it writes [0, left)) does not touch, that scan k-1 does not read (it
// reads the other buffer, or this one below `left`), and whose previous reader, scan k-2,
// was collected while piece k-1 was pushed.
int push_copy(adsb_decoder *d, const void *src, size_t n, hipMemcpyKind kind, bool async = false)
{
    const uint16_t *p = static_cast<const uint16_t *>(src);
    while (n) {
        const uint64_t room = d->stage_cap - kStageSlack - d->stage_fill;
        if (room == 0)
            return d->fail("staging buffer exhausted (stage_samples too small)");
        const size_t take = (size_t)std::min<uint64_t>(room, n);
        if (async) {
            d->piece++;
            const int cs = (int)(d->piece % adsb_decoder::kCopyStreams);
            hipStream_t cstream = d->dbg_async == 2 ? d->stream : d->copy_stream[cs];
            // Same rule as for the tail copy in process_stage: two writers that are not ordered never share a
            // cache line.  A piece that starts inside a 128-byte line (pushes of odd sizes) waits for the copy of
            // the piece before it, which ends in that line.
            if (d->piece > 1 && (d->stage_fill * sizeof(uint16_t)) % 128 != 0 && d->dbg_async != 2)
                HIP_TRY(d, hipStreamWaitEvent(cstream, d->ev_copy[cs ^ 1], 0));
            HIP_TRY(d, hipMemcpyAsync(d->stage[d->cur] + d->stage_fill, p, take * sizeof(uint16_t), kind, cstream));
            HIP_TRY(d, hipEventRecord(d->ev_copy[cs], cstream));
            if (d->dbg_async != 2)
                HIP_TRY(d, hipStreamWaitEvent(d->stream, d->ev_copy[cs], 0));
            if (d->dbg_async == 1)
                WAIT_STREAM(d, cstream, "a copy stream");
        } else {
            HIP_TRY(d, hipMemcpyAsync(d->stage[d->cur] + d->stage_fill, p, take * sizeof(uint16_t), kind, d->stream));
            d->copy_unconfirmed = true;
        }
        d->stage_fill += take;
        d->n_samples += take;
        p += take;
        n -= take;
        if (process_stage(d, false, async))
            return -1;
        if (async && d->piece > 1) {
            // adsb_push_async's contract: the buffer of the PREVIOUS piece is free when this call
            // returns.  Collecting that piece's scan implies it; a piece too small to launch a
            // scan leaves only its copy to wait for (already complete in every other case).
            WAIT_EVENT(d, d->ev_copy[(d->piece - 1) % adsb_decoder::kCopyStreams], "the host-to-device copy of the previous piece");
        }
    }
    return 0;
}

} // namespace

extern "C" {

adsb_decoder *adsb_create(const adsb_config *cfg_in)
{
    if ((g_cpu_refusal = adsb_host_cpu_refusal()) != nullptr)
        return nullptr;
    adsb_config cfg;
    adsb_debug_config dbg;
    if (const char *why = adsb::accept_config(cfg_in, cfg, dbg)) {
        g_create_error = why;
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device available: ") +
                         (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") +
                         " (libadsbdec_amd has no CPU fallback)";
        return nullptr;
    }
    int dev = cfg.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess)
        dev = 0;
    if (dev >= ndev) {
        g_create_error = "adsb_config.device is out of range";
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        g_create_error = "hipGetDeviceProperties failed";
        return nullptr;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName +
                         "; this library carries gfx950 (MI355X) code objects only";
        return nullptr;
    }
    adsb_decoder *d = new (std::nothrow) adsb_decoder();
    if (!d) {
        g_create_error = "out of memory";
        return nullptr;
    }
    d->cfg = cfg;
    d->dbg = dbg;
    d->device = dev;
    d->stage_cap = cfg.stage_samples ? round_down(cfg.stage_samples + 7, 8) : kDefaultStageSamples;
    if (d->stage_cap < (1u << 16))
        d->stage_cap = 1u << 16;

    std::thread warm, warm2; // cfg.warm_start: the process's first-use costs, paid beside the rest of this function
    struct JoinWarm {
        std::thread &a, &b;
        void join()
        {
            if (a.joinable())
                a.join();
            if (b.joinable())
                b.join();
        }
        ~JoinWarm() { join(); }
    } join_warm{warm, warm2}; // (every way out of this function waits for them)
    auto bail = [&](const char *what, hipError_t err) -> adsb_decoder * {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        join_warm.join();
        adsb_destroy(d);
        return nullptr;
    };
    if ((e = hipSetDevice(dev)) != hipSuccess)
        return bail("hipSetDevice", e);
    if (cfg.stream) {
        d->stream = static_cast<hipStream_t>(cfg.stream);
    } else {
        if ((e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", e);
        d->own_stream = true;
    }
    // The staging buffers first, so that a one-shot process (cfg.warm_start: the C host program) can pay the runtime's
    // first-use cost of a large page-locked host-to-device copy -- 7-9 ms inside the first such hipMemcpyAsync of a process,
    // profiles/r5_cli_timing.txt -- on a thread of its own while this one creates the other four streams (5-6 ms each).
    for (int i = 0; i < 2; i++)
        if ((e = hipMalloc(&d->stage[i], d->stage_cap * sizeof(uint16_t))) != hipSuccess)
            return bail("hipMalloc(stage)", e);
    if (cfg.warm_start) {
        const size_t bytes = std::min<size_t>(32u << 20, d->stage_cap * sizeof(uint16_t));
        try {
            warm = std::thread([d, bytes] {
                void *tmp = nullptr;
                if (hipSetDevice(d->device) != hipSuccess || hipHostMalloc(&tmp, bytes, hipHostMallocDefault) != hipSuccess)
                    return; // (best effort: the first push then pays what it always paid)
                std::memset(tmp, 0, 4096);
                // into BOTH staging buffers: the first copy into the second one -- at the stream's first compaction, seven
                // pushes into a 510 MiB file -- cost the C host program another 7 ms (profiles/r6_cli_timing.txt)
                for (int i = 0; i < 2; i++)
                    if (hipMemcpyAsync(d->stage[i], tmp, bytes, hipMemcpyHostToDevice, d->stream) != hipSuccess)
                        break;
                (void)hipStreamSynchronize(d->stream);
                (void)hipHostFree(tmp);
            });
        } catch (...) { // no thread to be had: nothing is warmed
        }
    }
    if (d->own_stream && !(tuning_env("ADSB_ALT_STREAMS") && atoi(tuning_env("ADSB_ALT_STREAMS")) == 0) &&
        (e = hipStreamCreateWithFlags(&d->stream2, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate(second scan stream)", e);
    for (int i = 0; i < adsb_decoder::kCopyStreams; i++)
        if ((e = hipStreamCreateWithFlags(&d->copy_stream[i], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&d->ev_copy[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipStreamCreate(copy)", e);
    if ((e = hipEventCreateWithFlags(&d->ev_tail, hipEventDisableTiming)) != hipSuccess)
        return bail("hipEventCreate(tail)", e);
    if (cfg.warm_start) {
        // ... and the first KERNEL on each copy stream: the staging buffer's first compaction puts the tail's copy kernel on
        // one of them, and the first dispatch on a stream that has only ever carried copies took 7 ms in the middle of the C
        // host program's pushes (profiles/r6_cli_timing.txt: push 7)
        try {
            warm2 = std::thread([d] {
                if (hipSetDevice(d->device) != hipSuccess)
                    return;
                for (hipStream_t cs : d->copy_stream)
                    (void)adsb::launch_copy_samples(d->stage[1] + 64, d->stage[1], 8, cs);
                // ... and the SECOND copy engine.  The runtime asks which engines are idle and takes another one when the
                // usual one is busy (hsa_amd_memory_copy_engine_status, hsa_amd_memory_async_copy_on_engine); an engine's
                // queue is created at its first use, 7.5 ms inside that call -- for the C host program in the copy behind
                // its first compaction, the first one issued while the previous piece's copy was still running
                // (profiles/r6_cli_trace.txt).  Two copies in flight at once, here, beside the rest of adsb_create.
                void *tmp = nullptr;
                const size_t bytes = std::min<size_t>(16u << 20, d->stage_cap * sizeof(uint16_t) / 4);
                if (hipHostMalloc(&tmp, bytes, hipHostMallocDefault) == hipSuccess) {
                    std::memset(tmp, 0, 4096);
                    for (int rep = 0; rep < 2; rep++)
                        for (int i = 0; i < adsb_decoder::kCopyStreams; i++)
                            (void)hipMemcpyAsync(reinterpret_cast<char *>(d->stage[1]) + (size_t)i * bytes, tmp, bytes, hipMemcpyHostToDevice,
                                                 d->copy_stream[i]);
                }
                for (hipStream_t cs : d->copy_stream)
                    (void)hipStreamSynchronize(cs);
                if (tmp)
                    (void)hipHostFree(tmp);
            });
        } catch (...) {
        }
    }
    for (ScanSlot &sl : d->slots) {
        if ((e = hipMalloc(&sl.d_counters, adsb::kDevCounterWords * sizeof(uint32_t))) != hipSuccess)
            return bail("hipMalloc(counters)", e);
        if ((e = hipMemset(sl.d_counters, 0, adsb::kDevCounterWords * sizeof(uint32_t))) != hipSuccess)
            return bail("hipMemset(counters)", e);
        if ((e = hipHostMalloc(&sl.h_counters, 2 * adsb::kCounterWords * sizeof(uint32_t), hipHostMallocCoherent)) != hipSuccess)
            return bail("hipHostMalloc(counters)", e);
        if ((e = hipEventCreate(&sl.ev_ready[0])) != hipSuccess || (e = hipEventCreate(&sl.ev_ready[1])) != hipSuccess)
            return bail("hipEventCreate", e);
    }
    {
        std::vector<uint32_t> synd(adsb::kSyndWords);
        adsb::make_syndrome_table(synd.data());
        if ((e = hipMalloc(&d->d_synd, synd.size() * sizeof(uint32_t))) != hipSuccess)
            return bail("hipMalloc(synd)", e);
        if ((e = hipMemcpy(d->d_synd, synd.data(), synd.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
            return bail("hipMemcpy(synd)", e);
    }
    if (cfg.fix_1bit) {
        std::vector<uint32_t> fix(adsb::kFixSlots);
        d->fix_mul = adsb::make_fix_table(fix.data());
        if ((e = hipMalloc(&d->d_fix, fix.size() * sizeof(uint32_t))) != hipSuccess)
            return bail("hipMalloc(fix)", e);
        if ((e = hipMemcpy(d->d_fix, fix.data(), fix.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
            return bail("hipMemcpy(fix)", e);
    }
    if (cfg.collect_stats) {
        for (int i = 0; i < 2; i++)
            if ((e = hipMalloc(&d->d_carry[i], (size_t)kCarryCap * sizeof(uint64_t))) != hipSuccess)
                return bail("hipMalloc(try carry)", e);
        // one allocation, so that adsb_reset clears both with one fill: 4 accumulators + 3 (4) carry counts
        if ((e = hipMalloc(&d->d_try_acc, kTryStateBytes)) != hipSuccess ||
            (e = hipMemset(d->d_try_acc, 0, kTryStateBytes)) != hipSuccess)
            return bail("hipMalloc(try counters)", e);
        d->d_carry_n = reinterpret_cast<uint32_t *>(d->d_try_acc + 4);
        d->frames_cap = 1u << 16; // accepted frames between two count passes (a 128 Mi-offset launch at 1 k frames/s: 13 k)
        if (dbg.frames_cap > 0) // tests: start small, so that the regrow path runs
            d->frames_cap = std::max<size_t>(8, (size_t)dbg.frames_cap);
        for (int i = 0; i < adsb_decoder::kFrameBufs; i++)
            if ((e = hipHostMalloc(&d->h_frames[i], d->frames_cap * sizeof(adsb::TryFrame), hipHostMallocDefault)) != hipSuccess ||
                (e = hipEventCreate(&d->ev_frames[i])) != hipSuccess)
                return bail("hipHostMalloc(accepted frames)", e);
        if ((e = hipMalloc(&d->d_frames, d->frames_cap * sizeof(adsb::TryFrame))) != hipSuccess)
            return bail("hipMalloc(accepted frames)", e);
        int prio_least = 0, prio_greatest = 0; // the count passes give way to the scans they run beside
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        if ((e = hipStreamCreateWithPriority(&d->count_stream, hipStreamNonBlocking, prio_least)) != hipSuccess)
            return bail("hipStreamCreate(count)", e);
        for (ScanSlot &sl : d->slots)
            if ((e = hipEventCreateWithFlags(&sl.ev_count, hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate(count)", e);
        d->res.log_accepted(true);
        d->res.log_into(reinterpret_cast<adsb::Resolver::LogEntry *>(d->h_frames[0] + 1), d->frames_cap - 1);
    }
    d->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        if (dbg.reader_min_tiles > 0)
            d->reader_min_tiles = (uint32_t)dbg.reader_min_tiles;
        if (d->cfg.host_threads >= 2)
            start_reader(d);
        if (d->cfg.host_threads >= 3)
            start_gang(d, std::min(d->cfg.host_threads - 2, 15));
    }
    d->no_streaming = dbg.no_streaming != 0;
    if (dbg.shard_head > 0)
        d->shard_head = (uint64_t)dbg.shard_head;
    d->dbg_async = tuning_env("ADSB_DEBUG_ASYNC") ? atoi(tuning_env("ADSB_DEBUG_ASYNC")) : 0;
    d->res.reset();
    join_warm.join();
    return d;
}

void adsb_destroy(adsb_decoder *d)
{
    if (!d)
        return;
    (void)hipSetDevice(d->device);
    if (d->reader) {
        d->reader->stop();
        delete d->reader;
    }
    if (d->gang) {
        d->res.set_gang(nullptr);
        d->gang->stop();
        delete d->gang;
    }
    for (hipStream_t cs : d->copy_stream)
        if (cs)
            (void)wait_stream(d, cs, "a copy stream (adsb_destroy)");
    if (d->stream)
        (void)wait_stream(d, d->stream, "the scan stream (adsb_destroy)");
    if (d->stream2) {
        (void)wait_stream(d, d->stream2, "the second scan stream (adsb_destroy)");
        (void)hipStreamDestroy(d->stream2);
    }
    if (d->count_stream) {
        (void)wait_stream(d, d->count_stream, "the try-count stream (adsb_destroy)");
        (void)hipStreamDestroy(d->count_stream);
    }
    for (int i = 0; i < adsb_decoder::kCopyStreams; i++) {
        if (d->ev_copy[i]) (void)hipEventDestroy(d->ev_copy[i]);
        if (d->copy_stream[i]) (void)hipStreamDestroy(d->copy_stream[i]);
    }
    if (d->ev_tail) (void)hipEventDestroy(d->ev_tail);
    if (d->ev_wait) (void)hipEventDestroy(d->ev_wait);
    for (int i = 0; i < 2; i++)
        if (d->stage[i])
            (void)hipFree(d->stage[i]);
    if (d->d_synd) (void)hipFree(d->d_synd);
    if (d->d_fix) (void)hipFree(d->d_fix);
    if (d->win_buf) (void)hipFree(d->win_buf);
    for (int i = 0; i < 2; i++)
        if (d->d_carry[i]) (void)hipFree(d->d_carry[i]);
    if (d->d_frames) (void)hipFree(d->d_frames);
    for (int b = 0; b < adsb_decoder::kFrameBufs; b++) {
        if (d->h_frames[b]) (void)hipHostFree(d->h_frames[b]);
        if (d->ev_frames[b]) (void)hipEventDestroy(d->ev_frames[b]);
    }
    if (d->d_try_acc) (void)hipFree(d->d_try_acc);
    for (ScanSlot &sl : d->slots) {
        if (sl.d_counters) (void)hipFree(sl.d_counters);
        if (sl.h_counters) (void)hipHostFree(sl.h_counters);
        if (sl.cands) (void)hipHostFree(sl.cands);
        if (sl.tries) (void)hipHostFree(sl.tries);
        if (sl.d_tries) (void)hipFree(sl.d_tries);
        if (sl.d_try_counts) (void)hipFree(sl.d_try_counts);
        if (sl.hand) (void)hipHostFree(sl.hand);
        for (hipEvent_t ev : sl.ev_ready)
            if (ev) (void)hipEventDestroy(ev);
        if (sl.ev_count) (void)hipEventDestroy(sl.ev_count);
    }
    if (d->own_stream && d->stream)
        (void)hipStreamDestroy(d->stream);
    delete d;
}

int adsb_reset(adsb_decoder *d)
{
    if (!d)
        return -1;
    bool busy = d->slot_count != 0;
    for (const ScanSlot &sl : d->slots)
        busy |= sl.busy;
    if (busy) {
        // launches still in flight (a push failed half-way, or adsb_push_async without adsb_sync):
        // let them end before their slots are reused -- their records are dropped with the stream
        HIP_TRY(d, hipSetDevice(d->device));
        for (hipStream_t cs : d->copy_stream)
            WAIT_STREAM(d, cs, "a copy stream");
        WAIT_STREAM(d, d->stream, "the scan stream");
        if (d->stream2)
            WAIT_STREAM(d, d->stream2, "the second scan stream");
        if (d->count_stream) {
            if (count_flush(d))
                return -1;
            WAIT_STREAM(d, d->count_stream, "the try-count stream");
        }
        for (ScanSlot &sl : d->slots) {
            // normally the report kernel behind each scan has left the counters zero; after a failed launch it may not have
            HIP_TRY(d, hipMemsetAsync(sl.d_counters, 0, adsb::kDevCounterWords * sizeof(uint32_t), d->stream));
            sl.launch_stream = nullptr; // every stream has been drained: nothing of the slot's past to order against ...
            sl.busy = false;
            sl.count_pending = false;
            sl.prof_pending[0] = sl.prof_pending[1] = false;
        }
        // ... except these fills: the slot's next launch may go to the second scan stream, which nothing orders behind
        // d->stream -- a late fill would zero the counters of a running scan
        WAIT_STREAM(d, d->stream, "the scan stream");
    }
    else if (wait_last_copy(d)) // a late asynchronous copy must not land in stage[0] beside the next stream's
        return -1;
    d->piece = 0;
    d->shard_on = false;
    d->final_follows = false;
    d->deferred_n = 0;
    d->deferred_slot = nullptr;
    d->deferred_base = 0;
    d->sink = ScanSink{};
    d->n_samples = 0;
    d->g_scanned = 0;
    d->finished = false;
    d->stage_first = 0;
    d->stage_fill = 0;
    d->cur = 0;
    d->res.reset();
    d->res.log_accepted(d->cfg.collect_stats != 0);
    if (d->acc_dirty) { // behind any count pass still queued -- or still to be enqueued (count_flush)
        if (d->pending.valid) {
            d->pending.clear_after = true;
        } else {
            HIP_TRY(d, hipSetDevice(d->device));
            HIP_TRY(d, hipMemsetAsync(d->d_try_acc, 0, kTryStateBytes, d->count_stream));
        }
    }
    d->acc_dirty = false;
    d->tries_unread = false;
    d->carry_maybe = false;
    d->have_prev_frame = false;
    // In a statistics run slot_head keeps turning: the next stream's first scan then does not have to wait for
    // the count pass that the last launch of this one left behind on the count stream.  Otherwise a stream of
    // one launch stays in slot 0 (the other slots' buffers are never allocated).
    if (!d->cfg.collect_stats)
        d->slot_head = 0;
    d->slot_count = 0;
    d->err.clear();
    return 0;
}

int adsb_push(adsb_decoder *d, const uint16_t *samples, size_t n)
{
    if (!d)
        return -1;
    if (d->finished)
        return d->fail("adsb_push after adsb_finish");
    if (stream_too_long(d, n))
        return -1;
    if (n == 0)
        return 0;
    if (!samples)
        return d->fail("adsb_push: NULL samples");
    HIP_TRY(d, hipSetDevice(d->device));
    if (d->cfg.push_overlap) {
        // The caller's ONE buffer (fileInput's iqbuff, air.c:230-239; the callback's transfer, air.c:173-177) is
        // only borrowed until its bytes are on the device: return when the COPY has completed and leave the scan
        // in flight -- the host is back in read() while the device scans, and this call has meanwhile collected
        // the frames of the previous one (frames arrive one call late, never reordered; adsb_finish / adsb_sync
        // deliver the rest).  That is adsb_push_async plus the wait for this piece's own copy.
        if (push_copy(d, samples, n, hipMemcpyHostToDevice, true))
            return -1;
        return wait_last_copy(d);
    }
    if (push_copy(d, samples, n, hipMemcpyHostToDevice))
        return -1;
    if (d->copy_unconfirmed) { // `samples` is only borrowed for the call: no scan behind the last copy has confirmed it
        WAIT_STREAM(d, d->stream, "the scan stream");
        d->copy_unconfirmed = false;
    }
    return 0;
}

int adsb_push_async(adsb_decoder *d, const uint16_t *samples, size_t n)
{
    if (!d)
        return -1;
    if (d->finished)
        return d->fail("adsb_push_async after adsb_finish");
    if (stream_too_long(d, n))
        return -1;
    if (n == 0)
        return 0;
    if (!samples)
        return d->fail("adsb_push_async: NULL samples");
    HIP_TRY(d, hipSetDevice(d->device));
    return push_copy(d, samples, n, hipMemcpyHostToDevice, true);
}

int adsb_sync(adsb_decoder *d)
{
    if (!d)
        return -1;
    HIP_TRY(d, hipSetDevice(d->device));
    if (scan_drain(d))
        return -1;
    if (!d->finished)
        d->res.advance(power_samples_produced(d->n_samples), d->g_scanned);
    for (hipStream_t cs : d->copy_stream)
        WAIT_STREAM(d, cs, "a copy stream");
    WAIT_STREAM(d, d->stream, "the scan stream"); // tail copies: every borrowed buffer is free
    if (d->stream2)
        WAIT_STREAM(d, d->stream2, "the second scan stream");
    return 0;
}

// "0000:c1:00.0" of HIP device `device`, as sysfs spells it (lower case); false: the runtime does not say
static bool device_bdf(int device, char (&bdf)[64])
{
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess)
        return false;
    for (char *c = bdf; *c; c++)
        if (*c >= 'A' && *c <= 'F')
            *c = (char)(*c - 'A' + 'a');
    return true;
}

int adsb_device_numa_node(int device)
{
    char bdf[64], path[160];
    if (!device_bdf(device, bdf))
        return -1;
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    int node = -1;
    if (f) {
        if (fscanf(f, "%d", &node) != 1)
            node = -1;
        fclose(f);
    }
    return node;
}

int adsb_device_cpulist(int device, char *out, size_t cap)
{
    if (!out || cap < 2)
        return -1;
    out[0] = 0;
    char bdf[64];
    if (!device_bdf(device, bdf))
        return -1;
    char path[160];
    if (adsb_device_numa_node(device) < 0)
        return 0;
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    FILE *f = fopen(path, "r");
    if (!f)
        return 0;
    const bool got = fgets(out, (int)cap, f) != nullptr;
    fclose(f);
    if (!got) {
        out[0] = 0;
        return 0;
    }
    size_t n = std::strlen(out);
    while (n && (out[n - 1] == '\n' || out[n - 1] == ' '))
        out[--n] = 0;
    return (int)n;
}

int adsb_host_register(void *p, size_t bytes)
{
    return (p && bytes && hipHostRegister(p, bytes, hipHostRegisterPortable) == hipSuccess) ? 0 : -1;
}

int adsb_host_unregister(void *p)
{
    return (p && hipHostUnregister(p) == hipSuccess) ? 0 : -1;
}

void *adsb_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) // (every device of the process may copy from it)
        return nullptr;
    return p;
}

void adsb_host_free(void *p)
{
    if (p && !adsb_host_release_mapped(p)) // (adsb_host_alloc_on / adsb_multi_host_alloc: a mapping of numa.cpp's)
        (void)hipHostFree(p);
}

} // extern "C"

namespace {

// adsb_push_device, optionally followed by adsb_finish in the same pass (`final`):
// the last in-place scan then runs to the exact end of the stream and no tail has to
// be staged.
int push_device_impl(adsb_decoder *d, const void *device_samples, size_t n, bool final)
{
    if (!d)
        return -1;
    if (d->finished)
        return d->fail("adsb_push_device after adsb_finish");
    if (final && d->shard_on)
        return d->fail("a shard stream ends with adsb_shard_end");
    if (stream_too_long(d, n))
        return -1;
    if (n && !device_samples)
        return d->fail("adsb_push_device: NULL samples");
    HIP_TRY(d, hipSetDevice(d->device));
    const uint16_t *p = static_cast<const uint16_t *>(device_samples);
    const bool aligned = (d->n_samples % 8 == 0) && ((uintptr_t)p % 16 == 0);
    if (!aligned || n < kInPlaceMinSamples) {
        if (n && push_copy(d, p, n, hipMemcpyDeviceToDevice))
            return -1;
        if (final)
            return adsb_finish(d);
        // the staging copy may still be queued when no scan was launched behind it (and
        // collected): the caller is free to reuse or free the buffer on return
        if (d->copy_unconfirmed) {
            WAIT_STREAM(d, d->stream, "the scan stream");
            d->copy_unconfirmed = false;
        }
        return 0;
    }

    // In-place scan.  First the seam: offsets whose window starts in earlier data.
    const uint64_t first = d->n_samples; // stream index of p[0]
    if (first != d->stage_first || d->stage_fill != 0) { // (not at the very start of a stream or of a shard: nothing earlier exists)
        if (push_copy(d, p, kSeamSamples, hipMemcpyDeviceToDevice))
            return -1;
    }
    // Bulk: every offset whose whole window lies inside this buffer.
    const uint64_t total = first + n;
    const uint64_t m_real = power_samples_produced(total);
    const uint64_t g_end = scannable_end(d, total, final);
    d->n_samples = total;
    if (final) {
        using clk = std::chrono::steady_clock;
        const bool dbg_on = tuning_env("ADSB_DEBUG_HOST") != nullptr;
        const auto t0 = clk::now();
        if (g_end > d->g_scanned) {
            d->alt_next = true; // in place: nothing on d->stream has to precede these launches
            const int rc_submit = scan_submit(d, p, first, n, d->g_scanned, g_end);
            d->alt_next = false;
            if (rc_submit)
                return -1;
            d->g_scanned = g_end;
        }
        const auto t1 = clk::now();
        d->final_follows = true;
        d->deferred_n = 0, d->deferred_slot = nullptr;
        const int rc = scan_drain(d);
        d->final_follows = false;
        if (rc)
            return -1;
        const auto t2 = clk::now();
        d->res.advance(2 * ((total + 3) / 4), d->g_scanned); // EOF rule: see process_stage()
        if (d->cfg.collect_stats && count_tries_pass(d, d->deferred_slot, d->deferred_n, d->deferred_base, true))
            return -1; // tries beyond the final position are never visited
        d->stage_fill = 0;
        d->finished = true;
        if (dbg_on) {
            auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            fprintf(stderr, "push_device_final: submit %.1f us, drain %.1f us, finish %.1f us\n", us(t0, t1), us(t1, t2),
                    us(t2, clk::now()));
        }
        return 0;
    }
    // Tail first: what the next push (or adsb_finish) still needs goes to the staging
    // buffer.  Issued ahead of the scans so that it is finished, in stream order, by
    // the time the last scan is collected (the caller may free the buffer on return).
    const uint64_t g_after = std::max(g_end, d->g_scanned);
    const uint64_t keep_first = line_aligned_keep(std::max<uint64_t>(g_after >= 8 ? 2 * (g_after - 8) : 0, first), total, first);
    const uint64_t left = total - keep_first;
    if (left > d->stage_cap - kStageSlack)
        return d->fail("in-place tail (%llu samples) exceeds the staging buffer", (unsigned long long)left);
    d->cur ^= 1; // the seam scan above may still be reading the other buffer
    HIP_TRY(d, adsb::launch_copy_samples(d->stage[d->cur], p + (keep_first - first), left, d->stream));
    d->stage_first = keep_first;
    d->stage_fill = left;
    if (g_end > d->g_scanned) {
        d->alt_next = true;
        const int rc_submit = scan_submit(d, p, first, n, d->g_scanned, g_end);
        d->alt_next = false;
        if (rc_submit)
            return -1;
        d->g_scanned = g_end;
    } else {
        WAIT_STREAM(d, d->stream, "the scan stream");
    }
    if (scan_drain(d))
        return -1;
    if (d->stream2) // the launches may all have gone to the second stream: the tail copy is not implied by their end
        WAIT_STREAM(d, d->stream, "the scan stream");
    d->res.advance(m_real, d->g_scanned);
    return 0;
}

} // namespace

extern "C" {

int adsb_push_device(adsb_decoder *d, const void *device_samples, size_t n)
{
    if (n == 0 && d && !d->finished)
        return 0;
    return push_device_impl(d, device_samples, n, false);
}

int adsb_push_device_final(adsb_decoder *d, const void *device_samples, size_t n)
{
    return push_device_impl(d, device_samples, n, true);
}

long adsb_decode_device(adsb_decoder *d, const void *device_samples, size_t n, const adsb_frame **frames)
{
    if (!d || !frames)
        return -1;
    if (adsb_reset(d) != 0 || push_device_impl(d, device_samples, n, true) != 0)
        return -1;
    return (long)d->res.take(frames);
}

int adsb_finish(adsb_decoder *d)
{
    if (!d)
        return -1;
    if (d->finished)
        return 0;
    if (d->shard_on)
        return d->fail("a shard stream ends with adsb_shard_end");
    HIP_TRY(d, hipSetDevice(d->device));
    if (process_stage(d, true))
        return -1;
    if (d->cfg.collect_stats && count_tries_pass(d, nullptr, 0, 0, true))
        return -1; // tries beyond the final position are never visited
    if (wait_last_copy(d)) // the contract of adsb_push_async: every borrowed buffer is free when adsb_finish returns,
        return -1;         // also when the last piece launched no scan that would have implied it
    d->finished = true;
    return 0;
}

long adsb_drain(adsb_decoder *d, adsb_frame *out, size_t cap)
{
    if (!d || (!out && cap))
        return -1;
    return (long)d->res.drain(out, cap);
}

long adsb_take(adsb_decoder *d, const adsb_frame **frames)
{
    if (!d || !frames)
        return -1;
    return (long)d->res.take(frames);
}

size_t adsb_pending(const adsb_decoder *d)
{
    return d ? d->res.pending() : 0;
}

int adsb_get_stats(const adsb_decoder *d, adsb_stats *out)
{
    if (!d || !out)
        return -1;
    adsb_decoder *m = const_cast<adsb_decoder *>(d); // the try counters live on the device until asked for
    if (m->cfg.collect_stats && (hipSetDevice(m->device) != hipSuccess || read_tries(m)))
        return -1;
    *out = m->res.stats();
    return 0;
}

int adsb_get_profile_sized(const adsb_decoder *d, adsb_profile *out, size_t size)
{
    if (!d || !out || size < offsetof(adsb_profile, offsets))
        return -1;
    adsb_decoder *m = const_cast<adsb_decoder *>(d); // kernel times are read from their events on demand
    for (auto &sl : m->slots)
        for (int pair = 0; pair < 2; pair++)
            if (slot_settle_profile(m, sl, pair))
                return -1;
    adsb_profile p = d->prof;
    // the threads of the handle's own that exist at this moment (they are started by the first launch behind a dense one,
    // or by adsb_create when cfg.host_threads asks for them, and live until adsb_destroy): none under ordinary traffic
    p.host_threads_running = (d->reader ? 1u : 0u) + (d->gang ? (uint32_t)d->gang->helpers() : 0u);
    std::memcpy(out, &p, std::min(size, sizeof p));
    return 0;
}

const char *adsb_last_error(const adsb_decoder *d)
{
    return d ? d->err.c_str() : g_cpu_refusal ? g_cpu_refusal : g_create_error.c_str();
}

// ---- stateless per-shard scan (multi-GPU path) -----------------
int adsb_scan_shard(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n,
                    uint64_t g_begin, uint64_t g_end, adsb_candidate *cands, size_t cand_cap,
                    size_t *n_cands, uint64_t *tries, size_t try_cap, size_t *n_tries)
{
    if (!d || !device_samples || !n_cands || !n_tries)
        return -1;
    if (first_sample % 8 || (uintptr_t)device_samples % 16)
        return d->fail("adsb_scan_shard: buffer must start at a multiple of 8 samples, 16-byte aligned");
    if (g_begin % 28)
        return d->fail("adsb_scan_shard: g_begin must be a multiple of 28");
    if (g_end > g_begin) {
        const uint64_t need_lo = g_begin >= 6 ? 2 * (g_begin - 6) : 0;
        const uint64_t need_hi = 2 * (g_end - 1 + ADSB_WINDOW);
        if (first_sample > need_lo || first_sample + n < need_hi)
            return d->fail("adsb_scan_shard: buffer does not cover the window of the owned offsets");
    }
    HIP_TRY(d, hipSetDevice(d->device));
    std::vector<adsb_candidate> cv;
    std::vector<uint64_t> tv;
    if (scan_drain(d))
        return -1;
    d->sink.cands = &cv;
    d->sink.tries = &tv;
    d->alt_next = true;
    int rc = scan_submit(d, static_cast<const uint16_t *>(device_samples), first_sample, n, g_begin, g_end);
    d->alt_next = false;
    if (rc == 0)
        rc = scan_drain(d);
    d->sink = ScanSink{};
    if (rc)
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

int adsb_scan_shard_host(adsb_decoder *d, const uint16_t *host_samples, uint64_t first_sample, size_t n, uint64_t g_begin,
                         uint64_t g_end, adsb_candidate *cands, size_t cand_cap, size_t *n_cands, uint64_t *tries, size_t try_cap,
                         size_t *n_tries)
{
    if (!d || !host_samples || !n_cands || !n_tries)
        return -1;
    HIP_TRY(d, hipSetDevice(d->device));
    if (n > d->win_cap) {
        if (d->win_buf)
            HIP_TRY(d, hipFree(d->win_buf));
        d->win_buf = nullptr;
        d->win_cap = 0;
        const size_t cap = (n + 65535) & ~(size_t)65535;
        HIP_TRY(d, hipMalloc(&d->win_buf, cap * sizeof(uint16_t)));
        d->win_cap = cap;
    }
    // (copied and waited for: the scan's launches may go to either compute stream, and a window is a few hundred KB.  On a
    // copy stream of the handle's: the synchronous hipMemcpy was seen to cost the process ~1 KB of host memory per call
    // that never came back -- tools/soak_probe.py)
    HIP_TRY(d, hipMemcpyAsync(d->win_buf, host_samples, n * sizeof(uint16_t), hipMemcpyHostToDevice, d->copy_stream[0]));
    WAIT_STREAM(d, d->copy_stream[0], "a copy stream");
    return adsb_scan_shard(d, d->win_buf, first_sample, n, g_begin, g_end, cands, cand_cap, n_cands, tries, try_cap, n_tries);
}

// The same scan, resolved on the fly by this handle's own resolver in chain mode (resolver.hpp): the streaming
// hand-off feeds it while the kernel runs, exactly like a stream's scan; the frames come out with shard-local ts.
int adsb_scan_shard_resolved(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n, uint64_t g_begin,
                             uint64_t g_end, adsb_shard_head *head, adsb_frame *frames, size_t frame_cap,
                             adsb_candidate *head_cands, size_t head_cap)
{
    return adsb_scan_shard_resolved_walk(d, device_samples, first_sample, n, g_begin, g_end, 0, head, frames, frame_cap, head_cands,
                                         head_cap, nullptr, 0);
}

} // extern "C"

namespace {

// what a shard's resolver knows when its chain has reached g_end (chain mode): into the head
void fill_shard_head(adsb_decoder *d, adsb_shard_head *head, uint64_t g_begin, uint64_t g_end, uint64_t head_end, size_t n_frames,
                     size_t bases_cap)
{
    head->g_begin = g_begin;
    head->g_end = g_end;
    head->n_frames = n_frames;
    head->n_head = d->shard_hv.size();
    head->head_end = head_end;
    head->skipped = d->res.skipped();
    if (bases_cap) { // (more bases than the caller's array holds: the stitcher must not use it)
        head->n_bases = d->res.walk_bases() <= bases_cap ? d->res.walk_bases() : 0;
        head->walk_final = d->res.walk_final() ? 1 : 0;
    }
    const adsb_stats &st = d->res.stats();
    for (int k = 0; k < 3; k++)
        head->ok[k] = st.ok[k];
    head->fixed = st.fixed;
}

// the shard's own Try count (collect_stats): every try of [g_begin, g_end) against the speculative frames, on the device
int shard_tries(adsb_decoder *d, adsb_shard_head *head)
{
    if (!d->cfg.collect_stats)
        return 0;
    if (count_tries_pass(d, nullptr, 0, 0, true) || read_tries(d))
        return -1;
    head->has_tries = 1;
    for (int k = 0; k < 3; k++)
        head->tries[k] = d->res.stats().try_[k];
    return 0;
}

// Scan + chain resolution of a shard that is resident in HBM; the results stay in the handle (resolver queue, shard_hv).
int scan_shard_resolved_core(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n, uint64_t g_begin,
                             uint64_t g_end, uint64_t total_samples, adsb_shard_head *head, const adsb_frame **fp, uint64_t *bases,
                             size_t bases_cap)
{
    std::memset(head, 0, sizeof *head);
    head->status = 1;
    *fp = nullptr;
    if (d->n_samples != 0 || d->res.pending() != 0) // (it runs this handle's own resolver: a stream in progress would be lost)
        return d->fail("adsb_scan_shard_resolved: the handle holds a stream (adsb_reset it, or use a handle of its own)");
    if (first_sample % 8 || (uintptr_t)device_samples % 16)
        return d->fail("adsb_scan_shard_resolved: buffer must start at a multiple of 8 samples, 16-byte aligned");
    if (g_begin % 28)
        return d->fail("adsb_scan_shard_resolved: g_begin must be a multiple of 28");
    if (g_end > g_begin) {
        const uint64_t need_lo = g_begin >= 6 ? 2 * (g_begin - 6) : 0;
        const uint64_t need_hi = 2 * (g_end - 1 + ADSB_WINDOW);
        if (first_sample > need_lo || first_sample + n < need_hi)
            return d->fail("adsb_scan_shard_resolved: buffer does not cover the window of the owned offsets");
    }
    HIP_TRY(d, hipSetDevice(d->device));
    if (scan_drain(d))
        return -1;
    const uint64_t head_end = std::min<uint64_t>(g_end, g_begin + d->shard_head); // (tests shrink the window to reach the stitcher's fallback)
    d->sink = ScanSink{};
    d->shard_hv.clear();
    d->res.start_chain(g_begin, head_end, &d->shard_hv);
    const size_t bcap = (bases && bases_cap) ? bases_cap : 0;
    if (bcap) // the shard's own walk of the deqframe calls, advanced beside the chain while the kernel runs
        d->res.start_walk(g_begin, g_end, total_samples, bases, bcap);
    d->alt_next = true;
    int rc = scan_submit(d, static_cast<const uint16_t *>(device_samples), first_sample, n, g_begin, g_end);
    d->alt_next = false;
    if (rc == 0)
        rc = scan_drain(d);
    if (rc)
        return -1;
    d->res.advance(0, g_end);
    if (shard_tries(d, head))
        return -1;
    const size_t nf = d->res.take(fp);
    fill_shard_head(d, head, g_begin, g_end, head_end, nf, bcap);
    head->status = 0;
    return 0;
}

} // namespace

extern "C" {

int adsb_scan_shard_resolved_walk(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n, uint64_t g_begin,
                                  uint64_t g_end, uint64_t total_samples, adsb_shard_head *head, adsb_frame *frames,
                                  size_t frame_cap, adsb_candidate *head_cands, size_t head_cap, uint64_t *bases,
                                  size_t bases_cap)
{
    if (!d || !device_samples || !head || (frame_cap && !frames) || (head_cap && !head_cands))
        return -1;
    const adsb_frame *fp = nullptr;
    const int rc = scan_shard_resolved_core(d, device_samples, first_sample, n, g_begin, g_end, total_samples, head, &fp, bases, bases_cap);
    bool fit = false;
    if (rc == 0) {
        fit = head->n_frames <= frame_cap && head->n_head <= head_cap;
        if (fit) {
            if (head->n_frames)
                std::memcpy(frames, fp, head->n_frames * sizeof(adsb_frame));
            if (head->n_head)
                std::memcpy(head_cands, d->shard_hv.data(), head->n_head * sizeof(adsb_candidate));
        } else {
            head->status = 1;
        }
    }
    const std::string why = d->err;
    if (adsb_reset(d) != 0) // (the device's Try accumulators start from zero again; the handle is an ordinary one again)
        return -1;
    if (rc) {
        d->err = why;
        return -1;
    }
    return fit ? 0 : -2;
}

int adsb_scan_shard_resolved_take(adsb_decoder *d, const void *device_samples, uint64_t first_sample, size_t n, uint64_t g_begin,
                                  uint64_t g_end, uint64_t total_samples, adsb_shard_head *head, const adsb_frame **frames,
                                  const adsb_candidate **head_cands, uint64_t *bases, size_t bases_cap)
{
    if (!d || !device_samples || !head || !frames || !head_cands)
        return -1;
    *head_cands = nullptr;
    if (adsb_reset(d) != 0) // what the previous call left in the handle (its frames, the Try accumulators) goes now
        return -1;
    if (scan_shard_resolved_core(d, device_samples, first_sample, n, g_begin, g_end, total_samples, head, frames, bases, bases_cap))
        return -1;
    *head_cands = d->shard_hv.empty() ? nullptr : d->shard_hv.data();
    return 0;
}

// ---- a shard fed piecewise: the same chain-mode resolution, driven by the handle's ordinary stream machinery --------------
int adsb_shard_begin(adsb_decoder *d, uint64_t first_sample, uint64_t g_begin, uint64_t g_end, uint64_t total_samples,
                     uint64_t *bases, size_t bases_cap)
{
    if (!d)
        return -1;
    if (first_sample % 8)
        return d->fail("adsb_shard_begin: first_sample must be a multiple of 8 samples");
    if (g_begin % 28 || g_end < g_begin)
        return d->fail("adsb_shard_begin: g_begin must be a multiple of 28 and g_end >= g_begin");
    if (first_sample > (g_begin >= 6 ? 2 * (g_begin - 6) : 0))
        return d->fail("adsb_shard_begin: the samples must start at least 6 pairs before the first owned offset");
    if (g_end > g_begin && 2 * (g_end - 1 + ADSB_WINDOW) > total_samples)
        return d->fail("adsb_shard_begin: the shard's last window lies beyond the stream");
    if (adsb_reset(d) != 0)
        return -1;
    d->shard_on = true;
    d->shard_g_begin = g_begin;
    d->shard_g_end = g_end;
    d->n_samples = first_sample;
    d->stage_first = first_sample;
    d->g_scanned = g_begin;
    d->shard_hv.clear();
    d->res.start_chain(g_begin, std::min<uint64_t>(g_end, g_begin + d->shard_head), &d->shard_hv);
    d->shard_bases_cap = (bases && bases_cap) ? bases_cap : 0;
    if (d->shard_bases_cap)
        d->res.start_walk(g_begin, g_end, total_samples, bases, bases_cap);
    return 0;
}

int adsb_shard_end(adsb_decoder *d, adsb_shard_head *head, const adsb_frame **frames, const adsb_candidate **head_cands)
{
    if (!d || !head || !frames || !head_cands)
        return -1;
    std::memset(head, 0, sizeof *head);
    head->status = 1;
    *frames = nullptr;
    *head_cands = nullptr;
    if (!d->shard_on)
        return d->fail("adsb_shard_end without adsb_shard_begin");
    const uint64_t g_begin = d->shard_g_begin, g_end = d->shard_g_end;
    if (g_end > g_begin && d->n_samples / 2 < g_end - 1 + ADSB_WINDOW)
        return d->fail("adsb_shard_end: %llu samples of the stream are in, the shard's last window ends at sample %llu",
                       (unsigned long long)d->n_samples, (unsigned long long)(2 * (g_end - 1 + ADSB_WINDOW)));
    HIP_TRY(d, hipSetDevice(d->device));
    if (process_stage(d, true)) // scans what is left, collects everything in flight, runs the chain to g_end
        return -1;
    if (d->g_scanned < g_end)
        return d->fail("internal: shard scanned to %llu of %llu", (unsigned long long)d->g_scanned, (unsigned long long)g_end);
    if (shard_tries(d, head))
        return -1;
    if (wait_last_copy(d)) // every borrowed buffer is free again
        return -1;
    d->finished = true; // (no further push: the next stream or shard starts with adsb_reset / adsb_shard_begin)
    const size_t nf = d->res.take(frames);
    *head_cands = d->shard_hv.empty() ? nullptr : d->shard_hv.data();
    fill_shard_head(d, head, g_begin, g_end, std::min<uint64_t>(g_end, g_begin + d->shard_head), nf, d->shard_bases_cap);
    head->status = 0;
    return 0;
}

} // extern "C"
