// handoff.hpp -- the HOST side of the device -> host hand-off stream (scan_kernel.h: marker + records per tile, written
// through by the kernel in tile COMPLETION order, validated by a checksum instead of a fence).
//
// What the reference has in this place is a mutex + condition-variable queue between the demodulator and the writer
// (output.c:159-202).  Here the producer is a GPU that cannot take a mutex, so the consumer polls memory the device
// writes and decides from the bytes alone when a tile is complete:
//   HandCursor    where the host stands in one launch's stream: marker check, tile ranges, frontier
//   StreamReader  cfg.host_threads = 2: a thread of the handle's own that runs the cursor and publishes the frontier
//   collect_alone / collect_behind_reader   the two consumer loops (the calling thread resolves through `flush`)
//
// Host-only code: no HIP in here.  The one question only the device runtime can answer -- "has the launch behind these
// bytes ended?" -- comes in as a callback, so that the whole of this file is built and run WITHOUT a GPU: the format is
// pinned by tests/test_handoff_cpu.py (adsb_handoff_walk), the threading by tests/cpp/handoff_tsan.cpp under
// ThreadSanitizer and AddressSanitizer with a thread that plays the device (random completion order, torn writes).
// x86-64 only (SSE2 loads, `pause`): the hosts MI355X ships in.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <thread>

#if !defined(__x86_64__)
#error "handoff.hpp reads the device's stores with SSE2 loads and paces its polls with `pause`: x86-64 hosts only (the hosts MI355X ships in); see DESIGN.md section 4"
#endif
#include <emmintrin.h>
#include <pthread.h>
#include <sched.h>

#include "scan_kernel_format.h"

namespace adsb {

// One launch's stream as the host sees it.
struct HandJob {
    const uint32_t *hand = nullptr; // the granules (pinned host memory the kernel writes through to)
    uint32_t ntiles = 0;            // tiles that write into it
    uint32_t gen = 0;               // the launch's tag (marker_check)
    uint32_t cap = 0;               // granules the stream can hold
    // Has the launch ended?  0: still running; 1: completed; -1: failed.  Null: nothing will ever complete these bytes
    // (adsb_handoff_walk over an image of a stream: one look).
    int (*done)(void *ctx) = nullptr;
    void *ctx = nullptr;
};

struct HandCursor {
    using clk = std::chrono::steady_clock;
    const HandJob s;
    uint32_t *t_start, *t_count; // per tile: granule index of its first record, and its record count (~0u: not in yet)
    const uint32_t gen, cap;
    uint32_t pos = 0;      // granules of the stream consumed
    uint32_t frontier = 0; // every tile below is in
    bool tries_listed = false; // some tile of a statistics run sent its tries through the launch-wide list (kMarkTries)
    uint32_t hold = ~0u;   // the lowest tile that says "some of my records are on the loose list": it, and every tile behind
                           // it, can only be handed on once the launch has ended -- but they are READ (and checked) meanwhile
    uint32_t tile = 0, nf = 0; // the marker tile_in() accepted last
    double wait_ms = 0;
    clk::time_point t_last_wait;

    HandCursor(const HandJob &job, uint32_t *ts, uint32_t *tc)
        : s(job), t_start(ts), t_count(tc), gen(job.gen), cap(job.cap), t_last_wait(clk::now())
    {
    }
    // marker {tile, n | flags, check}: valid once it and the XOR of the 2n granules behind it agree (16-byte loads; the
    // bytes are re-read on every poll).
    // The bytes are written by the device with no ordering whatsoever towards this thread -- that is the design: what is
    // consumed is decided by the 64-bit check, not by a happens-before edge (a range in which some granule has not landed
    // yet passes with probability 2^-64 per look) -- so the race detector is told to look away from
    // exactly these loads, and from nothing else in this file.
    __attribute__((no_sanitize("thread"))) bool tile_in()
    {
        const __m128i *gp = reinterpret_cast<const __m128i *>(s.hand) + pos;
        std::atomic_signal_fence(std::memory_order_seq_cst); // compiler: re-read the bytes on every poll
        const __m128i mk = _mm_load_si128(gp);
        alignas(16) uint32_t mw[4], a[4];
        _mm_store_si128(reinterpret_cast<__m128i *>(mw), mk);
        tile = mw[0], nf = mw[1];
        const uint32_t n = nf & 0xFFFFu;
        const bool fits = !(nf & kMarkNoFit);
        if (tile >= s.ntiles || (fits && (uint64_t)pos + 1 + 2ull * n > cap))
            return false; // not a marker of this launch (yet)
        // one pass over the records: the XOR of both granules, and the second summary -- rank-weighted, over {g_rel, pw}
        // of every record (scan_kernel_format.h) -- from the first granule's low words while it is in a register
        __m128i acc = _mm_setzero_si128();
        uint32_t sum = 0;
        if (fits)
            for (uint32_t r = 0; r < n; r++) {
                const __m128i g0 = _mm_load_si128(gp + 1 + 2 * r), g1 = _mm_load_si128(gp + 2 + 2 * r);
                acc = _mm_xor_si128(acc, _mm_xor_si128(g0, g1));
                const uint64_t w01 = (uint64_t)_mm_cvtsi128_si64(g0);
                sum += record_term(r, (uint32_t)w01, (uint32_t)(w01 >> 32));
            }
        _mm_store_si128(reinterpret_cast<__m128i *>(a), acc);
        uint32_t lo, hi;
        marker_check(tile, nf, gen, a[0], a[1], a[2], a[3], sum, lo, hi);
        return mw[2] == lo && mw[3] == hi;
    }
    // spin until tile_in(); gives up (false) once the kernel has long finished
    bool wait_tile()
    {
        constexpr int kPollPause = 4; // measured: 0..256 make no difference to the kernel or the step
        if (tile_in())
            return true;
        if (!s.done)
            return false;
        const auto t_w = clk::now();
        bool ok = false;
        uint64_t after_done = 0;
        for (uint64_t spins = 1;; spins++) {
            if (tile_in()) {
                ok = true;
                break;
            }
            // a few pauses between polls: the line being re-read has to be pulled out of this
            // core's cache by the very device write that is awaited
            for (int k = 0; k < kPollPause; k++)
                __builtin_ia32_pause();
            if ((spins & 0x3F) == 0) {
                const int q = s.done(s.ctx);
                if (q != 0 && (q < 0 || ++after_done > 2000))
                    break; // the launch failed, or it completed long ago: the bytes will not come
            }
        }
        t_last_wait = clk::now();
        wait_ms += std::chrono::duration<double, std::milli>(t_last_wait - t_w).count();
        return ok;
    }
    uint32_t deliverable() const { return std::min(frontier, hold); } // tiles below may go to the resolver now
    // Take the tile whose marker tile_in() just accepted.  0: taken (a tile that also has loose records lowers `hold`);
    // 1: its range ran past the array -- nothing of it, and no marker behind it, is in the stream: the stream ends here;
    // -1: the stream is corrupt.
    int take()
    {
        const uint32_t n = nf & 0xFFFFu;
        if (t_count[tile] != ~0u)
            return -1;
        if (nf & (kMarkOver | kMarkNoFit))
            hold = std::min(hold, tile);
        if (nf & kMarkTries)
            tries_listed = true;
        if (nf & kMarkNoFit)
            return 1;
        t_start[tile] = pos + 1;
        t_count[tile] = n;
        pos += std::max(marker_granules(nf), stream_granules(n)); // what the tile reserved (it may have kept fewer records than it reserved for)
        while (frontier < s.ntiles && t_count[frontier] != ~0u)
            frontier++;
        return 0;
    }
};

// How a collect ended: 0 every tile is in and has been handed on; 1 finish after completion (a tile has records on the loose
// list, or the stream is full: tiles from `delivered` on wait for the launch's end); -1 the stream is corrupt (a tile twice);
// -2 the launch ended and the bytes never came.  pos / tile: where the cursor stood; frontier: every tile below is in the
// stream and checked (t_start / t_count say where), whether or not it could be handed on yet.
struct CollectEnd {
    int status = 0;
    uint32_t pos = 0, tile = 0, frontier = 0;
    bool tries_listed = false; // kMarkTries seen: the launch-wide try list is in use, its length comes with the launch's counters
};

// A decoder's second host thread (cfg.host_threads = 2): it reads and checks the hand-off stream of the launch being
// collected and publishes how far the stream is complete, while the calling thread resolves behind it.  One thread
// doing both has ~125 us of work per 256 Mi-sample launch inside the ~105 us between the first tile's end and the
// last one's, and ends 15-20 us behind the kernel; split, neither side is the bottleneck.  The thread spins for a
// short while after a job (so that back-to-back launches find it awake), then sleeps.
//
// Synchronisation, in one place.  The job (job, t_start, t_count: plain members) is written by the caller BEFORE the
// seq_cst increment of job_seq and read by the reader AFTER its acquire load of job_seq.  t_start[] / t_count[] entries of
// tiles below `frontier` are written by the reader BEFORE the release store of frontier and read by the caller AFTER its
// acquire load.  end / wait_ms / busy_ms are written BEFORE the release store of done_seq and read AFTER the caller's
// acquire load of it.  `sleeping` + the mutex close the lost-wake-up window (seq_cst on both sides).
struct StreamReader {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<bool> sleeping{false}, quit{false};
    std::atomic<uint32_t> job_seq{0};
    void (*on_start)(void *) = nullptr; // runs first on the thread (the decoder selects its device there)
    void *on_start_ctx = nullptr;
    bool place = false; // keep the thread on the caller's L3 (place_reader_thread)
    int placed_l3 = -1;
    // the job (written by the caller before job_seq)
    HandJob job;
    uint32_t *t_start = nullptr, *t_count = nullptr;
    // progress and result (written by the reader)
    alignas(64) std::atomic<uint32_t> frontier{0};
    alignas(64) std::atomic<uint32_t> done_seq{0};
    CollectEnd end; // (status 1 / -1 / -2 as CollectEnd says)
    double wait_ms = 0, busy_ms = 0;

    static constexpr uint32_t kPublishEvery = 32; // tiles between two stores of `frontier` while the device is ahead
    static constexpr int kSpinUs = 400;

    void run_job()
    {
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        HandCursor cur(job, t_start, t_count);
        uint32_t published = 0;
        int status = 0;
        while (cur.frontier < job.ntiles) {
            if (cur.pos >= cur.cap) {
                status = 1;
                break;
            }
            if (!cur.tile_in()) {
                if (cur.deliverable() != published) // the device is behind: hand over what is in before waiting
                    frontier.store(published = cur.deliverable(), std::memory_order_release);
                if (!cur.wait_tile()) {
                    status = -2;
                    break;
                }
            }
            const int rc = cur.take();
            if (rc != 0) {
                status = rc;
                break;
            }
            if (cur.deliverable() - published >= kPublishEvery)
                frontier.store(published = cur.deliverable(), std::memory_order_release);
        }
        if (status == 0 && cur.hold != ~0u)
            status = 1;
        end.status = status;
        end.pos = cur.pos;
        end.tile = cur.tile;
        end.frontier = cur.frontier;
        end.tries_listed = cur.tries_listed;
        wait_ms = cur.wait_ms;
        busy_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count() - cur.wait_ms;
        frontier.store(cur.deliverable(), std::memory_order_release); // (what the caller may hand on: tiles below the first that holds)
    }
    void loop()
    {
        if (on_start)
            on_start(on_start_ctx);
        uint32_t seen = 0;
        for (;;) {
            // spin for a while, then sleep
            const auto t_idle = std::chrono::steady_clock::now();
            for (uint32_t spins = 1; job_seq.load(std::memory_order_acquire) == seen && !quit.load(std::memory_order_relaxed); spins++) {
                __builtin_ia32_pause();
                if ((spins & 0xFF) == 0 && std::chrono::steady_clock::now() - t_idle > std::chrono::microseconds(kSpinUs)) {
                    std::unique_lock<std::mutex> lk(mu);
                    sleeping.store(true);
                    cv.wait(lk, [&] { return quit.load() || job_seq.load() != seen; });
                    sleeping.store(false);
                }
            }
            if (quit.load())
                return;
            seen = job_seq.load(std::memory_order_acquire);
            run_job();
            done_seq.store(seen, std::memory_order_release);
        }
    }
    void start()
    {
        th = std::thread([this] { loop(); });
        pthread_setname_np(th.native_handle(), "adsb-reader"); // (what top -H and /proc/<pid>/task/*/comm show)
    }
    void post(const HandJob &j, uint32_t *ts, uint32_t *tc)
    {
        job = j, t_start = ts, t_count = tc;
        frontier.store(0, std::memory_order_relaxed);
        job_seq.fetch_add(1); // seq_cst, against `sleeping`
        if (sleeping.load()) {
            std::lock_guard<std::mutex> lk(mu);
            cv.notify_one();
        }
    }
    void stop()
    {
        if (!th.joinable())
            return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit.store(true);
            cv.notify_one();
        }
        th.join();
    }
};

// The reader thread is kept near the caller: on a core that shares the caller's L3 (the records it has checked are
// read again by the resolver), but neither the caller's own core nor its SMT sibling.  Checked again at every job
// (sched_getcpu is a vDSO call): a caller that has moved to another L3 takes the thread along -- left behind, on the
// other socket of a two-socket host, the pair is 2.5 x slower than one thread.  Best effort; silent on failure.
inline bool read_cpu_list(const char *fmt, int c, cpu_set_t *out)
{
    char path[160], buf[1024];
    snprintf(path, sizeof path, fmt, c);
    FILE *f = fopen(path, "r");
    if (!f)
        return false;
    const bool got = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    if (!got)
        return false;
    CPU_ZERO(out);
    for (char *p = buf; *p && *p != '\n';) { // "a-b,c,d-e"
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p)
            return false;
        long b = a;
        if (*e == '-')
            b = strtol(e + 1, &e, 10);
        for (long k = a; k <= b && k < CPU_SETSIZE; k++)
            CPU_SET((int)k, out);
        p = (*e == ',') ? e + 1 : e;
    }
    return true;
}

// the L3 a CPU belongs to, named by the lowest CPU that shares it (-1: unknown); sysfs is read once per CPU
inline int l3_of_cpu(int cpu)
{
    static std::atomic<int> cache[CPU_SETSIZE]; // 0: not looked up yet; else id + 2
    if (cpu < 0 || cpu >= CPU_SETSIZE)
        return -1;
    const int c = cache[cpu].load(std::memory_order_relaxed);
    if (c != 0)
        return c - 2;
    cpu_set_t l3;
    int id = -1;
    if (read_cpu_list("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu, &l3))
        for (int k = 0; k < CPU_SETSIZE; k++)
            if (CPU_ISSET(k, &l3)) {
                id = k;
                break;
            }
    cache[cpu].store(id + 2, std::memory_order_relaxed);
    return id;
}

// returns the L3 the thread was placed on (-1: not placed)
inline int place_reader_thread(std::thread &th, int cpu)
{
    if (cpu < 0)
        return -1;
    cpu_set_t l3, smt, allowed, want;
    if (!read_cpu_list("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu, &l3) ||
        !read_cpu_list("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu, &smt) ||
        sched_getaffinity(0, sizeof allowed, &allowed) != 0)
        return -1;
    CPU_ZERO(&want);
    int n = 0;
    for (int k = 0; k < CPU_SETSIZE; k++)
        if (CPU_ISSET(k, &l3) && CPU_ISSET(k, &allowed) && !CPU_ISSET(k, &smt))
            CPU_SET(k, &want), n++;
    if (n == 0 || pthread_setaffinity_np(th.native_handle(), sizeof want, &want) != 0)
        return -1;
    return l3_of_cpu(cpu);
}

// The tiles of the last resident round finish in a burst at the kernel's end: take them in small batches, so that little
// is left to do once the last one is in.
constexpr uint32_t kCollectGroup = 512, kCollectTailTiles = 768, kCollectTailGroup = 64;

// One thread: the calling thread reads the stream and resolves.  flush(upto) hands tiles [delivered, upto) on and must set
// delivered = upto; it is called whenever the device leaves the host nothing to read, or a group of tiles has accumulated.
template <class Flush>
CollectEnd collect_alone(const HandJob &job, uint32_t *t_start, uint32_t *t_count, uint32_t &delivered, Flush &&flush, double &wait_ms,
                         HandCursor::clk::time_point &t_last_wait)
{
    HandCursor cur(job, t_start, t_count);
    CollectEnd end;
    while (cur.frontier < job.ntiles) {
        if (cur.pos >= cur.cap) { // the stream is full: the rest of the launch is on the loose list
            end.status = 1;
            break;
        }
        if (!cur.tile_in()) {
            // the device is behind: use the time to resolve what is complete, then wait
            if (cur.deliverable() > delivered) {
                flush(cur.deliverable());
                continue;
            }
            if (!cur.wait_tile()) {
                end.status = -2;
                break;
            }
        }
        const int rc = cur.take();
        if (rc != 0) {
            end.status = rc;
            break;
        }
        if (cur.deliverable() > delivered &&
            cur.deliverable() - delivered >= (job.ntiles - delivered > kCollectTailTiles ? kCollectGroup : kCollectTailGroup))
            flush(cur.deliverable());
    }
    if (end.status >= 0 && cur.deliverable() > delivered)
        flush(cur.deliverable());
    if (end.status == 0 && cur.hold != ~0u)
        end.status = 1;
    end.pos = cur.pos;
    end.frontier = cur.frontier;
    end.tries_listed = cur.tries_listed;
    end.tile = end.status == -2 ? cur.frontier : cur.tile;
    wait_ms = cur.wait_ms;
    t_last_wait = cur.t_last_wait;
    return end;
}

// Two threads: the reader publishes its frontier, the calling thread resolves behind it.
// (idle(): called while there is nothing to hand on -- the caller may have work of its own waiting, e.g. batches that other
// threads have decided meanwhile; returns true if it did something)
struct NoIdle {
    bool operator()() const { return false; }
};
template <class Flush, class Idle = NoIdle>
CollectEnd collect_behind_reader(StreamReader &rd, const HandJob &job, uint32_t *t_start, uint32_t *t_count, uint32_t &delivered,
                                 Flush &&flush, double &wait_ms, HandCursor::clk::time_point &t_last_wait, Idle &&idle = Idle())
{
    using clk = HandCursor::clk;
    rd.post(job, t_start, t_count);
    const uint32_t seq = rd.job_seq.load(std::memory_order_relaxed);
    for (;;) {
        const bool fin = rd.done_seq.load(std::memory_order_acquire) == seq; // read BEFORE the frontier: a finished reader's is final
        const uint32_t f = rd.frontier.load(std::memory_order_acquire);
        if (f > delivered && (fin || f - delivered >= kCollectTailGroup)) {
            flush(f);
            continue;
        }
        if (fin)
            break;
        if (idle())
            continue;
        const auto t_w = clk::now();
        bool worked = false;
        while (rd.frontier.load(std::memory_order_relaxed) == f && rd.done_seq.load(std::memory_order_relaxed) != seq) {
            for (int k = 0; k < 32; k++) // poll gently: every look takes the line away from the thread that writes it
                __builtin_ia32_pause();
            // idle() that does something is resolve work of the caller's own, not a wait for the device: the wait interval
            // ends in front of it (round 5's advisor finding: it used to be booked as device wait, which understated
            // adsb_profile.host_ms on dense runs with a gang)
            const auto t_i = clk::now();
            if (idle()) {
                wait_ms += std::chrono::duration<double, std::milli>(t_i - t_w).count();
                worked = true;
                break;
            }
        }
        if (!worked) {
            t_last_wait = clk::now();
            wait_ms += std::chrono::duration<double, std::milli>(t_last_wait - t_w).count();
        }
    }
    CollectEnd end = rd.end; // (behind the acquire load of done_seq that ended the loop)
    if (end.status == -2)
        end.tile = delivered;
    return end;
}

} // namespace adsb
