// multi.cpp -- ONE process, several GPUs: the host of BASELINE configs[3] / configs[4] in C++ behind the C-ABI
// (adsb_multi_*, include/adsbdec_amd.h).  north_star: "the host stays in C ... sample buffers are chunked and sharded
// across the 8 GPUs of one node with a one-frame overlap so no RCCL collectives are needed (host gathers decoded frames)".
//
// What it stands behind in the reference: fileInput's loop (air.c:217-246) -- one producer, frames handed to netout in
// ascending order (output.c:159-182) -- for a capture that is cut into one contiguous shard per device.  The sequential
// rules the gather has to preserve are demod.c:86,99 (ts), demod.c:125-141 (greedy skip) and air.c:94-99 (the deqframe
// call pattern and its end-of-file horizon); stitch.hpp replays them over the shards' results.
//
// Structure: a worker thread per device, each with a decoder handle of its own (created by the worker: the GPU runtime's
// start is per device and slow, so the devices come up in parallel).  A decode posts one job to every worker:
//   shard   copy the shard's halo'd slice over the device's own link in pieces (adsb_shard_begin, adsb_push_async per
//           piece: the copy of piece k+1 overlaps the scan of piece k, the chain is resolved while the kernel runs),
//           adsb_shard_end; with statistics also the two small windows of tries the stitcher needs (adsb_scan_shard_host)
//   stream  an ordinary stream of its own (configs[3]: N captures on N devices)
// and the calling thread stitches (adsb_stitch_shards[_stats]) and gathers the frames.  A seam that cannot be decided from
// the head candidates (-3) sends the whole capture through ONE handle as an ordinary stream: slower, same bytes.
//
// Host-only code on purpose: nothing but the public C-ABI of the library is called from here (no HIP), so the threading
// can be built against a fake backend and run under ThreadSanitizer (tests/cpp/multi_tsan.cpp).
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/adsbdec_amd_diag.h"
#include "config_abi.hpp"

namespace {

using clk = std::chrono::steady_clock;
inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

thread_local std::string g_multi_create_error;

constexpr uint64_t kPieceSamples = 16ull << 20;   // 32 MiB per host-to-device piece
constexpr uint64_t kMinShardOffsets = 1ull << 17; // smaller shards are not worth a device (and statistics want the stream's
                                                  // last ADSB_TAIL_OFFSETS offsets and a head window inside ONE shard)
constexpr uint64_t kHeadTryReach = 1200;          // a frame that starts inside the head window ends at most this far behind it

enum class JobKind { None, Shard, Gather, Stream, Quit };

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
// A worker spins for a short while for its next job before it goes to sleep (a decode of slices that are resident in HBM takes
// a millisecond per call: a futex wake-up would be 5 % of it), in bursts of `pause` so that an SMT sibling keeps the core.
// The CALLER only ever spins for the short gather jobs: measured, a caller spinning through a shard job sat on the SMT
// sibling of the worker it had just woken (the scheduler's wake-affine choice) and made that worker 2.6 x slower.
constexpr double kWorkerSpinMs = 0.3, kGatherSpinMs = 0.3;
inline void relax_burst()
{
    for (int k = 0; k < 32; k++)
        cpu_relax();
}

struct Source { // where the samples of a job come from: exactly one of the three
    const uint16_t *mem = nullptr; // host memory, mem[0] = stream sample 0
    int fd = -1;                   // a file of uint16 samples (pread by the worker itself)
    const void *dev = nullptr;     // the worker's slice, resident in ITS device's HBM (dev[0] = stream sample `first`)
};

struct StreamResult {
    int rc = 0;
    std::string err;
    std::vector<adsb_frame> frames;
    adsb_stats stats{};
    double ms = 0;
};

struct Job {
    JobKind kind = JobKind::None;
    Source src;
    uint64_t total = 0;                       // samples of the whole stream
    uint64_t first = 0, n = 0, g_begin = 0, g_end = 0; // Shard: the plan's entry
    bool stats = false;
    uint64_t tail_first_offset = 0;           // Shard + stats: tries from here on are wanted as the tail window
    // Stream: streams [stream_lo, stream_lo + stream_step, ...) of the caller's arrays, results into `results`
    const Source *streams = nullptr;
    const uint64_t *stream_n = nullptr;
    int n_streams = 0, stream_lo = 0, stream_step = 1;
    StreamResult *results = nullptr;
    // Gather: this worker's final frames (`n_new` accepted by the seam repair, then `keep` speculative ones from `drop`
    // on, ts - ts_sub) into dst
    adsb_frame *dst = nullptr;
    const adsb_frame *new_src = nullptr;
    uint64_t n_new = 0, drop = 0, keep = 0;
    int64_t ts_sub = 0;
};

struct Worker {
    int index = 0, device = 0;
    adsb_config cfg{};
    adsb_debug_config dbg{}; // (the driver's copy of the caller's test knobs: cfg.debug points here while the handle is created)
    std::thread th;
    adsb_decoder *dec = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t posted = 0, done = 0; // jobs posted / finished (under mu)
    std::atomic<uint64_t> posted_a{0}, done_a{0}; // the same, for the spinning side of each hand-over
    std::atomic<uint64_t> beat{0}; // signs of life inside a job (one per piece pushed, window scanned, stream ended): what the
                                   // caller's deadline watches (wait_done)
    bool orphaned = false;         // (under mu) the driver gave this worker up: it cleans up after itself and ends
    // where the worker's slice of the last host-resident capture lives (adsb_multi_worker_placement); asked of the kernel
    // once per slice, not per decode
    int device_node = -1, slice_node = -1;
    double local_fraction = 0;
    const void *placed_ptr = nullptr;
    uint64_t placed_bytes = 0;
    Job job;
    // result of the last job
    int rc = 0;
    std::string err;
    double ms = 0, create_ms = 0;
    bool bound = false; // the thread runs on the CPUs of its device's NUMA node
    adsb_shard_head head{};
    const adsb_frame *frames = nullptr;
    const adsb_candidate *head_cands = nullptr;
    std::vector<uint64_t> bases, head_tries, tail_tries;
    uint64_t head_tries_end = 0, tail_from = ~0ull;
    std::vector<adsb_candidate> scratch_cands;
    static constexpr int kRing = 4;          // page-locked pieces of a file source: one borrowed by the copy in flight, one ready, two being read
    uint16_t *ring[kRing] = {};
    int ring_slots = 0;                      // of them allocated (a short slice needs fewer)
    uint64_t ring_samples = 0, piece = kPieceSamples;

    int fail(const char *fmt, ...)
    {
        char buf[640];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return rc = -1;
    }
    int fail_dec(const char *what)
    {
        const char *e = dec ? adsb_last_error(dec) : adsb_last_error(nullptr);
        return fail("device %d (worker %d): %s failed: %s", device, index, what, e ? e : "");
    }
};

// how long the driver waits for a worker that shows no sign of life (wait_done)
inline double worker_limit_s(const adsb_config &cfg)
{
    const double t = cfg.wait_timeout_s > 0 ? cfg.wait_timeout_s : 120;
    return t + std::max(5.0, t / 4);
}

} // namespace

struct adsb_multi {
    adsb_config cfg{};
    adsb_debug_config dbg{};
    std::vector<std::unique_ptr<Worker>> w;
    std::string err;
    std::vector<adsb_frame> out, new_frames;
    std::vector<StreamResult> streams;
    adsb_stats stats{};
    bool have_stats = false;
    adsb_multi_info info{};
    double create_ms = 0;
    uint64_t piece_samples = kPieceSamples;
    bool broken = false; // a worker stopped answering (gave_up): every later call fails at once

    long gave_up(const Worker &w)
    {
        broken = true;
        return fail("device %d (worker %d) has shown no sign of life for %.0f s (adsb_config.wait_timeout_s + margin): giving up; "
                    "this adsb_multi handle is unusable now -- destroy it, and keep the sample buffers / files of this call alive: the worker may still read them if it comes back",
                    w.device, w.index, worker_limit_s(cfg));
    }
    long fail(const char *fmt, ...)
    {
        char buf[768];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return -1;
    }
};

namespace {

// The next piece of a source: memory is pushed where it lies; a file is read into one of two page-locked buffers in turn
// (adsb_push_async borrows a buffer until the NEXT push returns, so two suffice).
// `slots` page-locked buffers of at least n samples each.  On the device's NUMA node and through adsb_host_alloc_on (huge
// pages registered with the runtime: 0.01 ms per MiB) where that works, else adsb_host_alloc (0.17 ms per MiB: for a
// one-shot host program the ring's allocation is part of the decode time).
bool ring_ready(Worker &w, uint64_t n, int slots = 2)
{
    slots = std::min(std::max(slots, 2), (int)Worker::kRing);
    if (w.ring_slots < slots || w.ring_samples < n) {
        const uint64_t cap = std::max(n, w.ring_samples);
        for (int i = 0; i < Worker::kRing; i++) {
            uint16_t *&b = w.ring[i];
            if (b && w.ring_samples >= cap)
                continue; // big enough already
            if (b)
                adsb_host_free(b);
            b = nullptr;
            if (i >= std::max(slots, w.ring_slots))
                continue;
            b = static_cast<uint16_t *>(adsb_host_alloc_on(2 * cap, w.device));
            if (!b)
                b = static_cast<uint16_t *>(adsb_host_alloc(2 * cap));
            if (!b) {
                w.ring_samples = 0;
                w.ring_slots = 0;
                w.fail("device %d (worker %d): cannot page-lock a %llu-byte read buffer", w.device, w.index, (unsigned long long)(2 * cap));
                return false;
            }
        }
        w.ring_samples = cap;
        w.ring_slots = std::max(slots, w.ring_slots);
    }
    return true;
}
bool read_samples(int fd, uint16_t *dst, uint64_t at, uint64_t n)
{
    uint64_t got = 0;
    while (got < 2 * n) {
        const ssize_t k = pread(fd, reinterpret_cast<char *>(dst) + got, 2 * n - got, (off_t)(2 * at + got));
        if (k <= 0)
            return false;
        got += (uint64_t)k;
    }
    return true;
}
const uint16_t *fetch(Worker &w, const Source &src, uint64_t at, uint64_t n, int turn)
{
    if (src.mem)
        return src.mem + at;
    if (!ring_ready(w, n))
        return nullptr;
    uint16_t *dst = w.ring[turn & 1];
    if (!read_samples(src.fd, dst, at, n)) {
        w.fail("device %d (worker %d): read of samples %llu.. failed or fell short", w.device, w.index, (unsigned long long)at);
        return nullptr;
    }
    return dst;
}

// A FILE source at the link's rate.  One thread's pread() from the page cache delivers 11.6 GB/s on the host measured
// (profiles/r5_ingest_probe.txt), a device's link takes 56: kFileReaders helper threads, started for the feed and gone
// with it, read AHEAD into the worker's ring of page-locked buffers.  The unit a thread claims is a 4 MiB block, not a
// piece: all of them work on the piece the worker needs next (the first piece of a feed is there after one block's time,
// not after one thread has read 32 MiB), a piece is pushed when its last block is in, and its buffer is free again once the
// push of the NEXT piece has returned (adsb_push_async's contract).  Pieces are the host source's 32 MiB: with 8 MiB
// pieces the same readers delivered 23.0 instead of 26.4 Gsamples/s, and twelve readers no more than eight -- the short
// pushes, not the reading, were the limit (profiles/r5_file_readers.txt).  (Zero copy -- windows of an mmap of the file
// registered with the runtime -- was measured too: the copies then run at the link's rate, but registering page-cache pages
// costs as much CPU per byte as copying them and does not scale with threads)
constexpr int kFileReaders = 8;
constexpr uint64_t kFileBlockSamples = 2ull << 20; // 4 MiB
int feed_file(Worker &w, const Source &src, uint64_t first, uint64_t n, uint64_t piece)
{
    const uint64_t npieces = (n + piece - 1) / piece;
    if (!ring_ready(w, piece, (int)std::min<uint64_t>(Worker::kRing, npieces + 1)))
        return -1;
    const uint64_t nring = (uint64_t)w.ring_slots;
    const uint64_t bpp = (piece + kFileBlockSamples - 1) / kFileBlockSamples; // blocks per piece
    const uint64_t nblocks = (npieces - 1) * bpp + ((n - (npieces - 1) * piece) + kFileBlockSamples - 1) / kFileBlockSamples;
    std::atomic<uint64_t> next{0}, freed{0}; // next block to claim; pieces [0, freed) have left their buffers
    std::atomic<bool> abort{false};
    std::unique_ptr<std::atomic<int>[]> left(new std::atomic<int>[npieces]); // blocks of the piece still to come; < 0: a read failed
    for (uint64_t k = 0; k < npieces; k++) {
        const uint64_t len = std::min(piece, n - k * piece);
        left[k].store((int)((len + kFileBlockSamples - 1) / kFileBlockSamples), std::memory_order_relaxed);
    }
    auto reader = [&] {
        for (;;) {
            const uint64_t b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= nblocks)
                return;
            const uint64_t k = b / bpp, off = (b % bpp) * kFileBlockSamples; // piece, and where in it
            for (unsigned spins = 0; k >= freed.load(std::memory_order_acquire) + nring; spins++) { // its slot is still borrowed
                if (abort.load(std::memory_order_relaxed))
                    return;
                if (spins < 64)
                    relax_burst();
                else
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            if (abort.load(std::memory_order_relaxed))
                return;
            const uint64_t plen = std::min(piece, n - k * piece), len = std::min(kFileBlockSamples, plen - off);
            if (read_samples(src.fd, w.ring[k % nring] + off, first + k * piece + off, len)) {
                left[k].fetch_sub(1, std::memory_order_acq_rel);
            } else {
                left[k].store(-1000000, std::memory_order_release);
                return;
            }
        }
    };
    std::vector<std::thread> readers;
    const int nthreads = (int)std::min<uint64_t>((uint64_t)kFileReaders, nblocks);
    int rc = 0;
    try {
        for (int t = 0; t < nthreads; t++)
            readers.emplace_back(reader);
    } catch (const std::exception &) { // no (more) threads to be had: with the ones that started; with none, the worker reads alone
    }
    if (readers.empty())
        return 1;
    for (uint64_t k = 0; k < npieces && rc == 0; k++) {
        int st;
        for (unsigned spins = 0; (st = left[k].load(std::memory_order_acquire)) > 0; spins++) {
            if (spins < 256)
                relax_burst();
            else
                std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        const uint64_t at = first + k * piece, len = std::min(piece, first + n - at);
        if (st < 0)
            rc = w.fail("device %d (worker %d): read of samples %llu.. failed or fell short", w.device, w.index, (unsigned long long)at);
        else if (adsb_push_async(w.dec, w.ring[k % nring], (size_t)len))
            rc = w.fail_dec("adsb_push_async");
        else
            freed.store(k, std::memory_order_release); // piece k - 1's buffer is free (adsb_push_async's contract); piece k's stays borrowed
        w.beat.fetch_add(1, std::memory_order_relaxed);
    }
    abort.store(true, std::memory_order_relaxed);
    for (auto &t : readers)
        t.join();
    return rc;
}

// DF-gate passes of the offsets [g_lo, g_hi) of the job's stream, through a stateless scan of just that window.
int window_tries(Worker &w, const Job &j, uint64_t g_lo, uint64_t g_hi, std::vector<uint64_t> &out)
{
    out.clear();
    if (g_hi <= g_lo)
        return 0;
    const uint64_t s0 = g_lo >= 8 ? 2 * (g_lo - 8) : 0;
    const uint64_t s1 = std::min<uint64_t>(j.total, 2 * (g_hi - 1 + ADSB_WINDOW));
    size_t nc = 0, nt = 0;
    size_t cap_c = std::max<size_t>(w.scratch_cands.size(), 4096), cap_t = std::max<size_t>(out.capacity(), 8192);
    for (int attempt = 0; attempt < 3; attempt++) {
        w.scratch_cands.resize(cap_c);
        out.resize(cap_t);
        int rc;
        if (j.src.dev) {
            rc = adsb_scan_shard(w.dec, static_cast<const uint16_t *>(j.src.dev) + (s0 - j.first), s0, (size_t)(s1 - s0), g_lo, g_hi,
                                 w.scratch_cands.data(), cap_c, &nc, out.data(), cap_t, &nt);
        } else {
            const uint16_t *p = fetch(w, j.src, s0, s1 - s0, 0);
            if (!p)
                return -1;
            rc = adsb_scan_shard_host(w.dec, p, s0, (size_t)(s1 - s0), g_lo, g_hi, w.scratch_cands.data(), cap_c, &nc, out.data(),
                                      cap_t, &nt);
        }
        w.beat.fetch_add(1, std::memory_order_relaxed);
        if (rc == 0) {
            out.resize(nt);
            return 0;
        }
        if (rc != -2)
            return w.fail_dec("adsb_scan_shard (tries of a window)");
        cap_c = std::max(cap_c, nc + 64);
        cap_t = std::max(cap_t, nt + 64);
    }
    return w.fail("device %d (worker %d): window of tries kept growing", w.device, w.index);
}

int feed(Worker &w, const Source &src, uint64_t first, uint64_t n, uint64_t piece)
{
    if (!src.mem && src.fd >= 0) {
        const int rc = feed_file(w, src, first, n, piece);
        if (rc <= 0)
            return rc;
        // (1: no helper thread could be started: piece by piece on this thread, a read and a copy in turn)
    }
    int turn = 0;
    for (uint64_t at = first; at < first + n; at += piece, turn++) {
        const uint64_t len = std::min(piece, first + n - at);
        const uint16_t *p = fetch(w, src, at, len, turn);
        if (!p)
            return -1;
        if (adsb_push_async(w.dec, p, (size_t)len))
            return w.fail_dec("adsb_push_async");
        w.beat.fetch_add(1, std::memory_order_relaxed);
    }
    return 0;
}

void run_shard(Worker &w, const Job &j, uint64_t piece)
{
    w.head = adsb_shard_head{};
    w.head.status = 1;
    w.frames = nullptr;
    w.head_cands = nullptr;
    w.head_tries.clear();
    w.tail_tries.clear();
    w.head_tries_end = 0;
    w.tail_from = ~0ull;
    if (j.g_end <= j.g_begin) { // nothing to own (a stream shorter than one window)
        w.head.g_begin = j.g_begin;
        w.head.g_end = j.g_end;
        w.head.status = 0;
        w.head.has_tries = j.stats ? 1 : 0;
        return;
    }
    // (the two windows are scanned outside the shard's stream: before it starts, and after it has ended -- a stateless
    // scan does not touch the resolver, so the frames adsb_shard_end handed out stay where they are)
    if (j.stats) {
        const uint64_t head_span = w.dbg.shard_head > 0 ? (uint64_t)w.dbg.shard_head : ADSB_SHARD_HEAD;
        w.head_tries_end = std::min(j.g_end, j.g_begin + head_span + kHeadTryReach);
        if (window_tries(w, j, j.g_begin, w.head_tries_end, w.head_tries))
            return;
    }
    const uint64_t calls = (j.g_end - j.g_begin) / 39780 + 8; // one deqframe call per 39 780 offsets, and a few
    if (w.bases.size() < calls)
        w.bases.resize(calls);
    if (j.src.dev) {
        // resident in this device's HBM: one call, one launch per 128 Mi offsets, resolved while the kernels run; the frames
        // stay in the handle's queue until this worker's next job
        if (adsb_scan_shard_resolved_take(w.dec, j.src.dev, j.first, (size_t)j.n, j.g_begin, j.g_end, j.total, &w.head, &w.frames,
                                          &w.head_cands, w.bases.data(), w.bases.size())) {
            w.head.status = 1;
            w.fail_dec("adsb_scan_shard_resolved_take");
            return;
        }
    } else {
    if (j.src.mem) { // which socket does this link pull its slice from?  (a move_pages query of 256 pages, once per slice)
        const void *sp = j.src.mem + j.first;
        if (sp != w.placed_ptr || 2 * j.n != w.placed_bytes) {
            w.placed_ptr = sp;
            w.placed_bytes = 2 * j.n;
            w.slice_node = -1;
            w.local_fraction = 0;
            (void)adsb_host_placement(sp, (size_t)(2 * j.n), w.device_node, &w.slice_node, &w.local_fraction);
        }
    } else {
        w.placed_ptr = nullptr;
        w.slice_node = -1;
        w.local_fraction = 0;
    }
    if (adsb_shard_begin(w.dec, j.first, j.g_begin, j.g_end, j.total, w.bases.data(), w.bases.size())) {
        w.fail_dec("adsb_shard_begin");
        return;
    }
    if (feed(w, j.src, j.first, j.n, piece))
        return;
    if (adsb_shard_end(w.dec, &w.head, &w.frames, &w.head_cands)) {
        w.head.status = 1;
        w.fail_dec("adsb_shard_end");
        return;
    }
    }
    if (j.stats && j.g_end > j.tail_first_offset) {
        w.tail_from = std::max(j.g_begin, j.tail_first_offset - j.tail_first_offset % 28);
        if (window_tries(w, j, w.tail_from, j.g_end, w.tail_tries))
            w.head.status = 1;
    }
}

// demod.c:86,99: ts counts loop passes from the stream's start; a shard's frames carry the count from the shard's start
inline void copy_frames_fix_ts(adsb_frame *dst, const adsb_frame *src, uint64_t n, int64_t ts_sub)
{
    for (uint64_t k = 0; k < n; k++) {
        dst[k] = src[k];
        dst[k].ts = (uint64_t)((int64_t)src[k].ts - ts_sub);
    }
}

void run_gather(Worker &w, const Job &j)
{
    if (j.n_new)
        std::memcpy(j.dst, j.new_src, j.n_new * sizeof(adsb_frame));
    if (j.keep)
        copy_frames_fix_ts(j.dst + j.n_new, w.frames + j.drop, j.keep, j.ts_sub);
}

void run_streams(Worker &w, const Job &j, uint64_t piece)
{
    for (int s = j.stream_lo; s < j.n_streams; s += j.stream_step) {
        StreamResult &r = j.results[s];
        const auto t0 = clk::now();
        r = StreamResult{};
        w.err.clear();
        w.rc = 0;
        const adsb_frame *fp = nullptr;
        long nf = -1;
        if (adsb_reset(w.dec)) {
            w.fail_dec("adsb_reset");
        } else if (feed(w, j.streams[s], 0, j.stream_n[s], piece) == 0) {
            if (adsb_finish(w.dec))
                w.fail_dec("adsb_finish");
            else if ((nf = adsb_take(w.dec, &fp)) < 0)
                w.fail_dec("adsb_take");
        }
        if (w.rc == 0 && w.cfg.collect_stats && adsb_get_stats(w.dec, &r.stats))
            w.fail_dec("adsb_get_stats");
        if (w.rc == 0 && nf > 0)
            r.frames.assign(fp, fp + nf);
        r.rc = w.rc;
        r.err = w.err;
        r.ms = ms_since(t0);
        w.beat.fetch_add(1, std::memory_order_relaxed);
    }
}

// The worker consumes its handle's hand-off stream: memory the device writes, polled and resolved while the kernel runs
// (DESIGN.md section 4).  From a core on the other socket that work was measured 3 x slower (1.90 against 0.61 ms of host
// time per 1 Gi samples: profiles/r4_ab_runs.txt) and the worker, not the kernel, bounded the step.  So every worker
// moves to the CPUs of its device's NUMA node -- those of them the process is allowed to use.  Best effort.
bool bind_near_device(int device)
{
    char list[1024];
    if (adsb_device_cpulist(device, list, sizeof list) <= 0)
        return false;
    cpu_set_t allowed, want;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0)
        return false;
    CPU_ZERO(&want);
    int n = 0;
    for (const char *p = list; *p;) { // "a-b,c,d-e"
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p)
            return false;
        long b = a;
        if (*e == '-')
            b = strtol(e + 1, &e, 10);
        for (long k = a; k <= b && k < CPU_SETSIZE; k++)
            if (CPU_ISSET((int)k, &allowed)) {
                CPU_SET((int)k, &want);
                n++;
            }
        p = (*e == ',') ? e + 1 : e;
        if (*e && *e != ',')
            break;
    }
    return n > 0 && pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
}

void worker_main(Worker *w, uint64_t piece)
{
    {   // the handle is created here, on the worker's own thread: the devices' runtimes come up side by side
        const auto t0 = clk::now();
        adsb_config cfg = w->cfg;
        cfg.device = w->device;
        cfg.stream = nullptr;
        cfg.debug = &w->dbg;
        w->bound = bind_near_device(w->device); // first: the handle's page-locked buffers are then allocated from this thread's node
        w->device_node = adsb_device_numa_node(w->device);
        w->dec = adsb_create(&cfg);
        if (!w->dec)
            w->fail_dec("adsb_create");
        w->create_ms = ms_since(t0);
        std::lock_guard<std::mutex> lk(w->mu);
        w->done = 1; // "job" 1 is the start-up
        w->done_a.store(1, std::memory_order_release);
        w->cv.notify_all();
    }
    uint64_t seen = 1;
    for (;;) {
        Job j;
        for (const auto t0 = clk::now(); w->posted_a.load(std::memory_order_acquire) <= seen && ms_since(t0) < kWorkerSpinMs;)
            relax_burst();
        bool orphan = false;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->posted > seen || w->orphaned; });
            orphan = w->orphaned;
            seen = w->posted;
            j = w->job;
        }
        if (orphan) { // the driver stopped waiting for this worker (wait_done's deadline) and is gone: nobody joins, nobody frees
            if (w->dec)
                adsb_destroy(w->dec);
            for (uint16_t *b : w->ring)
                if (b)
                    adsb_host_free(b);
            delete w;
            return;
        }
        if (j.kind == JobKind::Quit)
            break;
        const auto t0 = clk::now();
        w->rc = 0;
        w->err.clear();
        try {
            if (!w->dec)
                w->fail("device %d (worker %d): no decoder handle (adsb_create failed at start-up)", w->device, w->index);
            else if (j.kind == JobKind::Shard)
                run_shard(*w, j, piece);
            else if (j.kind == JobKind::Gather)
                run_gather(*w, j);
            else if (j.kind == JobKind::Stream)
                run_streams(*w, j, piece);
        } catch (const std::exception &e) { // (bad_alloc of a result vector: the job fails, the thread lives)
            w->fail("device %d (worker %d): %s", w->device, w->index, e.what());
        }
        if (j.kind != JobKind::Gather)
            w->ms = ms_since(t0);
        std::lock_guard<std::mutex> lk(w->mu);
        w->done = seen;
        w->done_a.store(seen, std::memory_order_release);
        w->cv.notify_all();
    }
    if (w->dec)
        adsb_destroy(w->dec);
    for (uint16_t *b : w->ring)
        if (b)
            adsb_host_free(b);
}

void post(Worker &w, const Job &j)
{
    std::lock_guard<std::mutex> lk(w.mu);
    w.job = j;
    w.posted++;
    w.posted_a.store(w.posted, std::memory_order_release);
    w.cv.notify_all();
}

// Wait for the job just posted to worker w.  A worker's own waits for its device are bounded (adsb_config.wait_timeout_s); the
// caller's wait for the worker is bounded too: if the worker shows no sign of life (Worker::beat) for that limit plus a
// margin, the wait ends with false and the driver marks itself broken -- it answers -1 from then on, naming the worker,
// instead of sleeping on a condition variable for ever.
bool wait_done(Worker &w, double spin_ms = 0)
{
    const uint64_t want = w.posted_a.load(std::memory_order_relaxed); // (this thread posted it)
    for (const auto t0 = clk::now(); spin_ms > 0 && ms_since(t0) < spin_ms;) {
        if (w.done_a.load(std::memory_order_acquire) == want)
            return true;
        relax_burst();
    }
    const double limit_ms = 1e3 * worker_limit_s(w.cfg);
    uint64_t beat = w.beat.load(std::memory_order_relaxed);
    auto t_beat = clk::now();
    std::unique_lock<std::mutex> lk(w.mu);
    // (wait_until on the system clock = pthread_cond_timedwait; wait_for would be pthread_cond_clockwait, which gcc 11's
    // ThreadSanitizer does not intercept and then reports as a double lock.  A clock step only moves one 250 ms tick.)
    while (!w.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(250), [&] { return w.done == w.posted; })) {
        const uint64_t b = w.beat.load(std::memory_order_relaxed);
        if (b != beat) {
            beat = b;
            t_beat = clk::now();
        } else if (ms_since(t_beat) > limit_ms) {
            return false;
        }
    }
    return true;
}

// How many shards a stream of `total` samples is cut into on n workers, and the plan.
int plan(int n_workers, uint64_t total, std::vector<uint64_t> (&p)[4])
{
    const uint64_t m = 2 * (total / 4);
    const uint64_t n_off = m >= ADSB_WINDOW ? m - ADSB_WINDOW + 1 : 0;
    int n = (int)std::min<uint64_t>((uint64_t)n_workers, std::max<uint64_t>(1, n_off / kMinShardOffsets));
    for (auto &v : p)
        v.assign((size_t)n, 0);
    adsb_plan_shards(total, n, p[0].data(), p[1].data(), p[2].data(), p[3].data());
    return n;
}

// One ordinary stream through worker 0: what an undecidable seam falls back to (same bytes as the sharded decode).
long whole_stream(adsb_multi *m, const Source &src, uint64_t total)
{
    m->streams.assign(1, StreamResult{});
    Job j;
    j.kind = JobKind::Stream;
    j.streams = &src;
    j.stream_n = &total;
    j.n_streams = 1;
    j.results = m->streams.data();
    post(*m->w[0], j);
    if (!wait_done(*m->w[0]))
        return m->gave_up(*m->w[0]);
    StreamResult &r = m->streams[0];
    if (r.rc)
        return m->fail("fallback decode on one device failed: %s", r.err.c_str());
    m->out.swap(r.frames);
    m->stats = r.stats;
    m->have_stats = m->cfg.collect_stats != 0;
    return (long)m->out.size();
}

long decode_sharded_impl(adsb_multi *m, const Source &src, const void *const *slices, int n_slices, uint64_t total,
                         const adsb_frame **frames)
{
    if (m->broken)
        return m->fail("this adsb_multi handle is unusable: a worker stopped answering earlier (destroy it)");
    if (!m || !frames)
        return -1;
    *frames = nullptr;
    m->err.clear();
    m->have_stats = false;
    m->info = adsb_multi_info{};
    if (total >= (1ull << 32))
        return m->fail("stream of 2^32 samples or more: the reference's sample counter wraps there (air.c:34)");
    const auto t_begin = clk::now();
    std::vector<uint64_t> p[4];
    const int n = plan((int)m->w.size(), total, p);
    if (slices && n_slices != n)
        return m->fail("adsb_multi_decode_device: %d slices given, the plan for %llu samples has %d shards (adsb_multi_plan)", n_slices,
                       (unsigned long long)total, n);
    const bool stats = m->cfg.collect_stats != 0;
    const uint64_t m_ref = 2 * ((total + 3) / 4);
    for (int i = 0; i < n; i++) {
        Job j;
        j.kind = JobKind::Shard;
        j.src = src;
        if (slices)
            j.src.dev = slices[i];
        j.total = total;
        j.g_begin = p[0][i];
        j.g_end = p[1][i];
        j.first = p[2][i];
        j.n = p[3][i];
        j.stats = stats;
        j.tail_first_offset = m_ref > ADSB_TAIL_OFFSETS ? m_ref - ADSB_TAIL_OFFSETS : 0;
        post(*m->w[i], j);
    }
    double worker_ms = 0;
    for (int i = 0; i < n; i++) {
        if (!wait_done(*m->w[i]))
            return m->gave_up(*m->w[i]);
        worker_ms = std::max(worker_ms, m->w[i]->ms);
    }
    for (int i = 0; i < n; i++)
        if (m->w[i]->rc || m->w[i]->head.status)
            return m->fail("shard %d of %d: %s", i, n, m->w[i]->err.empty() ? "failed" : m->w[i]->err.c_str());
    // ---- the serial part, on the calling thread: seams, ts offsets, horizon (and statistics), then the gather
    const auto t_serial = clk::now();
    std::vector<adsb_shard_part> parts((size_t)n);
    std::vector<adsb_shard_fix> fix((size_t)n);
    for (int i = 0; i < n; i++) {
        Worker &w = *m->w[i];
        adsb_shard_part &q = parts[i];
        std::memset(&q, 0, sizeof q);
        q.head = &w.head;
        q.frames = w.frames;
        q.head_cands = w.head_cands;
        q.bases = w.head.n_bases ? w.bases.data() : nullptr;
        q.tail_from = ~0ull;
        if (stats) {
            q.head_tries = w.head_tries.data();
            q.n_head_tries = w.head_tries.size();
            q.head_tries_end = w.head_tries_end;
            q.tail_tries = w.tail_tries.data();
            q.n_tail_tries = w.tail_tries.size();
            q.tail_from = w.tail_from;
        }
    }
    if (m->new_frames.size() < 4096)
        m->new_frames.resize(4096);
    size_t n_new = 0;
    uint64_t ws[2] = {0, 0};
    int rc;
    for (;;) {
        rc = stats ? adsb_stitch_shards_stats(parts.data(), n, total, fix.data(), m->new_frames.data(), m->new_frames.size(), &n_new,
                                              ws, &m->stats)
                   : adsb_stitch_shards_ex(parts.data(), n, total, fix.data(), m->new_frames.data(), m->new_frames.size(), &n_new, ws);
        if (rc != -2 || m->new_frames.size() >= (1u << 22))
            break;
        m->new_frames.resize(std::max(m->new_frames.size() * 4, n_new + 4096)); // new_cap too small: seams that accept many frames
    }
    if (rc != 0 && m->new_frames.size() > (1u << 16)) { // (a failure does not leave a grown buffer behind in the handle)
        m->new_frames.resize(4096);
        m->new_frames.shrink_to_fit();
    }
    m->info.shards = n;
    for (int i = 0; i < n; i++)
        m->info.workers_bound += m->w[i]->bound ? 1 : 0;
    m->info.calls_walked = ws[0];
    m->info.calls_jumped = ws[1];
    m->info.workers_ms = worker_ms;
    if (rc == -3) {
        if (slices)
            return m->fail("a seam cannot be decided from the head candidates and the capture is not in host memory: decode it as "
                           "one stream (adsb_decode_device) on one device");
        m->info.fallback = 1;
        const long k = whole_stream(m, src, total);
        if (k >= 0)
            *frames = m->out.data();
        m->info.total_ms = ms_since(t_begin);
        return k;
    }
    if (rc != 0)
        return m->fail("adsb_stitch_shards failed (%d)", rc);
    const double stitch_us = 1e3 * ms_since(t_serial);
    size_t count = 0;
    for (int i = 0; i < n; i++)
        count += (size_t)(fix[i].n_new + fix[i].keep);
    m->out.resize(count);
    adsb_frame *o = m->out.data();
    // The gather: shard i's final frames are new_frames[new_first ..] and its speculative frames from drop_front on, with
    // the shard's ts offset taken off.  Every worker copies its own share (it is awake: it has just finished), also when
    // there is only one: the frames lie in the worker's handle, whose resolver rewrites that memory during the next call --
    // read from the calling thread, on another core or socket, every line of it had to be fetched back first (measured: the
    // worker's host time per 1 Gi samples went from 0.55 to 1.9 ms, and it, not the kernel, bounded the step).  Only a
    // handful of frames is not worth a hand-over.
    const bool spread = count >= 2048;
    for (int i = 0; i < n; i++) {
        Job g;
        g.kind = JobKind::Gather;
        g.dst = o;
        g.new_src = m->new_frames.data() + fix[i].new_first;
        g.n_new = fix[i].n_new;
        g.drop = fix[i].drop_front;
        g.keep = fix[i].keep;
        g.ts_sub = fix[i].ts_sub;
        if (spread)
            post(*m->w[i], g);
        else
            run_gather(*m->w[i], g);
        o += fix[i].n_new + fix[i].keep;
    }
    if (spread)
        for (int i = 0; i < n; i++)
            if (!wait_done(*m->w[i], kGatherSpinMs))
                return m->gave_up(*m->w[i]);
    m->have_stats = stats;
    m->info.stitch_us = stitch_us;
    m->info.serial_us = 1e3 * ms_since(t_serial);
    m->info.total_ms = ms_since(t_begin);
    *frames = count ? m->out.data() : nullptr;
    return (long)count;
}

int decode_streams_impl(adsb_multi *m, const std::vector<Source> &src, const std::vector<uint64_t> &n)
{
    if (m->broken)
        return (int)m->fail("this adsb_multi handle is unusable: a worker stopped answering earlier (destroy it)");
    m->err.clear();
    m->info = adsb_multi_info{};
    const auto t0 = clk::now();
    const int ns = (int)src.size(), nw = (int)m->w.size();
    m->streams.assign((size_t)ns, StreamResult{});
    const int used = std::min(ns, nw);
    for (int i = 0; i < used; i++) {
        Job j;
        j.kind = JobKind::Stream;
        j.streams = src.data();
        j.stream_n = n.data();
        j.n_streams = ns;
        j.stream_lo = i;
        j.stream_step = nw;
        j.results = m->streams.data();
        post(*m->w[i], j);
    }
    for (int i = 0; i < used; i++) {
        if (!wait_done(*m->w[i]))
            return (int)m->gave_up(*m->w[i]);
        m->info.workers_ms = std::max(m->info.workers_ms, m->w[i]->ms);
    }
    m->info.shards = used;
    m->info.total_ms = ms_since(t0);
    for (int s = 0; s < ns; s++)
        if (m->streams[s].rc)
            return (int)m->fail("stream %d of %d: %s", s, ns, m->streams[s].err.c_str());
    return 0;
}

// No exception crosses the C ABI: a bad_alloc of a plan, gather or result vector (a dense 2 Gi-sample capture gathers
// millions of frames) ends the call with -1 and a message.  Every allocation of the two functions lies outside the
// windows in which a worker runs a job that points into the caller's frame, so returning from the catch is safe.
long decode_sharded(adsb_multi *m, const Source &src, const void *const *slices, int n_slices, uint64_t total,
                    const adsb_frame **frames)
{
    try {
        return decode_sharded_impl(m, src, slices, n_slices, total, frames);
    } catch (const std::exception &e) {
        return m->fail("adsb_multi decode: %s (out of memory?)", e.what());
    }
}
int decode_streams(adsb_multi *m, const std::vector<Source> &src, const std::vector<uint64_t> &n)
{
    try {
        return decode_streams_impl(m, src, n);
    } catch (const std::exception &e) {
        return (int)m->fail("adsb_multi decode of independent streams: %s (out of memory?)", e.what());
    }
}

} // namespace

extern "C" {

adsb_multi *adsb_multi_create(const adsb_config *cfg_in, int n_devices, const int *devices)
{
    if (n_devices <= 0 || n_devices > 64) {
        g_multi_create_error = "adsb_multi_create: n_devices must be 1..64";
        return nullptr;
    }
    adsb_multi *m = new (std::nothrow) adsb_multi();
    if (!m) {
        g_multi_create_error = "out of memory";
        return nullptr;
    }
    if (const char *why = adsb::accept_config(cfg_in, m->cfg, m->dbg)) {
        g_multi_create_error = why;
        delete m;
        return nullptr;
    }
    if (m->cfg.stage_samples) // a piece must fit the staging buffer beside the tail it keeps
        m->piece_samples = std::max<uint64_t>(1u << 15, std::min<uint64_t>(kPieceSamples, m->cfg.stage_samples / 2));
    try {
        for (int i = 0; i < n_devices; i++) {
            std::unique_ptr<Worker> w(new Worker());
            w->index = i;
            w->device = devices ? devices[i] : i;
            w->cfg = m->cfg;
            w->dbg = m->dbg;
            w->piece = m->piece_samples;
            m->w.push_back(std::move(w));
        }
        for (auto &w : m->w)
            w->th = std::thread(worker_main, w.get(), m->piece_samples);
    } catch (const std::exception &e) { // no memory for a worker, no thread to be had: the ones that started wait for their first job
        g_multi_create_error = std::string("adsb_multi_create: ") + e.what();
        for (auto &w : m->w)
            if (w->th.joinable()) {
                std::unique_lock<std::mutex> lk(w->mu);
                w->cv.wait(lk, [&] { return w->done >= 1; });
                w->posted = 1;
                w->posted_a.store(1, std::memory_order_release);
            }
        adsb_multi_destroy(m);
        return nullptr;
    }
    bool ok = true;
    for (auto &w : m->w) {
        std::unique_lock<std::mutex> lk(w->mu);
        if (!w->cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::seconds((long)worker_limit_s(m->cfg) + 60),
                              [&] { return w->done >= 1; })) {
            // (the runtime's start is the one wait the handle's own deadline cannot cover: hipSetDevice / hipStreamCreate block)
            if (ok) {
                char buf[256];
                snprintf(buf, sizeof buf, "device %d (worker %d): the runtime did not come up within %.0f s", w->device, w->index,
                         worker_limit_s(m->cfg) + 60);
                g_multi_create_error = buf;
            }
            ok = false;
            m->broken = true;
            continue;
        }
        w->posted = 1;
        w->posted_a.store(1, std::memory_order_release);
        if (!w->dec && ok) {
            g_multi_create_error = w->err;
            ok = false;
        }
        m->create_ms = std::max(m->create_ms, w->create_ms);
    }
    if (!ok) {
        adsb_multi_destroy(m);
        return nullptr;
    }
    return m;
}

void adsb_multi_destroy(adsb_multi *m)
{
    if (!m)
        return;
    if (m->broken) {
        // some worker never answered: nobody can join it.  Every worker is cut loose instead -- it frees its handle and itself
        // when (if) it comes back -- and the driver's own memory goes now, except what a job in flight may still write
        // (below).  The SOURCE buffers of that job stay the caller's to keep alive (adsb_multi_last_error says so).
        // Order matters: a worker that sees `orphaned` deletes itself, thread object and condition variable included, so the
        // thread is detached FIRST and the notification is sent while the lock is still held -- the last touch of *p here is
        // the unlock (round 6: TSan caught ~Worker racing with the notify_all / detach that used to follow the unlock).
        for (auto &w : m->w) {
            Worker *p = w.release();
            if (!p->th.joinable()) {
                delete p;
                continue;
            }
            p->th.detach();
            std::lock_guard<std::mutex> lk(p->mu);
            p->orphaned = true;
            p->cv.notify_all();
        }
        // A worker that is still inside a job writes its results where the job told it to: the per-stream results
        // (Job::results) and the gather's destination (Job::dst) are this handle's vectors.  They are given up with the
        // workers -- moved to the heap and never freed -- instead of being freed under a thread that may come back
        // (round 5's advisor finding).  A broken handle is a lost device: a few vectors are the smaller loss.  They stay
        // reachable from a list that is itself never destroyed (no static destructor runs under a detached thread).
        {
            struct GivenUp {
                std::vector<StreamResult> streams;
                std::vector<adsb_frame> out, new_frames;
            };
            static std::mutex mu;
            static auto *given_up = new std::vector<std::unique_ptr<GivenUp>>();
            std::unique_ptr<GivenUp> g(new GivenUp{std::move(m->streams), std::move(m->out), std::move(m->new_frames)});
            std::lock_guard<std::mutex> lk(mu);
            given_up->push_back(std::move(g));
        }
        delete m;
        return;
    }
    Job q;
    q.kind = JobKind::Quit;
    for (auto &w : m->w)
        if (w->th.joinable()) {
            post(*w, q);
            w->th.join();
        }
    delete m;
}

int adsb_multi_devices(const adsb_multi *m) { return m ? (int)m->w.size() : 0; }

int adsb_multi_plan(const adsb_multi *m, uint64_t total_samples, uint64_t *g_begin, uint64_t *g_end, uint64_t *first_sample,
                    uint64_t *n_samples)
{
    if (!m || !g_begin || !g_end || !first_sample || !n_samples)
        return -1;
    try {
        std::vector<uint64_t> p[4];
        const int n = plan((int)m->w.size(), total_samples, p);
        uint64_t *dst[4] = {g_begin, g_end, first_sample, n_samples};
        for (int k = 0; k < 4; k++)
            std::memcpy(dst[k], p[k].data(), (size_t)n * sizeof(uint64_t));
        return n;
    } catch (const std::exception &) {
        return -1;
    }
}

long adsb_multi_decode_host(adsb_multi *m, const uint16_t *samples, size_t n, const adsb_frame **frames)
{
    if (!m || (n && !samples))
        return -1;
    Source s;
    s.mem = samples;
    static const uint16_t none = 0;
    if (!samples)
        s.mem = &none;
    return decode_sharded(m, s, nullptr, 0, n, frames);
}

long adsb_multi_decode_file(adsb_multi *m, const char *path, const adsb_frame **frames)
{
    if (!m || !path || !frames)
        return -1;
    const int fd = open(path, O_RDONLY);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) {
        if (fd >= 0)
            close(fd);
        return m->fail("%s: not a readable regular file (each device reads its own slice: pipes go through adsb_push)", path);
    }
    Source s;
    s.fd = fd;
    const long k = decode_sharded(m, s, nullptr, 0, (uint64_t)sb.st_size / 2, frames); // a trailing odd byte is dropped (air.c:239)
    close(fd);
    return k;
}

long adsb_multi_decode_device(adsb_multi *m, uint64_t total_samples, const void *const *slices, int n_slices,
                              const adsb_frame **frames)
{
    if (!m || !slices)
        return -1;
    for (int i = 0; i < n_slices; i++)
        if (!slices[i])
            return m->fail("adsb_multi_decode_device: slice %d is NULL", i);
    return decode_sharded(m, Source{}, slices, n_slices, total_samples, frames);
}

int adsb_multi_decode_streams_host(adsb_multi *m, int n_streams, const uint16_t *const *samples, const size_t *n)
{
    if (!m || n_streams <= 0 || !samples || !n)
        return -1;
    std::vector<Source> src;
    std::vector<uint64_t> len;
    try {
        src.resize((size_t)n_streams);
        len.resize((size_t)n_streams);
    } catch (const std::exception &e) {
        return (int)m->fail("adsb_multi_decode_streams_host: %s", e.what());
    }
    for (int s = 0; s < n_streams; s++) {
        if (n[s] && !samples[s])
            return (int)m->fail("adsb_multi_decode_streams_host: stream %d is NULL", s);
        static const uint16_t none = 0;
        src[s].mem = samples[s] ? samples[s] : &none;
        len[s] = n[s];
    }
    return decode_streams(m, src, len);
}

int adsb_multi_decode_streams_file(adsb_multi *m, int n_streams, const char *const *paths)
{
    if (!m || n_streams <= 0 || !paths)
        return -1;
    std::vector<Source> src;
    std::vector<uint64_t> len;
    try {
        src.resize((size_t)n_streams);
        len.resize((size_t)n_streams);
    } catch (const std::exception &e) {
        return (int)m->fail("adsb_multi_decode_streams_file: %s", e.what());
    }
    int rc = 0;
    for (int s = 0; s < n_streams && rc == 0; s++) {
        struct stat sb;
        src[s].fd = paths[s] ? open(paths[s], O_RDONLY) : -1;
        if (src[s].fd < 0 || fstat(src[s].fd, &sb) != 0 || !S_ISREG(sb.st_mode))
            rc = (int)m->fail("%s: not a readable regular file", paths[s] ? paths[s] : "(null)");
        else
            len[s] = (uint64_t)sb.st_size / 2;
    }
    if (rc == 0)
        rc = decode_streams(m, src, len);
    for (Source &s : src)
        if (s.fd >= 0)
            close(s.fd);
    return rc;
}

long adsb_multi_stream_frames(const adsb_multi *m, int stream, const adsb_frame **frames)
{
    if (!m || !frames || stream < 0 || (size_t)stream >= m->streams.size() || m->streams[stream].rc)
        return -1;
    const auto &f = m->streams[stream].frames;
    *frames = f.empty() ? nullptr : f.data();
    return (long)f.size();
}

int adsb_multi_stream_stats(const adsb_multi *m, int stream, adsb_stats *out)
{
    if (!m || !out || stream < 0 || (size_t)stream >= m->streams.size() || m->streams[stream].rc)
        return -1;
    *out = m->streams[stream].stats;
    return 0;
}

uint16_t *adsb_multi_host_alloc(adsb_multi *m, uint64_t total_samples)
{
    if (!m)
        return nullptr;
    try {
        std::vector<uint64_t> p[4];
        const int n = plan((int)m->w.size(), total_samples, p);
        std::vector<int> devs((size_t)n);
        for (int i = 0; i < n; i++)
            devs[(size_t)i] = m->w[(size_t)i]->device;
        uint16_t *x = adsb_host_alloc_sharded(total_samples, n, p[2].data(), p[3].data(), devs.data());
        if (!x)
            m->fail("adsb_multi_host_alloc: cannot map and page-lock %llu bytes", (unsigned long long)(2 * total_samples));
        return x;
    } catch (const std::exception &e) {
        m->fail("adsb_multi_host_alloc: %s", e.what());
        return nullptr;
    }
}

int adsb_multi_worker_placement(const adsb_multi *m, int worker, adsb_worker_placement *out)
{
    if (!m || !out || worker < 0 || (size_t)worker >= m->w.size())
        return -1;
    const Worker &w = *m->w[(size_t)worker];
    out->device = w.device;
    out->device_node = w.device_node;
    out->thread_bound = w.bound ? 1 : 0;
    out->slice_node = w.slice_node;
    out->local_fraction = w.local_fraction;
    return 0;
}

int adsb_multi_get_stats(const adsb_multi *m, adsb_stats *out)
{
    if (!m || !out || !m->have_stats)
        return -1;
    *out = m->stats;
    return 0;
}

int adsb_multi_worker_profile_sized(const adsb_multi *m, int worker, adsb_profile *out, size_t size)
{
    if (!m || !out || worker < 0 || (size_t)worker >= m->w.size() || !m->w[worker]->dec)
        return -1;
    return adsb_get_profile_sized(m->w[worker]->dec, out, size); // (no job is running: the decode calls return behind their workers)
}

int adsb_multi_get_info(const adsb_multi *m, adsb_multi_info *out)
{
    if (!m || !out)
        return -1;
    *out = m->info;
    out->create_ms = m->create_ms;
    // the reader / gang threads the workers' handles own at this moment (a handle starts them behind its first dense launch and
    // keeps them until adsb_multi_destroy; they sleep while nothing is in flight)
    out->helper_threads = 0;
    if (!m->broken)
        for (const auto &w : m->w) {
            adsb_profile p;
            if (w->dec && adsb_get_profile_sized(w->dec, &p, sizeof p) == 0)
                out->helper_threads += (int32_t)p.host_threads_running;
        }
    return 0;
}

const char *adsb_multi_last_error(const adsb_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

} // extern "C"
