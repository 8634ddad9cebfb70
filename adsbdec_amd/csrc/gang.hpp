// gang.hpp -- more hands for the resolver at the channel's capacity (BASELINE configs[2]: 106 k accepted frames per launch).
//
// What is sequential in the reference's rules is the DECISION: which record is accepted, at which of its offsets, with which
// ts (demod.c:89,99,128,134,141: the greedy chain) and under which deqframe call (air.c:94-99).  Writing the 40-byte frame
// of an accepted record is not: it depends on the record and on the decision, on nothing before it.  So the resolver's
// thread decides -- 16 bytes per accepted frame: {ts, offset, where the record lies} -- and hands blocks of decisions to the
// threads of a FormatGang, which write the frames into the slots the resolver reserved for them in its output array
// (resolver.hpp: Resolver::emit).  A first attempt sent a word per frame through a ring to ONE writer thread and lost
// (every line of a ring changes hands twice: 4-5 ns per frame on the deciding side); blocks of 512 decisions cost the
// deciding side one task per block.  The same threads decide whole batches of tiles AHEAD of the resolver's thread, which
// then only takes the decisions over (Resolver::speculate_tiles, run_calls_tiles_ahead): a task is a function pointer.
//
// Host-only code, no HIP: built and run without a GPU (tests/cpp/resolver_paths.cpp under ThreadSanitizer).
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <cstddef>
#include <pthread.h>
#include <sched.h>

#include "../../include/adsbdec_amd_diag.h"

namespace adsb {

// One accepted frame, decided: ts (demod.c:99), the offset relative to the launch's first, and where the record lies (bytes
// from the stream's first granule, a multiple of 16) | 1 for a 112-bit frame.  Which of the record's offsets it is follows:
// g_rel - the record's own.
struct Decision {
    uint64_t ts;
    uint32_t g_rel, where;
};
static_assert(sizeof(Decision) == 16, "Decision is two 8-byte stores");
constexpr uint64_t kDecMaxStreamBytes = 1ull << 32;
constexpr uint32_t kDecLong = 1u;

// One accepted frame out of a hand-off record, as five aligned 8-byte stores: {g}{ts}{pw, len, bytes 0..2}{bytes 3..10}
// {bytes 11..13, reserved, 0} (the record holds the 14 bytes in w0..w3, the length and the flags behind them, the copies' pw in
// its last two words: scan_kernel_format.h).  Returns the frame's span (demod.c:109,120,123: lidx).
static_assert(sizeof(adsb_frame) == 40 && offsetof(adsb_frame, pw) == 16 && offsetof(adsb_frame, len) == 20 &&
                  offsetof(adsb_frame, frame) == 21 && offsetof(adsb_frame, reserved) == 35,
              "adsb_frame layout");
inline uint64_t write_frame(uint64_t *o, const uint32_t *r, uint32_t sub, uint64_t g, uint64_t ts)
{
    const uint32_t w0 = r[2], w1 = r[3], w2 = r[4], w3 = r[5];
    const uint32_t len = (w3 >> 16) & 0xFFu, fixed = (w3 >> 24) & 1u;
    const uint32_t pw = r[1 + ((4 + sub) & (0u - (uint32_t)(sub != 0)))]; // sub ? r[5 + sub] : r[1], without the branch
    o[0] = g;
    o[1] = ts;
    o[2] = (uint64_t)pw | (uint64_t)len << 32 | (uint64_t)(w0 & 0xFFFFFFu) << 40;
    o[3] = (uint64_t)(w0 >> 24) | (uint64_t)w1 << 8 | (uint64_t)(w2 & 0xFFFFFFu) << 40;
    o[4] = (uint64_t)(w2 >> 24) | (uint64_t)(w3 & 0xFFFFu) << 8 | (uint64_t)fixed << 24; // (tail padding zero: frames are compared and copied as bytes)
    return 80 + 80 * (uint64_t)len;
}

class FormatGang {
public:
    struct Counts { // of the frames written since the last take_counts(): the Ok row (valid.c:53,75), repaired frames
        uint64_t n11 = 0, n17 = 0, nfix = 0;
    };
    struct Task {
        void (*fn)(const Task &, Counts &) = nullptr; // null: format() -- write the frames of n decisions; else the resolver's own (a batch decided ahead: ctx)
        const uint32_t *stream = nullptr; // the launch's granules
        const Decision *dec = nullptr;
        uint32_t n = 0;
        adsb_frame *dst = nullptr; // n slots
        uint64_t g_base = 0;
        uint64_t ts_add = 0; // added to every decision's ts (a batch decided ahead counts its ts from zero: resolver.hpp)
        void *ctx = nullptr;
    };
    static constexpr uint32_t kBlock = 512; // decisions per task

    FormatGang()
    {
        // The ring is initialised HERE and nowhere else: a quiescent ring (everything posted has completed) is consistent
        // with posted_ / claimed_ / completed_ whatever their values, so stop() + start() need not touch it.  (Round 5's
        // start() reset the slots' seq to their indices and left the three counters alone: after 256 tasks the first
        // post() behind a restart waited for ever -- the advisor's finding; tests/cpp/resolver_paths.cpp restarts a gang now.)
        for (uint32_t i = 0; i < kRing; i++)
            ring_[i].seq.store(i, std::memory_order_relaxed);
    }
    FormatGang(const FormatGang &) = delete;
    FormatGang &operator=(const FormatGang &) = delete;
    ~FormatGang() { stop(); }

    bool start(int helpers)
    {
        if (!th_.empty())
            return true;
        acc_.reset(new (std::nothrow) Acc[(size_t)helpers + 1]); // (the last one: the posting thread's own, when it lends a hand)
        if (!acc_)
            return false;
        try {
            for (int i = 0; i < helpers; i++) {
                th_.emplace_back([this, i] { loop(i); });
                pthread_setname_np(th_.back().native_handle(), "adsb-format");
            }
        } catch (...) { // fewer threads than asked for will do; none will not
        }
        if (th_.empty())
            acc_.reset();
        return !th_.empty();
    }
    void stop()
    {
        if (th_.empty())
            return;
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_.store(true);
            cv_.notify_all();
        }
        for (std::thread &t : th_)
            t.join();
        th_.clear();
        quit_.store(false); // (start() may follow)
    }
    size_t helpers() const { return th_.size(); }
    std::vector<std::thread> &threads() { return th_; }

    // A launch begins / has ended: between the two the helpers poll, afterwards they poll on for kSpinUs (a dense launch
    // follows the other) and then sleep.
    void begin()
    {
        active_.store(true); // (seq_cst against `sleepers_`: either the helper sees the flag or this sees the sleeper)
        if (sleepers_.load()) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
    }
    void end() { active_.store(false, std::memory_order_relaxed); }

    void post(const Task &t)
    {
        if (!active_.load(std::memory_order_relaxed))
            begin();
        Slot &s = ring_[posted_ % kRing];
        for (uint32_t spins = 1; s.seq.load(std::memory_order_acquire) != posted_; spins++) // the ring is full: its oldest task has not been taken yet
            relax(spins);
        s.task = t;
        s.seq.store(posted_ + 1, std::memory_order_release);
        posted_++;
    }
    bool busy() const { return completed_.load(std::memory_order_acquire) != posted_; }
    void wait_all()
    {
        for (uint32_t spins = 1; completed_.load(std::memory_order_acquire) != posted_; spins++)
            if (!help())
                relax(spins);
    }
    // The posting thread lends a hand while it has to wait anyway: one task, if one is waiting.  (Its counts go to a slot of
    // its own; only this thread ever calls it.)
    bool help()
    {
        uint64_t c = claimed_.load(std::memory_order_relaxed);
        Slot &s = ring_[c % kRing];
        if (s.seq.load(std::memory_order_acquire) != c + 1 || !claimed_.compare_exchange_strong(c, c + 1, std::memory_order_acq_rel))
            return false;
        const Task t = s.task;
        s.seq.store(c + kRing, std::memory_order_release);
        if (t.fn)
            t.fn(t, acc_[th_.size()].c);
        else
            format(t, acc_[th_.size()].c);
        completed_.fetch_add(1, std::memory_order_release);
        return true;
    }
    // One poll's pause.  Every 256th gives the processor away: should the scheduler ever put two of these threads on one CPU
    // (seen in an 8-CPU VM: 24 ms per launch, the two alternating at the 4 ms tick), they alternate at the speed of a
    // system call instead.
    static void relax(uint32_t spins)
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
        if ((spins & 0xFF) == 0)
            sched_yield();
    }
    // (only with nothing in flight: behind wait_all())
    Counts take_counts()
    {
        Counts c;
        for (size_t i = 0; i <= th_.size(); i++) {
            c.n11 += acc_[i].c.n11, c.n17 += acc_[i].c.n17, c.nfix += acc_[i].c.nfix;
            acc_[i].c = Counts{};
        }
        return c;
    }

    static void format(const Task &t, Counts &c)
    {
        uint64_t *o = reinterpret_cast<uint64_t *>(t.dst);
        uint64_t n11 = 0, n17 = 0, nfix = 0;
        for (uint32_t i = 0; i < t.n; i++, o += 5) {
            const Decision d = t.dec[i];
            const uint32_t *r = t.stream + ((d.where & ~15u) >> 2);
            write_frame(o, r, d.g_rel - r[0], t.g_base + d.g_rel, d.ts + t.ts_add);
            const uint32_t df = (r[2] & 0xFFu) >> 3;
            n11 += df == 11;
            n17 += df == 17;
            nfix += (r[5] >> 24) & 1u;
        }
        c.n11 += n11, c.n17 += n17, c.nfix += nfix;
    }

private:
    static constexpr uint32_t kRing = 256;
    static constexpr int kSpinUs = 400;
    struct alignas(64) Slot {
        std::atomic<uint64_t> seq{0}; // == index: free for task `index`; == index + 1: task `index` is in; then index + kRing
        Task task;
    };
    struct alignas(64) Acc {
        Counts c;
    };

    void loop(int me)
    {
        using clk = std::chrono::steady_clock;
        auto t_idle = clk::now();
        for (uint32_t spins = 1;; spins++) {
            uint64_t c = claimed_.load(std::memory_order_relaxed);
            Slot &s = ring_[c % kRing];
            if (s.seq.load(std::memory_order_acquire) == c + 1) {
                if (!claimed_.compare_exchange_weak(c, c + 1, std::memory_order_acq_rel))
                    continue;
                const Task t = s.task;
                s.seq.store(c + kRing, std::memory_order_release);
                if (t.fn)
                    t.fn(t, acc_[me].c);
                else
                    format(t, acc_[me].c);
                completed_.fetch_add(1, std::memory_order_release);
                t_idle = clk::now();
                continue;
            }
            if (quit_.load(std::memory_order_relaxed))
                return;
            relax(spins);
            if ((spins & 0xFF) == 0 && !active_.load(std::memory_order_relaxed) && clk::now() - t_idle > std::chrono::microseconds(kSpinUs)) {
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1);
                cv_.wait(lk, [&] { return quit_.load() || active_.load(); });
                sleepers_.fetch_sub(1);
                t_idle = clk::now();
            }
        }
    }

    std::vector<std::thread> th_;
    std::unique_ptr<Acc[]> acc_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<bool> quit_{false};
    Slot ring_[kRing];
    alignas(64) uint64_t posted_ = 0;                 // ---- the resolver's line
    alignas(64) std::atomic<uint64_t> claimed_{0};    // ---- the helpers'
    alignas(64) std::atomic<uint64_t> completed_{0};
    alignas(64) std::atomic<bool> active_{false};     // ---- written at a launch's ends
    alignas(64) std::atomic<int> sleepers_{0};
};

} // namespace adsb
