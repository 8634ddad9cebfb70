// scan_kernel.h -- launch interface of the fused scan kernel (scan_kernel.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel_format.h" // the hand-off stream's format: shared with host-only code

namespace adsb {

// Launch shape (what every measurement of DESIGN.md was taken with; the experiments that set other values are history)

constexpr int kSleepStagger = 90;  // s_sleep units (64 cycles) between the starts of a CU's first five workgroups (0 / 45 / 70 / 110: +6.2 / +1.4 / +1.1 / +1.5 %, r6_ab_runs.txt 9a)
constexpr int kMinWaves = 5;       // __launch_bounds__ second argument: waves per SIMD (96 VGPRs; 5 workgroups' LDS fit a CU)

constexpr int kRun = 28;        // power samples per thread run (4 x 7: see scan_kernel.hip)
constexpr int kThreads = 256;   // 4 wavefronts
constexpr int kWaves = kThreads / 64;
constexpr int kWaveRuns = 63;   // distinct runs per wave and pass (lane 63 re-computes the next wave's first run)
constexpr int kPassRuns = kWaves * kWaveRuns;   // 252
constexpr int kReachRuns = 44;  // runs of bit planes one long-frame evaluation reaches ahead: ceil((27+1195)/28)
constexpr int kMaxPasses = 32;
constexpr int kQueueCap = 4 * kThreads; // survivors compacted per round
constexpr int kPlanePad = 8;
constexpr int kClistCap = 256;  // CRC-valid candidates staged per tile for the never-visited filter: one per thread (a tile of 48 k
                                // offsets full of 112-bit frames packed back to back decodes at ~130-190 offsets: BASELINE configs[2])
constexpr int ADSB_DECOFFSET_K = 1200; // longest span an accepted frame jumps (adsbdec.h:3)
constexpr int kCandWords = 6;   // {g_rel, pw, frame[0..13] | len<<16 in the last word}
constexpr int kSyndWords = 14 * 256;
constexpr int kFixSlots = 512;
constexpr int kCounterWords = 8; // the launch counters as the host sees them (ScanArgs::report)
// On the device every counter has a 128-byte line of its own (ScanArgs::counters[i * kCounterPad]; the two
// 64-bit profile maxima are counters 4 and 5).  All of a launch's tiles hit them with device-scope atomics.
constexpr int kCounterPad = 32;
constexpr int kDevCounterWords = 6 * kCounterPad;
constexpr int owned_runs(int passes) { return kPassRuns * passes - kReachRuns; }
constexpr int tile_offsets(int passes) { return kRun * owned_runs(passes); }
constexpr size_t lds_bytes(int passes)
{
    return sizeof(uint32_t) * (size_t)(3 * (kPassRuns * passes + kPlanePad) + kQueueCap + 16 + kClistCap * 6);
}
// Tile geometry of a launch.  A tile takes K = `passes` passes -- except that a launch MAY end in tiles of kTaperPasses: the
// tiles from index `big_tiles` on (0: none).  Why one would: a launch costs ~11 us beyond what its tiles cost at the steady
// rate (time = 10.7 us + 46 ns x tiles, profiles/r6_ab_runs.txt section 10), half of it the drain -- the last tiles run on
// CUs that are emptying, a wave to a SIMD, at a fraction of the issue rate -- and a tile of four passes drains in about half
// the time of one of seven.  Why it is not the default: choose_big_tiles() below.  Every user of the geometry -- the kernel,
// the count pass, the host's walk of the hand-off stream -- goes through these three functions (tests/cpp/tile_geometry.hip).
constexpr int kTaperPasses = 4;
__host__ __device__ inline int tile_passes(uint32_t tile, uint32_t big_tiles, int k)
{
    return (big_tiles == 0 || tile < big_tiles || k <= kTaperPasses) ? k : kTaperPasses;
}
__host__ __device__ inline uint64_t tile_first_run(uint32_t tile, uint32_t big_tiles, int k)
{
    if (big_tiles == 0 || tile <= big_tiles || k <= kTaperPasses)
        return (uint64_t)tile * (uint64_t)owned_runs(k);
    return (uint64_t)big_tiles * (uint64_t)owned_runs(k) + (uint64_t)(tile - big_tiles) * (uint64_t)owned_runs(kTaperPasses);
}
inline uint32_t tile_count(uint64_t n_offsets, uint32_t big_tiles, int k)
{
    const uint64_t runs = (n_offsets + kRun - 1) / kRun;
    const uint64_t head = (uint64_t)big_tiles * (uint64_t)owned_runs(k);
    if (big_tiles == 0 || k <= kTaperPasses || runs <= head)
        return (uint32_t)((runs + owned_runs(k) - 1) / owned_runs(k));
    return big_tiles + (uint32_t)((runs - head + owned_runs(kTaperPasses) - 1) / owned_runs(kTaperPasses));
}

// Host: how many of a launch's tiles take all K passes -- 0, all of them, unless `forced` > 0 (adsb_debug_config.big_tiles; the
// tests and tools/ab_interleaved.py).  The tail of small tiles is NOT shipped: 640 of them (cus x kMinWaves / 2) make the
// kernel 1.5-2 % faster -- 405 / 512 / 634 tiles of four passes -1.8 / -1.0..-2.0 / -1.4..-2.4 %, of two or three passes
// -1.6..-1.9 % at best, of five nothing -- and the CALL 3 % slower (adsb_decode_device 0.1512 -> 0.1563 ms; 256 small tiles
// +2.3 %, 128 +1.6 %; configs[2] +10 %): the host walks the hand-off stream tile by tile and was already the later of the
// two to finish; small tiles arrive 1.75 x as fast exactly where it has to catch up (profiles/r6_ab_runs.txt section 10).
inline uint32_t choose_big_tiles(uint64_t n_offsets, int passes, int cus, int forced)
{
    (void)n_offsets, (void)cus;
    return (passes > kTaperPasses && forced > 0) ? (uint32_t)forced : 0u;
}

constexpr uint64_t kMaxLaunchOffsets = (1ull << 30) - tile_offsets(kMaxPasses); // g_rel must fit 30 bits

struct ScanArgs {
    const uint32_t *x;   // (I,Q) pairs; x[0] is stream pair index pbuf0 (16-byte aligned, pbuf0 % 4 == 0)
    int64_t pbuf0;
    int64_t p_lo, p_hi;  // stream pair indices present in the buffer: [p_lo, p_hi)
    uint64_t g_begin;    // first offset to evaluate, multiple of 28
    uint64_t g_end;      // one past the last offset
    int df18;            // demod.c:26
    int passes;          // K: runs per thread; a tile owns owned_runs(K) runs
    uint32_t big_tiles;  // tiles from this index on take kTaperPasses passes (0: every tile takes K; tile_passes)
    int queue_cap;       // survivors compacted per round: 256..kQueueCap (kQueueCap unless testing)
    int all_candidates;  // 1: emit every CRC-valid offset (no never-visited filter)
    int clist_cap;       // CRC-valid candidates staged per tile: 1..kClistCap (kClistCap unless testing)
    // Streaming hand-off (hand == null: off).  `hand` is ONE stream of 16-byte granules
    // that the host reads strictly sequentially while the kernel runs.  A tile reserves
    // stream_granules(n) consecutive granules (whole 64-byte lines) with one atomicAdd on
    // counters[2] (so ranges appear in tile COMPLETION order) and writes
    //     marker  {tile, n | flags, check_lo, check_hi}
    //     n x     {g_rel, pw, w0, w1} {w2, w3 | len << 16 | flags << 24, pw', pw''}   (ascending g_rel; pw', pw'': the copies'
    //                                                                                pw of a record that stands for a run of copies:
    //                                                                                scan_kernel_format.h)
    // Nothing orders these stores on their way to host memory, so the marker carries
    // marker_check() of the records: the host consumes a tile only when the marker and
    // the XOR of the 2n granules behind it agree (gen changes with every launch, so
    // stale bytes never validate).  A tile whose range does not fit writes no records
    // there (they go to the loose list) and says so in its marker, if that fits.
    uint32_t *hand;
    uint32_t hand_cap;     // granules
    uint32_t gen;
    const uint32_t *fix_tab; // EXTENSION (not in the reference): 512-entry perfect hash syndrome -> bit, or null
    uint32_t fix_mul;
    const uint32_t *synd; // [14][256] CRC-24 syndrome table (make_syndrome_table)
    // Device counters, zero at launch: [0] loose candidates, [1] tries (may exceed the
    // capacities), [2] hand-off granules, [3] unused; with `profile` [4..5] max over tiles of
    // ~(start) and [6..7] max of end on the device's 100 MHz clock (64-bit).  launch_scan
    // puts one wave behind the scan (report_kernel) that writes them to `report` (pinned
    // host, same layout, word [3] = gen; null: no report) and zeroes them.
    uint32_t *counters;
    uint32_t *report;
    int profile;
    uint32_t *cands;     // kCandWords dwords per record
    uint32_t cand_cap;
    // Tries (statistics runs): words (g_rel << 2) | code.  With try_counts (device-resident counting of a
    // stream): tile t's rounds write try_counts[t] words into tries[t * kTryRegion ..], and only what does not fit
    // the region is appended to the launch-wide list tries[try_list_first ..] (try_cap words, counted in
    // counters[1]).  Without (try_counts == null, try_list_first == 0): everything goes to the list.
    uint32_t *tries;
    uint32_t try_cap;
    uint32_t *try_counts;
    uint32_t try_list_first;
};
constexpr int kTryRegion = 4 * kQueueCap; // try words per tile region: 8.5 % of a K = 7 tile's offsets (the adversarial capture has 7.6 %)

// Second, tiny kernel of statistics runs (valid.c:46,68 count a Try only for VISITED
// offsets): the try words of a launch stay on the device; once the host has
// resolved which frames were accepted, every try below `hi` (the resolver's position)
// is looked up in the sorted accepted frames -- inside (g, g+span) of one means the
// greedy scan jumped over it -- and three device-side counters accumulate.  Tries at or
// beyond `hi` are not decided yet and are carried to the next pass.  Nothing comes back to
// the host until statistics are asked for: a pass is enqueued and forgotten.
struct TryFrame { // an accepted frame as the count kernel needs it (one upload per pass)
    uint64_t g;
    uint32_t span, pad;
};
struct TryCountArgs {
    const uint32_t *tries;     // this launch's launch-wide list: (g_rel << 2) | code
    uint32_t n_tries;
    const uint32_t *regions;   // ... and its per-tile regions (ScanArgs::try_counts), or null
    const uint32_t *region_counts;
    uint32_t n_tiles;
    int passes;                // the launch's tile geometry (tile_first_run)
    uint32_t big_tiles;
    uint64_t g_base;
    const uint64_t *carry_in;  // undecided tries of earlier passes: (g << 2) | code
    const uint32_t *n_carry;   // device: how many (left there by the previous pass)
    const TryFrame *frames;    // accepted frames, ascending
    uint32_t n_frames;
    uint64_t hi;
    int final;                 // 1: end of stream -- tries >= hi are dropped (never visited)
    uint64_t *carry_out;
    uint32_t carry_cap;
    uint32_t *n_carry_out;     // device, zero at launch: entries appended to carry_out (may exceed carry_cap)
    uint32_t *n_carry_next;    // device: the count the NEXT pass will append to; this pass zeroes it
    unsigned long long *acc;   // device: [0..2] visited tries per DF code, accumulated over the stream's passes;
                               // [3] != 0: a carry list overflowed.  The host reads it when statistics are asked for.
};
hipError_t launch_count_tries(const TryCountArgs &args, hipStream_t stream);

// Host: fill the 14 x 256 syndrome table (crc.h generator 0xFFF409).
void make_syndrome_table(uint32_t *out /* kSyndWords */);
// Host: perfect hash of the single-bit syndromes of bits 5..111 of a long frame:
// tab[(syn * mul) >> 23] = (syn << 8) | bit. Returns the multiplier.
uint32_t make_fix_table(uint32_t *tab /* kFixSlots */);
// Host: choose the passes-per-tile for a launch of n_offsets on a device with `cus` compute units
// (balances halo overhead against tail quantisation).
int choose_passes(uint64_t n_offsets, int cus, bool dense = false);

hipError_t launch_scan(const ScanArgs &args, bool stats, hipStream_t stream);
// Device-to-device copy of n uint16 samples by the library's own kernel (the staging tail: see scan_kernel.hip).
hipError_t launch_copy_samples(uint16_t *dst, const uint16_t *src, size_t n, hipStream_t stream);

} // namespace adsb
