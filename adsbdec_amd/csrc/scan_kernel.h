// scan_kernel.h -- launch interface of the fused scan kernel (scan_kernel.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adsb {

constexpr int kRun = 28;       // power samples per thread run (4 x 7: see scan_kernel.hip)
constexpr int kThreads = 256;  // 4 wavefronts
constexpr int kPasses = 2;     // runs per thread
constexpr int kTileA = kRun * kThreads * kPasses; // 14336 power samples staged in LDS
constexpr int kHalo = 1204;    // >= ADSB_WINDOW (1196), multiple of 28
constexpr int kTileG = kTileA - kHalo; // 13132 offsets owned by one workgroup
constexpr int kCandWords = 6;  // {g_rel, pw, frame[0..13] | len<<16 in the last word}
constexpr uint64_t kMaxLaunchOffsets = (1ull << 30) - kTileG; // g_rel must fit 30 bits

static_assert(kTileG % 28 == 0, "tiles must start on a multiple of 28 power samples");
static_assert(kHalo >= 1196, "halo must cover one long-frame evaluation");

struct ScanArgs {
    const uint32_t *x;   // (I,Q) pairs; x[0] is stream pair index pbuf0 (16-byte aligned, pbuf0 % 4 == 0)
    int64_t pbuf0;
    int64_t p_lo, p_hi;  // stream pair indices present in the buffer: [p_lo, p_hi)
    uint64_t g_begin;    // first offset to evaluate, multiple of 28
    uint64_t g_end;      // one past the last offset
    int df18;            // demod.c:26
    uint32_t *counters;  // [0] candidates, [1] tries (may exceed the capacities)
    uint32_t *cands;     // kCandWords dwords per record
    uint32_t cand_cap;
    uint32_t *tries;     // (g_rel << 2) | code
    uint32_t try_cap;
};

hipError_t launch_scan(const ScanArgs &args, bool stats, hipStream_t stream);

} // namespace adsb
