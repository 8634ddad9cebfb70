// scan_stamps.h -- per-phase clocks of a tile, for a MEASUREMENT build of scan_kernel.hip only:
//     tools/build_variant.sh stamps -DADSB_PHASE_STAMPS   ->   tools/phase_probe.py   ->   profiles/r6_phase_stamps.txt
// Thread 0 of every tile adds the ticks of the device's 100 MHz clock between phase boundaries to one accumulator per
// phase (where does a tile's life go on the dense captures?).  In the shipped build every macro below is empty and the
// library exports no adsb_debug_phase_read (tests/test_build_flags.py checks both): this header is the one build knob the
// kernel source has.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef ADSB_PHASE_STAMPS
namespace adsb {
// One row of accumulators per tile (tile index mod kStampRows): a first version added every tile's clocks to ONE row with
// device-scope atomics -- 14 same-address atomics per tile serialise in the L2, and the kernel it measured ran 2.5 x slower
// than the shipped one.  Rows are private to a tile (until the index wraps: launches of up to 4 096 tiles), the adds are
// uncontended, and the host sums the rows.
constexpr int kStampRows = 4096, kStampCols = 16;
__device__ unsigned long long g_phase[kStampRows * kStampCols];
}
#define ADSB_STAMP_ROW (adsb::g_phase + (size_t)(blockIdx.x % adsb::kStampRows) * adsb::kStampCols)
#define ADSB_STAMP_BEGIN()                                       \
    uint64_t stamp_last = __builtin_amdgcn_s_memrealtime();      \
    const uint64_t stamp_first = stamp_last
#define ADSB_STAMP(i)                                                              \
    do {                                                                           \
        if (tid == 0) {                                                            \
            const uint64_t now_ = __builtin_amdgcn_s_memrealtime();                \
            atomicAdd(&ADSB_STAMP_ROW[i], (unsigned long long)(now_ - stamp_last));     \
            stamp_last = now_;                                                     \
        }                                                                          \
    } while (0)
#define ADSB_COUNT(i, v)                                               \
    do {                                                               \
        if (tid == 0)                                                  \
            atomicAdd(&ADSB_STAMP_ROW[i], (unsigned long long)(v));         \
    } while (0)
#define ADSB_STAMP_END(i)                                                                          \
    do {                                                                                           \
        __syncthreads();                                                                           \
        ADSB_STAMP(i);                                                                             \
        if (tid == 0)                                                                              \
            atomicMax(&ADSB_STAMP_ROW[14], (unsigned long long)(stamp_last - stamp_first)); /* the longest tile of the row */ \
    } while (0)
// the accumulators summed over the rows (entry 14: the maximum), and (reset != 0) back to zero
extern "C" int adsb_debug_phase_read(unsigned long long *out, int n, int reset)
{
    static unsigned long long v[adsb::kStampRows * adsb::kStampCols];
    if (hipMemcpyFromSymbol(v, HIP_SYMBOL(adsb::g_phase), sizeof v) != hipSuccess)
        return -1;
    for (int i = 0; i < n && i < adsb::kStampCols; i++) {
        unsigned long long acc = 0;
        for (int r = 0; r < adsb::kStampRows; r++)
            acc = i == 14 ? (v[r * adsb::kStampCols + i] > acc ? v[r * adsb::kStampCols + i] : acc) : acc + v[r * adsb::kStampCols + i];
        out[i] = acc;
    }
    if (reset) {
        for (auto &x : v)
            x = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(adsb::g_phase), v, sizeof v) != hipSuccess)
            return -1;
    }
    return 0;
}
#else
#define ADSB_STAMP_BEGIN() uint64_t stamp_last = 0
#define ADSB_STAMP(i) do { (void)stamp_last; } while (0)
#define ADSB_COUNT(i, v) do { } while (0)
#define ADSB_STAMP_END(i) do { } while (0)
#endif
