// scan_kernel_format.h -- the format of the device -> host hand-off stream (ScanArgs::hand), shared by the kernel that
// writes it (scan_kernel.hip) and the host code that reads it (handoff.hpp).  The reading side is built without HIP too
// (tests/cpp/handoff_tsan.cpp), hence the one macro: the functions below are device code only when a device compiler is
// looking.
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define ADSB_HD __host__ __device__
#else
#define ADSB_HD
#endif

namespace adsb {

// hand-off stream (ScanArgs::hand): 16-byte granules
constexpr int kGranuleWords = 4;
constexpr uint32_t kMarkOver = 0x10000u;  // marker flag: some records of the tile are on the loose list
constexpr uint32_t kMarkNoFit = 0x20000u; // marker flag: the tile's range ran past the array (records are loose)
constexpr uint32_t kMarkTries = 0x40000u; // marker flag (statistics runs): the tile's tries went through the launch-wide list -- its
                                          // records are all here, but the count pass needs the launch's counters (nothing is held up)
constexpr int kMarkLinesShift = 19;       // marker word 1, bits 19..31: 64-byte lines the tile reserved (it may keep fewer records
                                          // than it reserved for: the host skips to the next tile's marker by this)
ADSB_HD constexpr uint32_t marker_granules(uint32_t nf) { return (nf >> kMarkLinesShift) * 4u; }
// Granules a tile with n records reserves: marker + 2 n, rounded up to whole 64-byte lines, so
// that the host never reads (and caches) a line the device has yet to write another tile into
// -- every later device write to such a line has to pull it out of the CPU's cache first.
ADSB_HD constexpr uint32_t stream_granules(uint32_t n) { return (1u + 2u * n + 3u) & ~3u; }

// A record of the stream is two granules {g_rel, pw, w0, w1}{w2, w3 | len << 16 | flags << 24, pw', pw''}.  flags bit 0: repaired by
// the 1-bit extension.  flags bits 1..2: the record stands for 1 + that many candidates -- the SAME frame bytes decoded at
// the consecutive offsets g_rel, g_rel + 1 (pw'), g_rel + 2 (pw''): the half-sample shifted copies of one frame, which the
// tile cannot prove unreachable when frames stand back to back (demod.c:125-141 decides which copy the scan lands on) and
// which would otherwise cost three records per frame on a full channel.  Records of the loose list (6 words) never carry copies.
constexpr int kRecCopiesShift = 25;
ADSB_HD inline uint32_t rec_copies(const uint32_t *r) { return 1u + ((r[5] >> kRecCopiesShift) & 3u); }
ADSB_HD inline uint32_t rec_pw(const uint32_t *r, uint32_t k) { return k ? r[5 + k] : r[1]; }

// Check words of a tile marker (device writes them, host checks them).  Two independent summaries of the records behind
// the marker go in:
//   a0..a3  the XOR, word by word, of the 2n record granules;
//   sum     the rank-weighted sum of record_term() over the n records (mod 2^32).
// gen differs between any two launches that can touch the same bytes, so a marker written by an earlier launch never
// validates.  A range in which ONE granule has not landed yet always fails (its difference shows in the XOR); stale
// granules whose differences happen to cancel pairwise in the XOR -- the previous launch's records of the same frames
// at shifted offsets: same frame words, only g_rel / pw differ, and differ alike -- are what the sum is for: a weight
// per rank under ordinary addition does not cancel where XOR does.  For bytes that are unrelated to the new ones the
// check passes with probability 2^-64 per look.
ADSB_HD inline uint32_t record_term(uint32_t rank, uint32_t g_rel, uint32_t pw)
{
    return (2u * rank + 1u) * (g_rel ^ (pw << 13 | pw >> 19));
}
ADSB_HD inline void marker_check(uint32_t tile, uint32_t nf, uint32_t gen, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3,
                                 uint32_t sum, uint32_t &lo, uint32_t &hi)
{
    lo = a0 ^ (a2 << 16 | a2 >> 16) ^ gen ^ tile ^ (nf << 11 | nf >> 21) ^ sum;
    hi = a1 ^ (a3 << 16 | a3 >> 16) ^ ~gen ^ (tile << 7 | tile >> 25) ^ nf ^ (sum << 16 | sum >> 16);
}

} // namespace adsb
