// slicer_bits.h -- the PPM slicer's bit gather (demod.c:31-44 getabyte: 8 strict '>' compares per byte, 14 bytes), as
// word-wide logic on the D plane.  Host + device code, no HIP type in it: the kernel (scan_kernel.hip stage_b) and a CPU
// test (tests/cpp/slicer_bits.cpp, against the definition bit by bit) compile the same function.
//
// The D plane holds one bit per power sample, D[m] = a[m] > a[m+5], 28 to a word (bits 28..31 zero).  For a candidate at
// offset g = 28 v + sj the frame's bit k (k = 0 is the MSB of byte 0) is D[g + 80 + 10 k] (demod.c:34,109: data starts 80
// samples after the preamble's first, a bit every 10).  The CRC's syndrome table and the record want the frame as 14
// COLUMN bytes: column c = bits k = 14 b + c, b = 0..7, bit b of the byte.  Because 140 = 14 x 10 = 5 x 28, bit (b, c)
// sits at the same bit position as (0, c), five words further per b.
//
// Until round 5 the slicer picked the 112 bits one by one (a dependent LDS read, a shift and a funnel shift per bit: ~310
// VALU instructions and 112 LDS reads per candidate).  Here: the 41 words w[0..40] that hold them (row b = words 5b ..
// 5b+5) are ANDed with the row's column mask (the bits r + 10 c, r = (sj + 80) mod 28: column_masks()), four
// rows are merged into one word per word column with shifts of 0..3 (a nibble per column: the masks are 10 bits apart),
// the two groups of rows are laid into one linear bit string at the compile-time offsets 28 q and 28 q + 4, one funnel
// shift by r makes every column byte start at the FIXED bit 10 c, and 14 field extracts finish.  ~150 VALU, 41 LDS reads.
#pragma once

#include <stdint.h>

#include "scan_kernel_format.h" // ADSB_HD

namespace adsb {

constexpr int kColMaskRow = 8;                 // words per row of the table (6 used: two 16-byte loads)
constexpr int kColMaskWords = 28 * kColMaskRow;

// Host: tab[r * 8 + q], q = 0..5 = the bits of word q (28 valid bits per word) at stream positions r + 10 c, c = 0..13
inline void make_colmask_table(uint32_t *tab)
{
    for (int r = 0; r < 28; r++) {
        for (int q = 0; q < kColMaskRow; q++)
            tab[r * kColMaskRow + q] = 0;
        for (int c = 0; c < 14; c++) {
            const int pos = r + 10 * c;
            tab[r * kColMaskRow + pos / 28] |= 1u << (pos % 28);
        }
    }
}

// ({hi, lo} >> s) & 0xFFFFFFFF for s in 0..31: one v_alignbit_b32 on the device
ADSB_HD inline uint32_t funnel_right(uint32_t hi, uint32_t lo, uint32_t s)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (s & 31u));
}

// The six mask words of row r WITHOUT a table: the pattern of bits 10 c (c = 0..13: a 131-bit constant in five dwords)
// shifted left by r, cut into words of 28 bits.  17 instructions on the device -- against a dependent load in front of
// the 41 plane reads (from global memory: an L2 round trip; from LDS: 900 bytes per workgroup, which cost the kernel its
// fifth workgroup per CU: +3 %, profiles/r6_ab_runs.txt).
ADSB_HD inline void column_masks(uint32_t r, uint32_t (&m)[6])
{
    constexpr uint32_t B[5] = {0x40100401u, 0x10040100u, 0x04010040u, 0x01004010u, 0x00000004u}; // bits 0, 10, .., 130
    uint32_t S[6]; // the pattern << r (r <= 27: bit 157 at most)
    S[0] = B[0] << r;
#pragma unroll
    for (int d = 1; d < 6; d++) {
        const uint64_t pair = ((uint64_t)(d < 5 ? B[d] : 0u) << 32) | B[d - 1];
        S[d] = (uint32_t)((pair << r) >> 32);
    }
    m[0] = S[0] & 0x0FFFFFFFu;
#pragma unroll
    for (int q = 1; q < 6; q++) { // bits [28 q, 28 q + 28): they start at bit 32 - 4 q of dword q - 1
        const int sh = 32 - 4 * q;
        m[q] = funnel_right(S[q], S[q - 1], (uint32_t)sh) & 0x0FFFFFFFu;
    }
}

// dcol: the D plane from the candidate's own run on (dcol[0] holds offset g - sj); sj = 0..27.
// cw[j] = columns 4j .. 4j+3, one byte each (column c in byte c & 3 of cw[c >> 2]); cw[3] bits 16..31 are zero.
template <class Plane>
ADSB_HD inline void gather_columns(Plane dcol, int sj, uint32_t (&cw)[4])
{
    const int P = sj + 80;                 // 80 .. 107
    const int w0 = P >= 84 ? 3 : 2;        // P / 28
    const uint32_t r = (uint32_t)(P - 28 * w0);
    uint32_t m[6];
    column_masks(r, m);
    // rows 0..3 -> za, rows 4..7 -> zb: bit (b & 3) of the nibble that starts at the column's bit position
    uint32_t za[6], zb[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
        uint32_t a = dcol[w0 + q] & m[q];
        uint32_t b = dcol[w0 + q + 20] & m[q];
#pragma unroll
        for (int k = 1; k < 4; k++) {
            a |= (dcol[w0 + q + 5 * k] & m[q]) << k;
            b |= (dcol[w0 + q + 20 + 5 * k] & m[q]) << k;
        }
        za[q] = a, zb[q] = b;
    }
    // one linear bit string: za[q] at bit 28 q, zb[q] at bit 28 q + 4 (a word's nibbles may reach bit 30: they land in the
    // next word's lowest bits, where no column of that word can start -- columns are 10 bits apart and a byte is 8)
    uint32_t L[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; q++) {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const uint32_t v = half ? zb[q] : za[q];
            const int off = 28 * q + 4 * half, d = off >> 5, sh = off & 31;
            L[d] |= v << sh;
            if (sh != 0)
                L[d + 1] |= v >> (32 - sh);
        }
    }
    // ... shifted right by r: column c's byte is bits 10 c .. 10 c + 7
    uint32_t R[5];
#pragma unroll
    for (int d = 0; d < 5; d++)
        R[d] = funnel_right(L[d + 1], L[d], r);
    cw[0] = cw[1] = cw[2] = cw[3] = 0;
#pragma unroll
    for (int c = 0; c < 14; c++) {
        const int pos = 10 * c, d = pos >> 5, sh = pos & 31;
        uint32_t col = R[d] >> sh;
        if (sh > 24)
            col |= R[d + 1] << (32 - sh);
        cw[c >> 2] |= (col & 0xFFu) << (8 * (c & 3));
    }
}

} // namespace adsb
