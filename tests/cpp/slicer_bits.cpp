// slicer_bits.cpp -- the slicer's word-wide column gather (csrc/slicer_bits.h: the function the kernel runs, compiled for
// the host) against the definition of demod.c:31-44,109: frame bit k of the candidate at offset g is D[g + 80 + 10 k],
// column c collects k = 14 b + c into bit b.  Every sj, random planes (dense, sparse, all ones), run by tests/test_host_logic.py.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../adsbdec_amd/csrc/slicer_bits.h"

int main()
{
    std::mt19937_64 rng(20261003);
    std::vector<uint32_t> tab(adsb::kColMaskWords);
    adsb::make_colmask_table(tab.data());
    // the table itself: row r holds exactly 14 bits, at stream positions r + 10 c
    for (int r = 0; r < 28; r++) {
        int bits = 0;
        for (int q = 0; q < adsb::kColMaskRow; q++) {
            if (tab[r * adsb::kColMaskRow + q] >> 28) {
                printf("mask word beyond 28 bits (r %d, q %d)\n", r, q);
                return 1;
            }
            bits += __builtin_popcount(tab[r * adsb::kColMaskRow + q]);
        }
        if (bits != 14 || tab[r * adsb::kColMaskRow + 6] || tab[r * adsb::kColMaskRow + 7]) {
            printf("mask row %d has %d bits\n", r, bits);
            return 1;
        }
    }
    // ... and the masks computed without it (what the kernel does) are its rows
    for (uint32_t r = 0; r < 28; r++) {
        uint32_t m[6];
        adsb::column_masks(r, m);
        for (int q = 0; q < 6; q++)
            if (m[q] != tab[r * adsb::kColMaskRow + q]) {
                printf("column_masks(%u)[%d] = %08x, table %08x\n", r, q, m[q], tab[r * adsb::kColMaskRow + q]);
                return 1;
            }
    }
    const int kWords = 64;
    long checked = 0;
    for (int it = 0; it < 20000; it++) {
        std::vector<uint32_t> pl(kWords);
        const int kind = it % 5;
        for (auto &w : pl) {
            uint32_t v = (uint32_t)rng();
            if (kind == 1)
                v &= (uint32_t)rng() & (uint32_t)rng(); // sparse
            else if (kind == 2)
                v |= (uint32_t)rng() | (uint32_t)rng(); // dense
            else if (kind == 3)
                v = 0x0FFFFFFFu;
            else if (kind == 4)
                v = (it & 1) ? 0x05555555u : 0x0AAAAAAAu;
            w = v & 0x0FFFFFFFu;
        }
        auto D = [&](int pos) { return (pl[pos / 28] >> (pos % 28)) & 1u; }; // stream position -> bit
        for (int v0 = 0; v0 < 12; v0 += 5)                                   // the candidate's run
            for (int sj = 0; sj < 28; sj++) {
                uint32_t cw[4];
                adsb::gather_columns(pl.data() + v0, sj, cw);
                for (int c = 0; c < 14; c++) {
                    uint32_t want = 0;
                    for (int b = 0; b < 8; b++)
                        want |= D(28 * v0 + sj + 80 + 10 * (14 * b + c)) << b;
                    const uint32_t got = (cw[c >> 2] >> (8 * (c & 3))) & 0xFFu;
                    if (got != want) {
                        printf("it %d v0 %d sj %d column %d: got %02x want %02x\n", it, v0, sj, c, got, want);
                        return 1;
                    }
                    checked++;
                }
                if (cw[3] >> 16) {
                    printf("cw[3] carries bits beyond column 13\n");
                    return 1;
                }
            }
    }
    printf("ok: %ld column bytes\n", checked);
    return 0;
}
