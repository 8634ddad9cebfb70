// tile_geometry.hip -- host-side check of the tile geometry the kernel and the host must
// agree on (scan_kernel.h: tile_passes / tile_first_run / tile_count), with and without a tail of small tiles,
// and of the host's choice of that tail (choose_big_tiles).
// Built with hipcc and run on the CPU by tests/test_host_logic.py.
#include <cstdio>
#include <random>

#include "../../adsbdec_amd/csrc/scan_kernel.h"

int main()
{
    std::mt19937_64 rng(7);
    for (int it = 0; it < 20000; it++) {
        const int k = 2 + (int)(rng() % 6);
        const uint32_t stagger = (rng() & 1) ? (uint32_t)(rng() % 3000) : 0u; // (big_tiles: tiles from this index on are small)
        const uint64_t n = 1 + rng() % (it % 3 ? 3000000ull : 400000000ull);
        const uint32_t tiles = adsb::tile_count(n, stagger, k);
        const uint64_t runs = (n + adsb::kRun - 1) / adsb::kRun;
        // the tiles cover all runs, and the last one is needed
        if (adsb::tile_first_run(tiles, stagger, k) < runs || (tiles && adsb::tile_first_run(tiles - 1, stagger, k) >= runs)) {
            printf("tile_count(%llu, %u, %d) = %u does not bracket %llu runs\n", (unsigned long long)n, stagger, k, tiles,
                   (unsigned long long)runs);
            return 1;
        }
        // consecutive tiles abut: first_run(t+1) - first_run(t) == owned_runs(passes of t)
        for (int s = 0; s < 50; s++) {
            const uint32_t t = (uint32_t)(rng() % (tiles + 8));
            const uint64_t d = adsb::tile_first_run(t + 1, stagger, k) - adsb::tile_first_run(t, stagger, k);
            const int kt = adsb::tile_passes(t, stagger, k);
            const int want = (stagger == 0 || t < stagger || k <= adsb::kTaperPasses) ? k : adsb::kTaperPasses;
            if (d != (uint64_t)adsb::owned_runs(kt) || kt != want) {
                printf("tile %u (stagger %u, k %d): width %llu, passes %d\n", t, stagger, k, (unsigned long long)d, kt);
                return 1;
            }
        }
    }
    // the host's choice: no tail of small tiles unless one is forced, and never for tiles that are short already
    if (adsb::choose_big_tiles(1 << 20, 7, 256, 0) != 0 || adsb::choose_big_tiles(134216525, 7, 256, 0) != 0 ||
        adsb::choose_big_tiles(134216525, 4, 256, 17) != 0 || adsb::choose_big_tiles(134216525, 7, 256, 17) != 17) {
        printf("choose_big_tiles: default / short tiles / forced\n");
        return 1;
    }
    // LDS of the largest tile fits four workgroups per CU
    if (adsb::lds_bytes(6) * 4 > 160 * 1024) {
        printf("lds_bytes(6) = %zu: four workgroups no longer fit a CU\n", adsb::lds_bytes(6));
        return 1;
    }
    printf("ok\n");
    return 0;
}
