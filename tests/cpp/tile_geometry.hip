// tile_geometry.hip -- host-side check of the tile geometry the kernel and the host must
// agree on (scan_kernel.h: tile_passes / tile_first_run / tile_count), staggered or not.
// Built with hipcc and run on the CPU by tests/test_host_logic.py.
#include <cstdio>
#include <random>

#include "../../adsbdec_amd/csrc/scan_kernel.h"

int main()
{
    std::mt19937_64 rng(7);
    for (int it = 0; it < 20000; it++) {
        const int k = 2 + (int)(rng() % 5);
        const uint32_t stagger = (k >= 5 && (rng() & 1)) ? 4u * (uint32_t)(rng() % 300) : 0u;
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
            if (d != (uint64_t)adsb::owned_runs(kt) || kt < 2 || kt > k) {
                printf("tile %u (stagger %u, k %d): width %llu, passes %d\n", t, stagger, k, (unsigned long long)d, kt);
                return 1;
            }
        }
    }
    // LDS of the largest tile fits four workgroups per CU
    if (adsb::lds_bytes(6) * 4 > 160 * 1024) {
        printf("lds_bytes(6) = %zu: four workgroups no longer fit a CU\n", adsb::lds_bytes(6));
        return 1;
    }
    printf("ok\n");
    return 0;
}
