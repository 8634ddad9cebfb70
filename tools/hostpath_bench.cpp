// hostpath_bench.cpp -- CPU-only timing of the host side of the streaming hand-off
// (stream validation + in-place resolve) on a synthetic stream shaped like the bench
// capture: 3941 tiles, ~15k records.  Tuning aid, not part of the library.
//   g++ -O2 -std=c++17 -I adsbdec_amd/csrc tools/hostpath_bench.cpp -o tools/bin/hostpath_bench
#include <chrono>
#include <cstdio>
#include <emmintrin.h>
#include <random>
#include <vector>
#include "resolver.hpp"
static inline uint32_t rotl(uint32_t v, int s) { return v << s | v >> (32 - s); }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const uint32_t ntiles = 3941, per = 34048, gen = 0x1234567u;
    std::mt19937 rng(1);
    std::vector<uint32_t> hand_v;
    uint64_t nrec = 0;
    for (uint32_t t = 0; t < ntiles; t++) {
        std::vector<uint32_t> gs;
        for (uint32_t g = rng() % 9000; g < per; g += 7000 + rng() % 6000) {
            gs.push_back(g);
            if (rng() % 8 == 0 && g + 1 < per) gs.push_back(g + 1);
        }
        const uint32_t n = gs.size();
        uint32_t acc[4] = {0, 0, 0, 0};
        const size_t mpos = hand_v.size();
        hand_v.insert(hand_v.end(), 4, 0u);
        for (uint32_t g : gs) {
            uint32_t r[8] = {t * per + g, 1000 + g % 77, (17u << 3) | (uint32_t)(rng() << 8), (uint32_t)rng(), (uint32_t)rng(),
                             ((uint32_t)rng() & 0xFFFFu) | (14u << 16), 0, 0};
            for (int k = 0; k < 8; k++) acc[k & 3] ^= r[k];
            hand_v.insert(hand_v.end(), r, r + 8);
            nrec++;
        }
        uint32_t *m = hand_v.data() + mpos;
        m[0] = t, m[1] = n;
        m[2] = acc[0] ^ rotl(acc[2], 16) ^ gen ^ t ^ rotl(n, 11);
        m[3] = acc[1] ^ rotl(acc[3], 16) ^ ~gen ^ rotl(t, 7) ^ n;
    }
    // 16-byte aligned copy
    std::vector<__m128i> store(hand_v.size() / 4 + 1);
    uint32_t *hand = reinterpret_cast<uint32_t *>(store.data());
    std::copy(hand_v.begin(), hand_v.end(), hand);
    printf("%u tiles, %llu records, %zu KiB stream\n", ntiles, (unsigned long long)nrec, hand_v.size() * 4 / 1024);
    adsb::Resolver res;
    std::vector<uint32_t> order, t_start(ntiles), t_count(ntiles);
    std::vector<adsb_frame> out(nrec);
    for (int rep = 0; rep < 8; rep++) {
        res.reset();
        double tr = 0;
        const double t0 = now();
        uint32_t pos = 0, frontier = 0, delivered = 0;
        std::fill(t_count.begin(), t_count.end(), ~0u);
        while (frontier < ntiles) {
            const __m128i *gp = reinterpret_cast<const __m128i *>(hand) + pos;
            const __m128i mk = _mm_load_si128(gp);
            const uint32_t tile = (uint32_t)_mm_cvtsi128_si32(mk), nf = (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(mk, 4));
            const uint32_t n = nf & 0xFFFFu;
            if (tile >= ntiles || n > 4096) return 1;
            __m128i acc = _mm_setzero_si128();
            for (uint32_t k = 0; k < 2 * n; k++) acc = _mm_xor_si128(acc, _mm_load_si128(gp + 1 + k));
            alignas(16) uint32_t a[4], mw[4];
            _mm_store_si128((__m128i *)a, acc);
            _mm_store_si128((__m128i *)mw, mk);
            if (mw[2] != (a[0] ^ rotl(a[2], 16) ^ gen ^ tile ^ rotl(nf, 11)) || mw[3] != (a[1] ^ rotl(a[3], 16) ^ ~gen ^ rotl(tile, 7) ^ nf))
                return 2;
            t_start[tile] = pos + 1;
            t_count[tile] = n;
            pos += 1 + 2 * n;
            while (frontier < ntiles && t_count[frontier] != ~0u) frontier++;
            if (frontier - delivered >= 512 || frontier == ntiles) {
                const double ta = now();
                res.advance_tiles(hand, t_start.data(), t_count.data(), delivered, frontier, 0, 0, (uint64_t)ntiles * per + 100000,
                                  (uint64_t)frontier * per);
                delivered = frontier;
                tr += now() - ta;
            }
        }
        const double t1 = now();
        const size_t nf = res.drain(out.data(), out.size());
        const double t2 = now();
        printf("rep %d: parse+validate %.1f us, resolve %.1f us, total %.1f us, drain %.1f us, %zu frames\n", rep, t1 - t0 - tr, tr,
               t1 - t0, t2 - t1, nf);
    }
    return 0;
}
