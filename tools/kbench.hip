// kbench.hip -- stand-alone timing of adsb::scan_kernel variants (tile shape via
// -DADSB_THREADS / -DADSB_PASSES, ablations via -DADSB_ABLATE). Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I adsbdec_amd/csrc tools/kbench.hip -o kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "scan_kernel.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline uint32_t mix(uint32_t h)
{
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}

// Gaussian noise of standard deviation sigma around 2048 (Box-Muller on hashed uniforms)
__global__ void fill_noise(uint16_t *x, size_t n, uint32_t seed, float sigma)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t h1 = mix((uint32_t)i * 2654435761u ^ seed), h2 = mix(h1 ^ 0x9E3779B9u);
        float u1 = ((h1 >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
        float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
        int v = (int)rintf(2048.0f + sigma * g);
        v = v < 0 ? 0 : (v > 4095 ? 4095 : v);
        x[i] = (uint16_t)v;
    }
}

int main(int argc, char **argv)
{
    size_t n = (size_t)(argc > 1 ? atoll(argv[1]) : 256) << 20;
    int iters = argc > 2 ? atoi(argv[2]) : 20;
    n -= n % 28;
    uint16_t *x; uint32_t *counters, *cands;
    CK(hipMalloc(&x, n * 2));
    CK(hipMalloc(&counters, adsb::kDevCounterWords * 4));
    CK(hipMalloc(&cands, (1u << 20) * 24));
    fill_noise<<<4096, 256>>>(x, n, 12345, argc > 4 ? (float)atof(argv[4]) : 8.0f);
    CK(hipDeviceSynchronize());
    adsb::ScanArgs a{};
    a.x = (const uint32_t *)x; a.pbuf0 = 0; a.p_lo = 0; a.p_hi = n / 2;
    a.g_begin = 0; a.g_end = (n / 2 - 1195) / 28 * 28; a.df18 = 0;
    uint32_t *report; CK(hipHostMalloc(&report, 32, hipHostMallocCoherent));
    a.report = report; a.counters = counters; a.cands = cands; a.cand_cap = 1u << 20; a.tries = nullptr; a.try_cap = 0;
    std::vector<uint32_t> synd(adsb::kSyndWords); adsb::make_syndrome_table(synd.data());
    uint32_t *dsynd; CK(hipMalloc(&dsynd, synd.size() * 4)); CK(hipMemcpy(dsynd, synd.data(), synd.size() * 4, hipMemcpyHostToDevice));
    a.synd = dsynd; a.queue_cap = adsb::kQueueCap; a.all_candidates = 0; a.clist_cap = adsb::kClistCap; a.fix_tab = nullptr; a.fix_mul = 0; a.hand = nullptr; a.hand_cap = 0; a.gen = 0;
    if (getenv("KB_HAND") && atoi(getenv("KB_HAND"))) { // hand-off stream into pinned host memory, nobody reading it
        uint32_t *hand; const size_t gran = 4u << 20;
        CK(hipHostMalloc(&hand, gran * 16, hipHostMallocCoherent));
        a.hand = hand; a.hand_cap = (uint32_t)gran; a.gen = 12345;
    }
    a.pipe = getenv("KB_PIPE") && atoi(getenv("KB_PIPE")) ? 1 : 0; // the pipelined kernel (persistent five-wave workgroups)
    a.passes = argc > 3 && atoi(argv[3]) > 0 ? atoi(argv[3]) : adsb::choose_passes(a.g_end - a.g_begin, 256, a.pipe != 0);
    a.stagger = a.pipe ? 0u : adsb::choose_stagger(a.g_end - a.g_begin, 256, a.passes); // ADSB_STAGGER=0 turns it off
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) { CK(hipMemset(counters, 0, adsb::kDevCounterWords * 4)); CK(adsb::launch_scan(a, false, 0)); }
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < iters; i++) {
        
        CK(hipEventRecord(e0, 0)); CK(adsb::launch_scan(a, false, 0)); CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
#if ADSB_TILE_CLOCK
    {   // one more launch with the per-tile clock buffer; dump "tile begin_us end_us xcc cu" to stdout
        const uint32_t nt = adsb::tile_count(a.g_end - a.g_begin, a.stagger, a.passes);
        uint32_t *dclk; CK(hipMalloc(&dclk, (size_t)nt * 16)); CK(hipMemset(dclk, 0, (size_t)nt * 16));
        a.tile_clock = dclk;
         CK(adsb::launch_scan(a, false, 0)); CK(hipDeviceSynchronize());
        std::vector<uint32_t> h((size_t)nt * 4); CK(hipMemcpy(h.data(), dclk, h.size() * 4, hipMemcpyDeviceToHost));
        uint32_t t0c = ~0u; for (uint32_t i = 0; i < nt; i++) t0c = std::min(t0c, h[4 * i]);
#if ADSB_TILE_CLOCK == 2
        {   // the clock the chip holds while this kernel runs: shader cycles / device real time (100 MHz) per tile
            std::vector<double> ghz;
            for (uint32_t i = 0; i < nt; i++) {
                const double us = (h[4 * i + 1] - h[4 * i]) * 0.01;
                if (us > 5.0)
                    ghz.push_back(h[4 * i + 2] / us * 1e-3);
            }
            std::sort(ghz.begin(), ghz.end());
            if (!ghz.empty())
                printf("in-kernel shader clock over %zu tiles: median %.3f GHz (p5 %.3f, p95 %.3f)\n", ghz.size(), ghz[ghz.size() / 2],
                       ghz[ghz.size() / 20], ghz[ghz.size() - 1 - ghz.size() / 20]);
        }
#endif
        FILE *f = fopen(getenv("ADSB_CLOCK_OUT") ? getenv("ADSB_CLOCK_OUT") : "tile_clock.txt", "w");
        for (uint32_t i = 0; i < nt; i++)
            fprintf(f, "%u %.2f %.2f %u %u %u\n", i, (h[4 * i] - t0c) * 0.01, (h[4 * i + 1] - t0c) * 0.01, h[4 * i + 3] & 15u,
                    (h[4 * i + 2] >> 8) & 15u, (h[4 * i + 2] >> 13) & 7u); // xcc, cu_id, sh/se bits
        fclose(f);
        a.tile_clock = nullptr;
    }
#endif
    // the clock governor needs ~20 ms of load to settle: with 200 iterations or
    // more, only the second half counts
    if (t.size() >= 200)
        t.erase(t.begin(), t.begin() + t.size() / 2);
    std::sort(t.begin(), t.end());
    CK(hipDeviceSynchronize());
    const uint32_t hc[2] = {report[0], report[1]}; // the last tile's report (the device counters are zero again)
    double med = t[t.size() / 2];
    printf("pipe=%d passes=%d stagger=%u ablate=%d minwaves=%d tile=%d lds=%zu | median %.4f ms min %.4f | %.1f GB/s alg | %.1f Gsamples/s | cands=%u\n",
           a.pipe, a.passes, a.stagger, a.pipe ? ADSB_PIPE_ABLATE : ADSB_ABLATE, a.pipe ? ADSB_PIPE_WAVES : ADSB_MIN_WAVES, adsb::tile_offsets(a.passes),
           a.pipe ? adsb::lds_bytes_pipe(a.passes) : adsb::lds_bytes(a.passes), med, t[0],
           2.0 * n / med / 1e6, n / med / 1e6, hc[0]);
    return 0;
}
