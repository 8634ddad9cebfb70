// kbench.hip -- stand-alone timing of the shipped adsb::scan_kernel on device-generated noise (kernel time only: no host
// consumer).  Not part of the library.   kbench [Mi samples] [iterations] [passes] [sigma]; KB_HAND=1: hand-off stream on.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -I adsbdec_amd/csrc tools/kbench.hip -o tools/bin/kbench
// (The ablation builds of rounds 2-3 -- loads only, Stage A only, cache-resident input, per-tile clocks, the pipelined kernel --
// were #if branches of scan_kernel.hip; they left the tree with round 4 and live in its history: DESIGN_HISTORY.md.)
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
    a.passes = argc > 3 && atoi(argv[3]) > 0 ? atoi(argv[3]) : adsb::choose_passes(a.g_end - a.g_begin, 256);
    a.big_tiles = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) { CK(hipMemset(counters, 0, adsb::kDevCounterWords * 4)); CK(adsb::launch_scan(a, false, 0)); }
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < iters; i++) {
        
        CK(hipEventRecord(e0, 0)); CK(adsb::launch_scan(a, false, 0)); CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    // the clock governor needs ~20 ms of load to settle: with 200 iterations or
    // more, only the second half counts
    if (t.size() >= 200)
        t.erase(t.begin(), t.begin() + t.size() / 2);
    std::sort(t.begin(), t.end());
    CK(hipDeviceSynchronize());
    const uint32_t hc[2] = {report[0], report[1]}; // the last tile's report (the device counters are zero again)
    double med = t[t.size() / 2];
    printf("passes=%d tile=%d lds=%zu | median %.4f ms min %.4f | %.1f GB/s alg | %.1f Gsamples/s | cands=%u\n", a.passes,
           adsb::tile_offsets(a.passes), adsb::lds_bytes(a.passes), med, t[0], 2.0 * n / med / 1e6, n / med / 1e6, hc[0]);
    return 0;
}
