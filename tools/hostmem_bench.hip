// hostmem_bench.hip -- how expensive are host reads of device-written pinned memory?
// (tuning aid for the hand-off stream; not part of the library)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <emmintrin.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint32_t *p, size_t n, uint32_t v)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = v + (uint32_t)i;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 128 * 1024; // dwords = 512 KiB
    const unsigned flags[3] = {hipHostMallocCoherent, hipHostMallocNonCoherent, hipHostMallocDefault};
    const char *names[3] = {"coherent", "noncoherent", "default"};
    for (int f = 0; f < 3; f++) {
        uint32_t *h;
        CK(hipHostMalloc(&h, n * 4, flags[f]));
        memset(h, 0, n * 4);
        std::vector<uint32_t> dst(n);
        for (int rep = 0; rep < 3; rep++) {
            fill<<<(n + 255) / 256, 256>>>(h, n, rep * 7919u);
            CK(hipDeviceSynchronize());
            double t0 = now();
            uint64_t acc = 0;
            const volatile uint32_t *v = h;
            for (size_t i = 0; i < n; i++) acc += v[i];
            double t1 = now();
            for (size_t i = 0; i < n; i++) acc += v[i];
            double t2 = now();
            fill<<<(n + 255) / 256, 256>>>(h, n, rep * 104729u + 1);
            CK(hipDeviceSynchronize());
            double t3 = now();
            __m128i a = _mm_setzero_si128();
            for (size_t i = 0; i < n; i += 4) a = _mm_add_epi32(a, _mm_load_si128((const __m128i *)(h + i)));
            double t4 = now();
            fill<<<(n + 255) / 256, 256>>>(h, n, rep * 31u + 2);
            CK(hipDeviceSynchronize());
            double t5 = now();
            memcpy(dst.data(), h, n * 4);
            double t6 = now();
            acc += _mm_cvtsi128_si32(a) + dst[5];
            printf("%-12s rep %d: 4B loads first %.1f us, again %.1f us | 16B loads first %.1f us | memcpy first %.1f us  (512 KiB) [%llu]\n",
                   names[f], rep, t1 - t0, t2 - t1, t4 - t3, t6 - t5, (unsigned long long)acc);
        }
        CK(hipHostFree(h));
    }
    return 0;
}
