// init_probe.hip -- where the GPU runtime's start goes: wall time of the first HIP calls of a process, one by one, in the order
// adsb_create makes them (the C host program's "runtime init" line is their sum).
//   hipcc --offload-arch=gfx950 -O2 tools/init_probe.hip -o tools/bin/init_probe && tools/bin/init_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <unistd.h>
using clk = std::chrono::steady_clock;
static clk::time_point t_prev;
static void lap(const char *what)
{
    const auto t = clk::now();
    printf("%8.2f ms  %s\n", std::chrono::duration<double, std::milli>(t - t_prev).count(), what);
    t_prev = clk::now();
}
__global__ void touch(unsigned *p) { p[threadIdx.x] = threadIdx.x; }

int main()
{
    const auto t0 = t_prev = clk::now();
    int n = 0;
    hipGetDeviceCount(&n);
    lap("hipGetDeviceCount (runtime + HSA start)");
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    lap("hipGetDeviceProperties");
    hipSetDevice(0);
    lap("hipSetDevice");
    hipStream_t s[5];
    hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
    lap("hipStreamCreateWithFlags #1");
    for (int i = 1; i < 5; i++)
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    lap("hipStreamCreateWithFlags #2..5");
    hipEvent_t ev[8];
    for (auto &e : ev)
        hipEventCreate(&e);
    lap("hipEventCreate x 8");
    unsigned *d_small = nullptr, *d_big = nullptr, *h = nullptr, *h_big = nullptr;
    hipMalloc(&d_small, 4096);
    lap("hipMalloc 4 KiB (first)");
    hipMemset(d_small, 0, 4096);
    lap("hipMemset (first blit: queue + blit kernel)");
    hipHostMalloc(&h, 4096, hipHostMallocCoherent);
    lap("hipHostMalloc 4 KiB coherent (first)");
    hipMalloc(&d_big, 64u << 20);
    lap("hipMalloc 64 MiB");
    hipHostMalloc(&h_big, 64u << 20, hipHostMallocDefault);
    lap("hipHostMalloc 64 MiB");
    hipMemcpy(d_small, h, 4096, hipMemcpyHostToDevice);
    lap("hipMemcpy H2D 4 KiB (first)");
    touch<<<1, 64, 0, s[0]>>>(d_small);
    hipStreamSynchronize(s[0]);
    lap("first launch of an own kernel on stream 1 + sync (code object load, queue)");
    touch<<<1, 64, 0, s[1]>>>(d_small);
    hipStreamSynchronize(s[1]);
    lap("first launch on stream 2 + sync (another queue)");
    touch<<<1, 64, 0, s[0]>>>(d_small);
    hipStreamSynchronize(s[0]);
    lap("second launch on stream 1 + sync");
    printf("%8.2f ms  total\n", std::chrono::duration<double, std::milli>(clk::now() - t0).count());
    fflush(stdout);
    _exit(0);
}
