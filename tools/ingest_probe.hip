// ingest_probe.hip -- what does it cost to get a FILE of samples to the device?  (tuning aid for the C host program's ring and
// the multi-GPU driver's file source; not part of the library)
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/ingest_probe tools/ingest_probe.hip -lpthread
//   tools/bin/ingest_probe /dev/shm/capture.u16        (a file of >= 512 MiB; created if absent)
//
// Measures, per 32 MiB window: hipHostRegister / hipHostUnregister of anonymous memory (4 KiB pages, transparent huge pages)
// with 1, 2 and 4 threads at once; the same on windows of an mmap of the file (MAP_SHARED, MAP_PRIVATE, with and without
// MAP_POPULATE) -- does the runtime accept page-cache pages at all? -- and the host-to-device copy rate from each kind of
// memory; pread() into page-locked memory with 1, 2 and 4 threads; plain memcpy.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t W = 32u << 20;

static void parallel(int nt, const std::function<void(int)> &fn)
{
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back(fn, t);
    for (auto &x : th)
        x.join();
}

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/dev/shm/ingest_probe.u16";
    const size_t total = 512u << 20;
    int fd = open(path, O_RDWR | O_CREAT, 0644);
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0)
        return 1;
    if ((size_t)sb.st_size < total) {
        std::vector<char> blk(W);
        for (size_t i = 0; i < W; i++)
            blk[i] = (char)(i * 2654435761u >> 13);
        for (size_t at = 0; at < total; at += W)
            if (pwrite(fd, blk.data(), W, (off_t)at) != (ssize_t)W)
                return 1;
    }
    hipStream_t st;
    if (hipSetDevice(0) != hipSuccess || hipStreamCreate(&st) != hipSuccess)
        return 1;
    void *dev = nullptr;
    hipMalloc(&dev, total);
    const int nwin = (int)(total / W);

    auto copy_rate = [&](const char *what, char *base, bool per_window_sync) {
        const double t0 = now_ms();
        for (int k = 0; k < nwin; k++) {
            hipMemcpyAsync((char *)dev + (size_t)k * W, base + (size_t)k * W, W, hipMemcpyHostToDevice, st);
            if (per_window_sync)
                hipStreamSynchronize(st);
        }
        hipStreamSynchronize(st);
        const double dt = now_ms() - t0;
        printf("    H2D from %-44s %7.2f ms = %5.1f GB/s\n", what, dt, total / dt / 1e6);
    };

    // ---- anonymous memory
    for (int thp = 0; thp < 2; thp++) {
        char *buf = nullptr;
        if (posix_memalign((void **)&buf, 2u << 20, total) != 0)
            return 1;
        madvise(buf, total, thp ? MADV_HUGEPAGE : MADV_NOHUGEPAGE);
        double t0 = now_ms();
        memset(buf, 1, total);
        printf("anonymous memory, %s: first touch of 512 MiB %.1f ms\n", thp ? "transparent huge pages" : "4 KiB pages", now_ms() - t0);
        for (int nt : {1, 2, 4}) {
            std::atomic<int> fails{0};
            t0 = now_ms();
            parallel(nt, [&](int t) {
                hipSetDevice(0);
                for (int k = t; k < nwin; k += nt)
                    if (hipHostRegister(buf + (size_t)k * W, W, hipHostRegisterDefault) != hipSuccess)
                        fails++;
            });
            const double t_reg = now_ms() - t0;
            if (nt == 1 && !fails.load())
                copy_rate("registered anonymous memory", buf, false);
            t0 = now_ms();
            parallel(nt, [&](int t) {
                hipSetDevice(0);
                for (int k = t; k < nwin; k += nt)
                    hipHostUnregister(buf + (size_t)k * W);
            });
            const double t_unreg = now_ms() - t0;
            printf("  %d thread(s): hipHostRegister of %d x 32 MiB %7.2f ms = %5.1f GB/s (%d failed); unregister %7.2f ms\n", nt, nwin, t_reg,
                   total / t_reg / 1e6, fails.load(), t_unreg);
        }
        free(buf);
    }
    // ---- the file, mapped
    struct { const char *name; int flags; } maps[] = {{"MAP_SHARED", MAP_SHARED}, {"MAP_SHARED | MAP_POPULATE", MAP_SHARED | MAP_POPULATE},
                                                       {"MAP_PRIVATE", MAP_PRIVATE}, {"MAP_PRIVATE | MAP_POPULATE", MAP_PRIVATE | MAP_POPULATE}};
    for (auto &mp : maps)
        for (int prot_w = 0; prot_w < 2; prot_w++) {
            double t0 = now_ms();
            char *m = (char *)mmap(nullptr, total, PROT_READ | (prot_w ? PROT_WRITE : 0), mp.flags, fd, 0);
            if (m == MAP_FAILED) {
                printf("mmap %s failed\n", mp.name);
                continue;
            }
            madvise(m, total, MADV_WILLNEED);
            madvise(m, total, MADV_HUGEPAGE);
            const double t_map = now_ms() - t0;
            int fails = 0;
            hipError_t last = hipSuccess;
            t0 = now_ms();
            for (int k = 0; k < nwin; k++) {
                const hipError_t e = hipHostRegister(m + (size_t)k * W, W, prot_w ? hipHostRegisterDefault : hipHostRegisterReadOnly);
                if (e != hipSuccess) {
                    fails++;
                    last = e;
                    (void)hipGetLastError();
                }
            }
            const double t_reg = now_ms() - t0;
            printf("mmap(file, %s, %s): map %.2f ms; hipHostRegister of %d windows %7.2f ms = %5.1f GB/s, %d failed%s%s\n", mp.name,
                   prot_w ? "PROT_READ|WRITE" : "PROT_READ", t_map, nwin, t_reg, total / t_reg / 1e6, fails, fails ? ": " : "",
                   fails ? hipGetErrorString(last) : "");
            if (!fails) {
                copy_rate("registered windows of the mapping", m, false);
                for (int k = 0; k < nwin; k++)
                    hipHostUnregister(m + (size_t)k * W);
            } else {
                for (int k = 0; k < nwin; k++)
                    (void)hipHostUnregister(m + (size_t)k * W);
                (void)hipGetLastError();
            }
            copy_rate("the mapping, NOT registered (pageable copy)", m, false);
            munmap(m, total);
        }
    // ---- does registering windows of ONE mapping scale with threads?  (a fresh mapping per run: the runtime keeps no memory of it)
    for (int nt : {1, 2, 3, 4, 8}) {
        char *m = (char *)mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED)
            break;
        madvise(m, total, MADV_WILLNEED);
        std::atomic<int> fails{0};
        double t0 = now_ms();
        parallel(nt, [&](int t) {
            hipSetDevice(0);
            for (int k = t; k < nwin; k += nt)
                if (hipHostRegister(m + (size_t)k * W, W, hipHostRegisterReadOnly) != hipSuccess)
                    fails++;
        });
        const double t_reg = now_ms() - t0;
        t0 = now_ms();
        parallel(nt, [&](int t) {
            hipSetDevice(0);
            for (int k = t; k < nwin; k += nt)
                (void)hipHostUnregister(m + (size_t)k * W);
        });
        const double t_unreg = now_ms() - t0;
        printf("mmap(file, MAP_SHARED, PROT_READ), %d thread(s) registering windows side by side: %7.2f ms = %5.1f GB/s (%d failed); unregister %7.2f ms\n", nt,
               t_reg, total / t_reg / 1e6, fails.load(), t_unreg);
        munmap(m, total);
    }
    // ... and the whole pipeline: R threads register windows ahead, the main thread copies window k once it is registered
    for (int nt : {1, 2, 3, 4}) {
        char *m = (char *)mmap(nullptr, total, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED)
            break;
        madvise(m, total, MADV_WILLNEED);
        std::vector<std::atomic<int>> ready(nwin);
        for (auto &r : ready)
            r.store(0);
        const double t0 = now_ms();
        std::vector<std::thread> rg;
        for (int t = 0; t < nt; t++)
            rg.emplace_back([&, t] {
                hipSetDevice(0);
                for (int k = t; k < nwin; k += nt)
                    ready[k].store(hipHostRegister(m + (size_t)k * W, W, hipHostRegisterReadOnly) == hipSuccess ? 1 : -1, std::memory_order_release);
            });
        int bad = 0;
        for (int k = 0; k < nwin; k++) {
            int r;
            while ((r = ready[k].load(std::memory_order_acquire)) == 0)
                ;
            bad += r < 0;
            hipMemcpyAsync((char *)dev + (size_t)k * W, m + (size_t)k * W, W, hipMemcpyHostToDevice, st);
        }
        hipStreamSynchronize(st);
        const double dt = now_ms() - t0;
        for (auto &x : rg)
            x.join();
        for (int k = 0; k < nwin; k++)
            (void)hipHostUnregister(m + (size_t)k * W);
        printf("file -> device, zero copy: %d thread(s) register windows of the mapping ahead of the copies: %7.2f ms = %5.1f GB/s (%d windows pageable)\n", nt, dt,
               total / dt / 1e6, bad);
        munmap(m, total);
    }
    // ---- pread into page-locked memory, memcpy
    char *pin = nullptr;
    hipHostMalloc((void **)&pin, total, hipHostMallocDefault);
    copy_rate("hipHostMalloc memory", pin, false);
    for (int nt : {1, 2, 4, 8}) {
        const double t0 = now_ms();
        parallel(nt, [&](int t) {
            for (int k = t; k < nwin; k += nt) {
                size_t got = 0;
                while (got < W) {
                    const ssize_t n = pread(fd, pin + (size_t)k * W + got, W - got, (off_t)((size_t)k * W + got));
                    if (n <= 0)
                        break;
                    got += (size_t)n;
                }
            }
        });
        const double dt = now_ms() - t0;
        printf("pread of the file into page-locked memory, %d thread(s): %7.2f ms = %5.1f GB/s\n", nt, dt, total / dt / 1e6);
    }
    {
        char *src = (char *)malloc(total);
        memset(src, 3, total);
        for (int nt : {1, 2, 4}) {
            const double t0 = now_ms();
            parallel(nt, [&](int t) {
                for (int k = t; k < nwin; k += nt)
                    memcpy(pin + (size_t)k * W, src + (size_t)k * W, W);
            });
            const double dt = now_ms() - t0;
            printf("memcpy into page-locked memory, %d thread(s): %7.2f ms = %5.1f GB/s\n", nt, dt, total / dt / 1e6);
        }
        free(src);
    }
    // one reader + copy pipelined: window k+1 is pread while window k is copied to the device
    for (int nt : {1, 2, 3, 4, 6}) {
        std::vector<std::atomic<int>> ready(nwin);
        for (auto &r : ready)
            r.store(0);
        const double t0 = now_ms();
        std::vector<std::thread> rd;
        for (int t = 0; t < nt; t++)
            rd.emplace_back([&, t] {
                for (int k = t; k < nwin; k += nt) {
                    size_t got = 0;
                    while (got < W) {
                        const ssize_t n = pread(fd, pin + (size_t)k * W + got, W - got, (off_t)((size_t)k * W + got));
                        if (n <= 0)
                            break;
                        got += (size_t)n;
                    }
                    ready[k].store(1, std::memory_order_release);
                }
            });
        for (int k = 0; k < nwin; k++) {
            while (!ready[k].load(std::memory_order_acquire))
                ;
            hipMemcpyAsync((char *)dev + (size_t)k * W, pin + (size_t)k * W, W, hipMemcpyHostToDevice, st);
        }
        hipStreamSynchronize(st);
        for (auto &x : rd)
            x.join();
        const double dt = now_ms() - t0;
        printf("file -> device, %d pread thread(s) pipelined with the copies: %7.2f ms = %5.1f GB/s\n", nt, dt, total / dt / 1e6);
    }
    hipHostFree(pin);
    hipFree(dev);
    close(fd);
    if (argc <= 1)
        unlink(path);
    return 0;
}
