// numa.cpp -- WHERE a host-resident capture lives, on a two-socket node with eight GPUs.
//
// The reference's input buffer is `iqbuff = malloc(...)` (air.c:230): one thread, one buffer, whatever node the allocator
// picks.  Here a capture in host memory is pulled by up to eight devices at once, each over its own PCIe link, and half of
// those links hang off the other socket: a slice that lives on the wrong socket crosses the socket fabric on its way to
// the device, and four of eight links share that fabric.  So:
//   adsb_host_alloc_on(bytes, device)       page-locked memory on the NUMA node of `device`
//   adsb_multi_host_alloc(m, total_samples) ONE capture, laid out shard by shard (adsb_multi_plan) on the node of the device
//                                           that will pull that shard
//   adsb_host_placement(p, bytes, ...)      where do these pages live?  (move_pages query)
// Host-only code, no HIP: what needs the runtime (page-locking, "which node is device k on?") comes through the public
// C-ABI (adsb_host_register, adsb_device_numa_node), so that tests/cpp/multi_tsan.cpp runs this file against its fake
// device backend with a made-up two-node machine.  No libnuma in the image: mbind / move_pages are raw syscalls.
// Everything is best effort: on a one-node machine, or where the policy calls are refused, the memory is simply
// page-locked where the kernel put it, and the placement query says so.
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "../../include/adsbdec_amd_diag.h"

namespace {

constexpr size_t kHuge = 2u << 20;
constexpr int kMpolBind = 2, kMpolPreferred = 1; // <linux/mempolicy.h>
constexpr unsigned kMpolMfMove = 1u << 1;

struct Block {
    size_t bytes = 0;      // of the mapping
    bool registered = false;
};
std::mutex g_mu;
std::unordered_map<void *, Block> g_blocks; // mappings handed out by this file (adsb_host_free looks here first)

long sys_mbind(void *addr, unsigned long len, int mode, const unsigned long *mask, unsigned long maxnode, unsigned flags)
{
    return syscall(SYS_mbind, addr, len, mode, mask, maxnode, flags);
}
long sys_move_pages(int pid, unsigned long count, void **pages, const int *nodes, int *status, int flags)
{
    return syscall(SYS_move_pages, pid, count, pages, nodes, status, flags);
}

// prefer `node` for [p, p + bytes) (page-aligned inside): MPOL_BIND where allowed, else MPOL_PREFERRED; false: neither
bool bind_range(char *p, size_t bytes, int node)
{
    if (node < 0 || node >= 1024 || bytes == 0)
        return false;
    unsigned long mask[16] = {0};
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    if (sys_mbind(p, bytes, kMpolBind, mask, 1024, 0) == 0)
        return true;
    return sys_mbind(p, bytes, kMpolPreferred, mask, 1024, 0) == 0;
}

void touch(char *p, size_t bytes)
{
    const long page = sysconf(_SC_PAGESIZE);
    for (size_t at = 0; at < bytes; at += (size_t)page)
        p[at] = 0;
}

char *map_block(size_t bytes, size_t *mapped)
{
    const size_t len = (std::max<size_t>(bytes, 1) + kHuge - 1) / kHuge * kHuge;
    // over-map by one huge page to get a 2 MiB-aligned start (transparent huge pages: 512 x fewer pages to fault and to pin)
    char *raw = static_cast<char *>(mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (raw == MAP_FAILED)
        return nullptr;
    char *p = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) / kHuge * kHuge);
    if (p > raw)
        munmap(raw, (size_t)(p - raw));
    if (p + len < raw + len + kHuge)
        munmap(p + len, (size_t)(raw + len + kHuge - (p + len)));
    madvise(p, len, MADV_HUGEPAGE);
    *mapped = len;
    return p;
}

void *finish_block(char *p, size_t len)
{
    Block b;
    b.bytes = len;
    b.registered = adsb_host_register(p, len) == 0;
    if (!b.registered) { // (no runtime, or it refuses: page-locking is the point of this memory)
        munmap(p, len);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_blocks[p] = b;
    return p;
}

} // namespace

extern "C" {

void *adsb_host_alloc_on(size_t bytes, int device)
{
    size_t len = 0;
    char *p = map_block(bytes, &len);
    if (!p)
        return nullptr;
    (void)bind_range(p, len, adsb_device_numa_node(device));
    touch(p, len); // first touch under the policy: the pages exist, on that node, before they are pinned
    return finish_block(p, len);
}

uint16_t *adsb_host_alloc_sharded(uint64_t total_samples, int n_shards, const uint64_t *first_sample, const uint64_t *n_samples,
                                  const int *devices)
{
    if (n_shards < 0 || (n_shards && (!first_sample || !n_samples || !devices)))
        return nullptr;
    size_t len = 0;
    char *p = map_block((size_t)total_samples * sizeof(uint16_t), &len);
    if (!p)
        return nullptr;
    // Shard i's samples [first_sample[i], first_sample[i] + n_samples[i]) overlap their neighbours' by the halo (2 408
    // samples): the boundary between two nodes is put where shard i+1 starts, rounded to a huge page -- the halo of one of
    // the two crosses the fabric, 5 KB in a gigabyte.
    size_t at = 0;
    for (int i = 0; i < n_shards; i++) {
        size_t end = i + 1 < n_shards ? (size_t)first_sample[i + 1] * sizeof(uint16_t) / kHuge * kHuge : len;
        end = std::min(std::max(end, at), len);
        if (end > at)
            (void)bind_range(p + at, end - at, adsb_device_numa_node(devices[i]));
        at = end;
    }
    touch(p, len);
    return static_cast<uint16_t *>(finish_block(p, len));
}

// (adsb_host_free's first look: 1 = the block was one of this file's and is gone)
int adsb_host_release_mapped(void *p)
{
    Block b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_blocks.find(p);
        if (it == g_blocks.end())
            return 0;
        b = it->second;
        g_blocks.erase(it);
    }
    if (b.registered)
        (void)adsb_host_unregister(p);
    munmap(p, b.bytes);
    return 1;
}

int adsb_host_placement(const void *p, size_t bytes, int want_node, int *major_node, double *fraction_on_want)
{
    if (major_node)
        *major_node = -1;
    if (fraction_on_want)
        *fraction_on_want = 0;
    if (!p || bytes == 0)
        return -1;
    // up to 256 pages, evenly spread
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p) / page * page, hi = reinterpret_cast<uintptr_t>(p) + bytes;
    const size_t npages = (hi - lo + page - 1) / page, n = std::min<size_t>(npages, 256);
    std::vector<void *> pages(n);
    std::vector<int> status(n, -1);
    for (size_t k = 0; k < n; k++)
        pages[k] = reinterpret_cast<void *>(lo + (npages * k / n) * page);
    if (sys_move_pages(0, n, pages.data(), nullptr, status.data(), 0) != 0)
        return -1; // (no permission, no NUMA support: unknown)
    int count[64] = {0}, known = 0, on_want = 0;
    for (size_t k = 0; k < n; k++)
        if (status[k] >= 0 && status[k] < 64) { // (negative: not present / error for that page)
            count[status[k]]++;
            known++;
            on_want += status[k] == want_node;
        }
    if (!known)
        return -1;
    int best = 0;
    for (int k = 1; k < 64; k++)
        if (count[k] > count[best])
            best = k;
    if (major_node)
        *major_node = best;
    if (fraction_on_want)
        *fraction_on_want = want_node >= 0 ? (double)on_want / known : 0;
    return 0;
}

} // extern "C"
