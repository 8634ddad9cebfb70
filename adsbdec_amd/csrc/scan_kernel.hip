// scan_kernel.hip -- the fused demodulation kernel for gfx950 (MI355X).
//
// One launch evaluates EVERY preamble offset g in [g_begin, g_end) of a stream of
// real uint16 20 MS/s samples and appends a record for each offset at which the
// reference would have found a CRC-valid frame had it visited that offset:
//
//   air.c:54-92    u16 -> f32, fs/4 sign, 14-tap FIR (7 I + 7 Q), |.|^2, /2 decimation
//   demod.c:102-107 preamble test   p1 > 2*s1 && p2 > 2*s2  (f32 add -> int)
//   demod.c:46-81   DF gate on byte 0 (DF11 / DF17 / DF18 with -a)
//   demod.c:31-44   PPM slicer, 8 strict '>' compares per byte
//   crc.h:36-42     CRC-24 residual == 0 (valid.c:51,73)
//
// The greedy skip, ts and the end-of-file horizon are sequential and are replayed
// on the host over these sparse records (resolver.hpp).
//
// Layout / mapping (HBM-bound integer+f32 scan; no MFMA -- there is no contraction):
//   * input is read once as (I,Q) uint16 PAIRS, one dword each, 16 B per lane-load;
//   * a workgroup owns kTileG consecutive offsets and computes kTileA = kTileG + halo
//     power samples into LDS (f32); the halo (1204 >= 1196) is the reach of one
//     long-frame evaluation, so tiles are independent and no power sample ever
//     goes to HBM;
//   * a thread computes a RUN of 28 consecutive power samples. 28 = 4 x 7 keeps the
//     FIR's summation order -- which in the reference depends on (sample index mod
//     14), i.e. on m mod 7 -- a compile-time property of each output,
//     so the 7 rounding orders are straight-line code, and 28 dwords = 112 B keeps
//     every LDS access a conflict-free 16-byte one (28*t mod 64 hits 16 distinct
//     4-bank slots for any 16 lanes with distinct t mod 16);
//   * arithmetic is strict binary32: multiply, then add (file is built with
//     -ffp-contract=off; the ISA is checked for the absence of v_fma/v_mac).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernel.h"

namespace adsb {

namespace {

// air.c:36-45. Each tap is (float)<double literal>, as in the reference's
// `static const float dsfilter[] = { 0.012627, ... }`.
template <int K>
__device__ __forceinline__ constexpr float tap()
{
    constexpr double lit[14] = {0.012627, 0.025254, 0.037881, 0.050508, 0.063135,
                                0.075761, 0.088388, 0.088388, 0.075761, 0.063135,
                                0.050508, 0.037881, 0.025254, 0.012627};
    return (float)lit[K];
}

// One product-accumulate step of output J (J = index inside the 28-run).
//
// Output m uses pairs m-6..m; the pair of AGE a (a = 0 newest) meets taps
// T[12-2a] (I) and T[13-2a] (Q) (air.c:69-75 with o = 14 - fidx%14).  The
// reference adds in physical ring order k = 0,2,..,12: the pair whose index is a
// multiple of 7 first, then forward in time to the newest, then the wrapped older
// ones: ages p, p-1, .., 0, 6, 5, .., p+1 with p = m mod 7.
// vi/vq hold the run's pairs with the fs/4 sign already applied: slot s = rel+6.
template <int J, int STEP>
__device__ __forceinline__ void fir_step(const float (&vi)[34], const float (&vq)[34], float &si,
                                         float &sq)
{
    constexpr int p = J % 7;
    constexpr int age = (STEP <= p) ? (p - STEP) : (6 - (STEP - p - 1));
    constexpr int slot = J - age + 6;
    constexpr float ti = tap<12 - 2 * age>();
    constexpr float tq = tap<13 - 2 * age>();
    if constexpr (STEP == 0) {
        si = ti * vi[slot]; // 0.0f + x == x up to the sign of zero, which the square erases
        sq = tq * vq[slot];
    } else {
        si = si + ti * vi[slot];
        sq = sq + tq * vq[slot];
    }
}

template <int J>
__device__ __forceinline__ float power_sample(const float (&vi)[34], const float (&vq)[34])
{
    float si, sq;
    fir_step<J, 0>(vi, vq, si, sq);
    fir_step<J, 1>(vi, vq, si, sq);
    fir_step<J, 2>(vi, vq, si, sq);
    fir_step<J, 3>(vi, vq, si, sq);
    fir_step<J, 4>(vi, vq, si, sq);
    fir_step<J, 5>(vi, vq, si, sq);
    fir_step<J, 6>(vi, vq, si, sq);
    return si * si + sq * sq; // air.c:76,91
}

template <int J0>
__device__ __forceinline__ float4 power_quad(const float (&vi)[34], const float (&vq)[34])
{
    return make_float4(power_sample<J0>(vi, vq), power_sample<J0 + 1>(vi, vq),
                       power_sample<J0 + 2>(vi, vq), power_sample<J0 + 3>(vi, vq));
}

// demod.c:31-44: bit i of a byte starting at power index l is a[l+10i] > a[l+10i+5].
__device__ __forceinline__ uint32_t slice_byte(const float *a)
{
    uint32_t b = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        b |= (a[10 * i] > a[10 * i + 5]) ? (0x80u >> i) : 0u;
    return b;
}

} // namespace

template <bool kStats>
__global__ __launch_bounds__(kThreads) void scan_kernel(const ScanArgs args)
{
    __shared__ __attribute__((aligned(16))) float a_lds[kTileA];
    __shared__ uint32_t crc_lds[256];

    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)args.g_begin + (int64_t)blockIdx.x * kTileG; // first owned offset

    // crc.h:1-34: MSB-first byte table of generator 0xFFF409, built in place.
    {
        uint32_t c = (uint32_t)tid << 16;
#pragma unroll
        for (int k = 0; k < 8; k++)
            c = (c & 0x800000u) ? ((c << 1) ^ 0xFFF409u) : (c << 1);
        crc_lds[tid] = c & 0xFFFFFFu;
    }

    // ---------------- phase 1: front end, kTileA power samples into LDS ----------------
    // A tile is "interior" when every pair it loads lies inside the buffer.
    const bool interior = (t0 - 8 >= args.p_lo) && (t0 + kTileA <= args.p_hi);
#pragma unroll 1
    for (int pass = 0; pass < kPasses; pass++) {
        const int run = pass * kThreads + tid;
        const int64_t pr0 = t0 + (int64_t)kRun * run - 8; // first pair loaded; multiple of 4
        uint32_t w[36];
        if (interior) {
            const uint4 *src = reinterpret_cast<const uint4 *>(args.x + (pr0 - args.pbuf0));
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const uint4 q = src[k];
                w[4 * k + 0] = q.x;
                w[4 * k + 1] = q.y;
                w[4 * k + 2] = q.z;
                w[4 * k + 3] = q.w;
            }
        } else {
            // Stream start (the ring is zero-initialised, air.c:33: a missing pair
            // is 0x0800,0x0800 -> v = 0) and the ragged end of a buffer.
#pragma unroll
            for (int k = 0; k < 36; k++) {
                const int64_t pr = pr0 + k;
                w[k] = (pr >= args.p_lo && pr < args.p_hi) ? args.x[pr - args.pbuf0] : 0x08000800u;
            }
        }

        // air.c:64-67,79-82: v = (float)x - 2048; pairs with odd index are negated
        // (samples n mod 4 in {2,3}).  The run starts at an even pair index, so the
        // sign is a compile-time property of the slot.  -(x-2048) == 2048-x exactly.
        float vi[34], vq[34];
#pragma unroll
        for (int s = 0; s < 34; s++) {
            const uint32_t d = w[s + 2];
            const float fi = (float)(d & 0xFFFFu);
            const float fq = (float)(d >> 16);
            if ((s & 1) == 0) { // slot s <-> rel pair s-6: same parity
                vi[s] = fi - 2048.0f;
                vq[s] = fq - 2048.0f;
            } else {
                vi[s] = 2048.0f - fi;
                vq[s] = 2048.0f - fq;
            }
        }

        float4 *dst = reinterpret_cast<float4 *>(a_lds + kRun * run);
        dst[0] = power_quad<0>(vi, vq);
        dst[1] = power_quad<4>(vi, vq);
        dst[2] = power_quad<8>(vi, vq);
        dst[3] = power_quad<12>(vi, vq);
        dst[4] = power_quad<16>(vi, vq);
        dst[5] = power_quad<20>(vi, vq);
        dst[6] = power_quad<24>(vi, vq);
    }
    __syncthreads();

    // ---------------- phase 2: preamble test for 28 offsets per thread ----------------
    const int64_t owned_end = (int64_t)args.g_end - t0; // offsets of this tile below g_end
#pragma unroll 1
    for (int pass = 0; pass < kPasses; pass++) {
        const int run = pass * kThreads + tid;
        const int gl0 = kRun * run;
        int64_t nvalid = kTileG - gl0;
        if (owned_end - gl0 < nvalid)
            nvalid = owned_end - gl0;
        if (nvalid <= 0)
            continue;

        float f[76];
        {
            const float4 *src = reinterpret_cast<const float4 *>(a_lds + gl0);
#pragma unroll
            for (int k = 0; k < 19; k++) {
                const float4 q = src[k];
                f[4 * k + 0] = q.x;
                f[4 * k + 1] = q.y;
                f[4 * k + 2] = q.z;
                f[4 * k + 3] = q.w;
            }
        }
        // demod.c:102-105: p1 = a[g]+a[g+10], s1 = a[g+5]+a[g+15], p2 = a[g+35]+a[g+45],
        // s2 = a[g+30]+a[g+40]; every one of them is c[k] = (int)(a[k] + a[k+10]).
        int c[63];
#pragma unroll
        for (int k = 0; k < 63; k++)
            c[k] = __float2int_rz(f[k] + f[k + 10]);
        uint32_t mask = 0;
#pragma unroll
        for (int j = 0; j < kRun; j++) {
            const bool hit = (c[j] > 2 * c[j + 5]) && (c[j + 35] > 2 * c[j + 30]); // SN = 2
            mask |= hit ? (1u << j) : 0u;
        }
        if (nvalid < kRun)
            mask &= (1u << (int)nvalid) - 1u;

        // ------------- phase 3: DF gate, slicer, CRC for the offsets that passed -------------
        while (mask) {
            const int j = __ffs(mask) - 1;
            mask &= mask - 1;
            const int gl = gl0 + j;
            const float *a = a_lds + gl;

            const uint32_t b0 = slice_byte(a + 80); // demod.c:109-112
            const uint32_t dfv = b0 >> 3;
            int nbytes;
            uint32_t code;
            if (dfv == 11) { // demod.c:72-77
                nbytes = 7;
                code = 0;
            } else if (dfv == 17) { // demod.c:64-67
                nbytes = 14;
                code = 1;
            } else if (dfv == 18 && args.df18) { // demod.c:57-62
                nbytes = 14;
                code = 2;
            } else {
                continue; // demod.c:113-116
            }
            const uint32_t g_rel = (uint32_t)(t0 - (int64_t)args.g_begin) + (uint32_t)gl;
            if (kStats) { // valid.c:46,68 count every DF-gate pass that is visited
                const uint32_t slot = atomicAdd(&args.counters[1], 1u);
                if (slot < args.try_cap)
                    args.tries[slot] = (g_rel << 2) | code;
            }

            // valid.c:49-51 / 71-73: table CRC over the first n-3 bytes, xor last three
            uint32_t crc = crc_lds[b0]; // CrcStep(b0, 0) == table[b0]
            uint32_t tail = 0;
            for (int k = 1; k < nbytes; k++) {
                const uint32_t b = slice_byte(a + 80 + 80 * k);
                if (k < nbytes - 3)
                    crc = (crc << 8) ^ crc_lds[(b ^ (crc >> 16)) & 0xFFu];
                else
                    tail = (tail << 8) | b;
            }
            if (((crc & 0xFFFFFFu) ^ tail) != 0)
                continue;

            // CRC-valid: emit {g_rel, pw, frame[14], len}
            const int p1 = __float2int_rz(a[0] + a[10]);
            const int p2 = __float2int_rz(a[35] + a[45]);
            const uint32_t pw = (uint32_t)((p1 + p2) / 4); // demod.c:127,133
            const uint32_t slot = atomicAdd(&args.counters[0], 1u);
            if (slot < args.cand_cap) {
                uint32_t *rec = args.cands + (size_t)slot * kCandWords;
                uint32_t wds[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 14; k++) {
                    const uint32_t b = (k < nbytes) ? slice_byte(a + 80 + 80 * k) : 0u;
                    wds[k >> 2] |= b << (8 * (k & 3));
                }
                wds[3] |= (uint32_t)nbytes << 16;
                rec[0] = g_rel;
                rec[1] = pw;
                rec[2] = wds[0];
                rec[3] = wds[1];
                rec[4] = wds[2];
                rec[5] = wds[3];
            }
        }
    }
}

hipError_t launch_scan(const ScanArgs &args, bool stats, hipStream_t stream)
{
    if (args.g_end <= args.g_begin)
        return hipSuccess;
    const uint64_t n = args.g_end - args.g_begin;
    const uint64_t blocks = (n + kTileG - 1) / kTileG;
    if (stats)
        hipLaunchKernelGGL(scan_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, args);
    else
        hipLaunchKernelGGL(scan_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, args);
    return hipGetLastError();
}

} // namespace adsb
