// scan_kernel.hip -- the fused demodulation kernel for gfx950 (MI355X).
//
// One launch evaluates EVERY preamble offset g in [g_begin, g_end) of a stream of
// real uint16 20 MS/s samples and appends a record for each offset at which the
// reference would have found a CRC-valid frame had it visited that offset:
//
//   air.c:54-92     u16 -> f32, fs/4 sign, 14-tap FIR (7 I + 7 Q), |.|^2, /2 decimation
//   demod.c:102-107 preamble test   p1 > 2*s1 && p2 > 2*s2  (f32 add -> int)
//   demod.c:46-81   DF gate on byte 0 (DF11 / DF17 / DF18 with -a)
//   demod.c:31-44   PPM slicer, 8 strict '>' compares per byte
//   crc.h:36-42     CRC-24 residual == 0 (valid.c:51,73)
//
// The greedy skip, ts and the end-of-file horizon are sequential and are replayed
// on the host over these sparse records (resolver.hpp).
//
// Structure (a streaming integer+f32 scan, VALU-bound on gfx950; no MFMA -- there is no
// contraction, and every product and sum must be rounded separately):
//
//  Stage A (all the arithmetic; no barriers, no divergence).  A thread computes a
//  RUN of 28 consecutive power samples in registers.  28 = 4 x 7 keeps the FIR's
//  summation order -- which in the reference depends on (sample index mod 14), i.e.
//  on m mod 7 -- a compile-time property of each output, so the seven
//  rounding orders are straight-line code with immediate taps.  Every comparison
//  the reference will ever make on those samples is then reduced to ONE BIT per
//  power sample, packed 28 to a word ("bit planes"):
//      D [m] = a[m]   > a[m+5]                (every slicer / DF-gate decision)
//      E1[m] = c[m]   > 2*c[m+5]              (p1 > 2*s1 for the offset g = m)
//      E2[m] = c[m+5] > 2*c[m]                (p2 > 2*s2 for the offset g = m-30)
//  with c[m] = (int)(a[m] + a[m+10]) -- all four preamble sums of demod.c:102-105
//  have that form.  The 16 samples a thread needs from the following run come from
//  the next lane by DPP (wave_shl:1); lane 63 of a wave re-computes the first run
//  of the next wave (1/64 redundancy) so waves never exchange data and Stage A has
//  no barrier.  Power samples never leave registers; LDS receives 12 bytes per 28
//  samples.
//
//  Stage B (bit logic).  For the 28 offsets of a run the preamble test and the DF
//  gate are ~30 word-wide bit operations on funnel-shifted plane words (SIMD within
//  a register: no per-offset branch).  Survivors (~0.5 % of offsets on noise) are
//  compacted through an LDS queue, so the slicer runs with dense lanes.  The slicer
//  gathers the 112 frame bits as 14 columns of 8 bits (bit k = 14b + c sits in word
//  +5b at a fixed bit position, because 140 = 5 x 28) and checks the CRC as the XOR
//  of 14 syndrome-table lookups; only CRC-valid offsets (~1e-4) take the slow path
//  that rebuilds the bytes in order and recomputes pw from the input samples.
//  CRC-valid candidates of the tile are filtered (never-visited rule), ranked, and leave
//  as one write-through store of adjacent lanes into the host's hand-off stream
//  (scan_kernel.h), or -- fallbacks -- through the launch-wide loose list.
//
//  One workgroup of four waves per tile: Stage A, a barrier, Stage B between barriers.  (A pipelined variant -- persistent
//  five-wave workgroups, Stage B of tile t on a wave of its own beside Stage A of tile t + 1 -- was built in round 3,
//  bit-identical and 45 % slower; it and the other experiments that lost are in the history of this file, see
//  DESIGN_HISTORY.md.)
//
//  A workgroup owns owned_runs(K) = 252 K - 44 runs and computes 252 K (+1): the halo
//  (the 1196-sample reach of a long frame) costs 44 runs of planes per tile (2 % at
//  K = 8) instead of 1204 float samples of LDS.
//
// Arithmetic rounds like strict binary32 multiply, then add (built with -ffp-contract=off: the compiler contracts
// nothing).  The fused operations in the ISA are written here, each rounding the real number the reference's separate
// operation rounds: the 56 sign tests of the preamble comparison, the FIR's products on the converted sample and the
// accumulations of its twelve shared products ("The FIR of one run" below); tests/test_build_flags.py counts them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "scan_kernel.h"
#include "slicer_bits.h"
#include "scan_stamps.h" // ADSB_STAMP / ADSB_COUNT: nothing in the shipped build (a measurement build's per-phase tile clocks)

namespace adsb {

static_assert(5 * lds_bytes(7) <= 160 * 1024, "five workgroups of a K = 7 tile share a CU's LDS");

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

// The FIR of one run: 28 outputs, each the sum of seven (I, Q) products in the reference's order.
//
// Output m uses pairs m-6..m; the pair of AGE a (a = 0 newest) meets taps T[12-2a] (I) and T[13-2a] (Q) (air.c:69-75 with
// o = 14 - fidx%14).  The reference adds in physical ring order k = 0,2,..,12: the pair whose index is a multiple of 7
// first, then forward in time to the newest, then the wrapped older ones: ages p, p-1, .., 0, 6, 5, .., p+1 with
// p = m mod 7.  Seven static orders, straight-line code.
//
// The I and Q sums are the two halves of ONE packed-f32 register pair (v_pk_*_f32: two results per lane for the issue slot
// of one; tools/valu_bench.hip).  Every product is rounded, then every sum: binary32 multiply and binary32 add, as in the
// reference -- written with two fused forms that round the SAME real numbers (round 6; parity bit for bit on every test
// and fuzz run, profiles/r6_ab_runs.txt section 8):
//
//  (1) vv[slot] is the converted sample x itself (what the typed load delivers), not x - 2048 (air.c:64-67).  For a tap t,
//      t (x - 2048) = t x - 2048 t as real numbers and 2048 t is a binary32 number (a power of two times t), so
//      fma(t, x, -2048 t) rounds exactly the real number the reference's multiplication rounds: the 34 subtractions of a
//      run are gone.  The fs/4 sign (pairs with odd index are negated, air.c:79-82; a compile-time property of the slot)
//      is the sign of both constants: operand modifiers.  An instruction reads ONE scalar operand, so 2048 t sits in
//      vector registers: four pairs -- ages 4..6 use the pairs of ages 2..0 with the halves exchanged (the filter is
//      symmetric: op_sel), and they are written by volatile moves at the head of each pass so that they do not live
//      across the plane arithmetic, where all 96 registers are taken.
//
//  (2) (float)0.025254 is exactly 2 x (float)0.012627 (a power-of-two multiple of a decimal literal rounds to the same
//      multiple), so fl(2 h0 x) = 2 fl(h0 x) and  s + fl(2 h0 x)  rounds like  fma(fl(h0 x), 2, s).  The pair that is the
//      NEWEST of output J (age 0: taps 2 h0, h0) is the OLDEST of output J + 6 (age 6: taps h0, 2 h0), and both add it in
//      the same step (J mod 7 = p: step p; (J + 6) mod 7 = p - 1: age 6 comes at step p): ONE product M = h0 (I, Q) serves
//      both, each through one fused multiply-add by (2, 1) or (1, 2).  The first step of an output has no addition
//      (p = 0 / p = 6): nothing to share there.  Outputs J and J + 6 have to be advanced together for this: the plan below.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// kc[0..3] = 2048 (T[12-2b], T[13-2b]) in vector registers, kc[4..7] = (T[12-2b], T[13-2b]) in scalar ones, b = 0..3
struct FirConsts {
    f32x2 c2048[4];
    f32x2 taps[4];
};

// t (x - 2048) or -t (x - 2048) for the pair in `slot` and the taps of age `age`, from the converted sample x
template <int SLOT, int AGE>
__device__ __forceinline__ f32x2 fir_product(const f32x2 (&vv)[34], const FirConsts &k)
{
    constexpr int b = AGE <= 3 ? AGE : 6 - AGE;
    const f32x2 c = AGE <= 3 ? k.c2048[b] : __builtin_shufflevector(k.c2048[b], k.c2048[b], 1, 0);
    const f32x2 t = AGE <= 3 ? k.taps[b] : __builtin_shufflevector(k.taps[b], k.taps[b], 1, 0);
    if constexpr (SLOT & 1)
        return __builtin_elementwise_fma(-t, vv[SLOT], c);
    else
        return __builtin_elementwise_fma(t, vv[SLOT], -c);
}

// The order the 28 outputs are computed in, and the groups advanced together (independent accumulation chains interleaved in
// program order, so that a dependent add rarely issues right behind the product it consumes).  The first eight outputs go
// in index order: every input pair is still live there (68 registers), and an output that is finished frees the pair only
// it still needed.  From output 8 on the registers that have come free pay for chain order (J, J + 6, J + 12, ..): 12 of
// the 18 shareable products are shared.  (Chain order from output 4 or 6 shares 13-15 and spills; measured plans in
// profiles/r6_ab_runs.txt section 8.)
struct FirPlan {
    int out[28];   // position -> output
    int group[28]; // position -> first position of its group
    int size[28];  // first position of a group -> outputs in the group
};
__host__ __device__ inline constexpr FirPlan fir_plan()
{
    FirPlan pl{};
    constexpr int groups[6][6] = {{0, 1, 2, 3, -1, -1},     {4, 5, 6, 7, -1, -1},     {8, 14, 20, 26, -1, -1},
                                  {9, 15, 21, 27, -1, -1},  {10, 16, 22, 11, 17, 23}, {12, 18, 24, 13, 19, 25}};
    int pos = 0;
    for (int g = 0; g < 6; g++) {
        const int first = pos;
        for (int k = 0; k < 6 && groups[g][k] >= 0; k++) {
            pl.out[pos] = groups[g][k];
            pl.group[pos] = first;
            pos++;
        }
        pl.size[first] = pos - first;
    }
    return pl;
}

// One step of the output at position POS of the plan
template <int POS, int STEP>
__device__ __forceinline__ void fir_step(const f32x2 (&vv)[34], f32x2 &s, f32x2 &shared, const FirConsts &k)
{
    constexpr FirPlan plan = fir_plan();
    constexpr int J = plan.out[POS];
    constexpr int p = J % 7;
    constexpr int age = (STEP <= p) ? (p - STEP) : (6 - (STEP - p - 1));
    constexpr int slot = J - age + 6; // slot s <-> pair s - 6 of the run
    // the partner is the neighbour in the plan, and only a neighbour inside the same group is advanced in the same step
    constexpr int next = POS + 1 < 28 ? POS + 1 : POS, prev = POS > 0 ? POS - 1 : POS;
    constexpr bool next_is_partner = next != POS && plan.out[next] == J + 6 && plan.group[next] == plan.group[POS];
    constexpr bool prev_is_partner = prev != POS && plan.out[prev] == J - 6 && plan.group[prev] == plan.group[POS];
    static_assert(tap<12>() == 2.0f * tap<13>() && tap<1>() == 2.0f * tap<0>() && tap<0>() == tap<13>(), "the exact doubling of (2)");
    if constexpr (age == 0 && STEP > 0 && next_is_partner) {
        // M = +-h0 (x - 2048) for both halves: the pair (h0, h0) is the high half of the age-0 constants, twice
        const f32x2 c = __builtin_shufflevector(k.c2048[0], k.c2048[0], 1, 1), h = __builtin_shufflevector(k.taps[0], k.taps[0], 1, 1);
        if constexpr (slot & 1)
            shared = __builtin_elementwise_fma(-h, vv[slot], c);
        else
            shared = __builtin_elementwise_fma(h, vv[slot], -c);
        constexpr f32x2 two_one = {2.0f, 1.0f};
        s = __builtin_elementwise_fma(shared, two_one, s);
    } else if constexpr (age == 6 && STEP > 0 && prev_is_partner) {
        constexpr f32x2 one_two = {1.0f, 2.0f};
        s = __builtin_elementwise_fma(shared, one_two, s); // made by output J - 6 in this step
    } else if constexpr (STEP == 0) {
        s = fir_product<slot, age>(vv, k); // 0.0f + x == x up to the sign of zero, which the square erases
    } else {
        s = s + fir_product<slot, age>(vv, k);
    }
}

template <int POS0, int N, int STEP>
__device__ __forceinline__ void fir_group_step(const f32x2 (&vv)[34], f32x2 (&s)[N], const FirConsts &k)
{
    if constexpr (STEP < 7) {
        f32x2 shared = {0.0f, 0.0f};
        fir_step<POS0, STEP>(vv, s[0], shared, k);
        if constexpr (N > 1) fir_step<POS0 + 1, STEP>(vv, s[1], shared, k);
        if constexpr (N > 2) fir_step<POS0 + 2, STEP>(vv, s[2], shared, k);
        if constexpr (N > 3) fir_step<POS0 + 3, STEP>(vv, s[3], shared, k);
        if constexpr (N > 4) fir_step<POS0 + 4, STEP>(vv, s[4], shared, k);
        if constexpr (N > 5) fir_step<POS0 + 5, STEP>(vv, s[5], shared, k);
        fir_group_step<POS0, N, STEP + 1>(vv, s, k);
    }
}

// a[0..27] = the run's power samples (air.c:76,91)
template <int POS0>
__device__ __forceinline__ void power_run(const f32x2 (&vv)[34], float *a, const FirConsts &k)
{
    if constexpr (POS0 < 28) {
        constexpr FirPlan plan = fir_plan();
        constexpr int n = plan.size[POS0];
        f32x2 s[n];
        fir_group_step<POS0, n, 0>(vv, s, k);
#pragma unroll
        for (int i = 0; i < n; i++) {
            const f32x2 sq = s[i] * s[i];
            // A plain `sq.x + sq.y` gets SLP-packed across two outputs at the price of three transposing moves per pair;
            // keep it one scalar add.
            float r;
            asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(sq.x), "v"(sq.y));
            a[plan.out[POS0 + i]] = r;
        }
        power_run<POS0 + n>(vv, a, k);
    }
}

// Same arithmetic for power samples at RUN-TIME indices (rare path: pw of a
// CRC-valid candidate needs a[g], a[g+10], a[g+35], a[g+45]).  Rounds exactly like
// power_sample<>: same products, same order (one static order per phase p = m mod
// 7), first product not added to zero.  All loads are issued before any use: the
// whole workgroup waits for this path at the next barrier.
template <int P>
__device__ __forceinline__ float power_ordered(const f32x2 (&pr)[7])
{
    // pr[a] = (T[12-2a], T[13-2a]) * (I, Q) of the pair of age a; order p, p-1, .., 0, 6, .., p+1
    f32x2 s = pr[P];
#pragma unroll
    for (int a = P - 1; a >= 0; a--)
        s = s + pr[a];
#pragma unroll
    for (int a = 6; a > P; a--)
        s = s + pr[a];
    const f32x2 sq = s * s;
    return sq.x + sq.y;
}

// The 28 pairs behind the four power samples of demod.c:102-105 at offset g (0, +10, +35, +45; seven pairs each): the
// loads on their own, so that a caller can issue them early and do other work while they are in flight.
__device__ __forceinline__ void pw_load(const uint32_t *__restrict__ x, int64_t pbuf0, int64_t p_lo, int64_t p_hi, int64_t g,
                                        uint32_t (&raw)[4][7])
{
    const int off[4] = {0, 10, 35, 45}; // demod.c:102-105
    if (g - 6 >= p_lo && g + 45 < p_hi) {
        // the usual case, every pair inside the buffer: one address, 28 loads at immediate offsets
        const uint32_t *b = x + (g - pbuf0);
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int a = 0; a < 7; a++)
                raw[k][a] = b[off[k] - a];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int a = 0; a < 7; a++) {
                const int64_t pi = g + off[k] - a;
                raw[k][a] = (pi >= p_lo && pi < p_hi) ? x[pi - pbuf0] : 0x08000800u;
            }
    }
}

__device__ __forceinline__ uint32_t pw_compute(const uint32_t (&raw)[4][7], int64_t g)
{
    const int off[4] = {0, 10, 35, 45};
    // power indices stay below 2^31 (streams of < 2^32 samples): 32-bit arithmetic for phase and parity
    const uint32_t g32 = (uint32_t)g, g7 = g32 % 7u;
    float pw_s[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t m = g32 + (uint32_t)off[k];
        const int p = (int)((g7 + (uint32_t)off[k]) % 7u);
        const f32x2 taps[7] = {{tap<12>(), tap<13>()}, {tap<10>(), tap<11>()}, {tap<8>(), tap<9>()},
                               {tap<6>(), tap<7>()},   {tap<4>(), tap<5>()},   {tap<2>(), tap<3>()},
                               {tap<0>(), tap<1>()}};
        f32x2 pr[7];
#pragma unroll
        for (int a = 0; a < 7; a++) {
            const uint32_t d = raw[k][a];
            const f32x2 f = {(float)(d & 0xFFFFu), (float)(d >> 16)};
            const f32x2 mid = {2048.0f, 2048.0f};
            const f32x2 v = (((m - a) & 1) == 0) ? (f - mid) : (mid - f); // fs/4 sign of the pair
            pr[a] = taps[a] * v;
        }
        float r;
        switch (p) {
        case 0: r = power_ordered<0>(pr); break;
        case 1: r = power_ordered<1>(pr); break;
        case 2: r = power_ordered<2>(pr); break;
        case 3: r = power_ordered<3>(pr); break;
        case 4: r = power_ordered<4>(pr); break;
        case 5: r = power_ordered<5>(pr); break;
        default: r = power_ordered<6>(pr); break;
        }
        pw_s[k] = r;
    }
    const int p1 = __float2int_rz(pw_s[0] + pw_s[1]);
    const int p2 = __float2int_rz(pw_s[2] + pw_s[3]);
    return (uint32_t)((p1 + p2) / 4); // demod.c:127,133
}

__device__ __noinline__ uint32_t pw_at(const uint32_t *__restrict__ x, int64_t pbuf0, int64_t p_lo, int64_t p_hi, int64_t g)
{
    uint32_t raw[4][7];
    pw_load(x, pbuf0, p_lo, p_hi, g, raw);
    return pw_compute(raw, g);
}

// The slicer gathers the frame as 14 column bytes (frame bit k = 14 b + c is bit b
// of column c; four columns per word).  Rebuild the 14 frame bytes in order (bit k
// is bit 7 - k%8 of byte k/8), packed little-endian into wds[0..3], with the
// length in byte 14.  A static 112-bit transpose: 2 operations per bit.
__device__ __forceinline__ void columns_to_bytes(const uint32_t (&cw)[4], bool is_short, uint32_t (&wds)[4])
{
    wds[0] = wds[1] = wds[2] = wds[3] = 0;
#pragma unroll
    for (int k = 0; k < 112; k++) {
        const int b = k / 14, c = k % 14;
        const uint32_t bit = (cw[c >> 2] >> (8 * (c & 3) + b)) & 1u;
        const int n = k >> 3;
        wds[n >> 2] |= bit << (8 * (n & 3) + 7 - (k & 7));
    }
    if (is_short) { // DF11: 56 bits = 7 bytes
        wds[1] &= 0x00FFFFFFu;
        wds[2] = 0;
        wds[3] = 0;
    }
    wds[3] |= (is_short ? 7u : 14u) << 16;
}

// One 16-byte granule of the hand-off stream, written THROUGH to host memory (sc0 sc1).
// A plain store may sit in the L2 until its line is evicted or the kernel ends (measured:
// single tiles reaching the host ~30 us after their neighbours, which stalls the host's
// in-order resolver and leaves it a burst of work at the very end).  Write-through
// stores are a scarce resource, though -- the device retires only ~40 M of them per
// second, whatever their size (measured: two 8-byte ones per granule from the threads
// that finish the records made the kernel 11x slower) -- so a tile writes its whole range
// with ONE instruction of adjacent lanes: a few line-sized requests per tile.
// No ordering is implied or needed: the tile's checksum validates the bytes.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_granule_through(uint32_t *hand, uint32_t gran, u32x4 v)
{
    u32x4 *dst = reinterpret_cast<u32x4 *>(hand) + gran;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(v) : "memory");
}

// acc = (acc << 1) | sign(v): one v_alignbit_b32
__device__ __forceinline__ uint32_t push_sign(uint32_t acc, uint32_t v)
{
    return __builtin_amdgcn_alignbit(acc, v, 31);
}

// value held by lane+1 (DPP wave_shl:1); lane 63 receives 0
__device__ __forceinline__ float from_next_lane(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}

// v_bitop3_b32 (gfx950): EXPR over the words A, B, C; the immediate is EXPR's truth table
#define ADSB_BITOP3(a, b, c, EXPR) \
    __builtin_amdgcn_bitop3_b32((a), (b), (c), []() constexpr { constexpr uint32_t A = 0xF0u, B = 0xCCu, C = 0xAAu; return (uint32_t)((EXPR) & 0xFFu); }())

// the same without the final mask: bits 28..31 of the result are not meaningful
template <int POS>
__device__ __forceinline__ uint32_t take28u(const uint32_t *w)
{
    constexpr int k = POS / 28, s = POS % 28;
    uint32_t r = w[k] >> s;
    if constexpr (s != 0)
        r |= w[k + 1] << (28 - s);
    return r;
}

// 28 bits starting at bit `POS` of the stream w[0] | w[1]<<28 | w[2]<<56 | ... (28 valid bits per word)
template <int POS>
__device__ __forceinline__ uint32_t take28(const uint32_t *w)
{
    constexpr int k = POS / 28, s = POS % 28;
    uint32_t r = w[k] >> s;
    if constexpr (s != 0)
        r |= w[k + 1] << (28 - s);
    return r & 0x0FFFFFFFu;
}

} // namespace

// ------------------------------ Stage A ------------------------------
// One tile's arithmetic for one of the tile's four Stage A waves: K passes, a run of 28 power samples per lane and
// pass, three plane words per run into LDS.  No barrier, no divergence.
__device__ __forceinline__ void stage_a(const uint32_t *__restrict__ xin, const int64_t pbuf0, const int64_t p_lo,
                                        const int64_t p_hi, const int64_t t0, const int K, const int wave, const int lane,
                                        uint32_t *pl_d, uint32_t *pl_e1, uint32_t *pl_e2)
{
    // first run of wave w in pass ps: a pass of the four waves is 252 consecutive runs
    auto first_run = [&](int ps) { return kWaveRuns * (kWaves * ps + wave); };
    // Input: the 34 pairs (6 of pre-halo + 28) a run needs are 17 TYPED buffer loads of 8
    // bytes per lane (buffer_load_format_xyzw, data format 16_16_16_16, number format
    // USCALED): the load path itself converts the four uint16 to four floats -- exactly, and
    // for free next to 66 v_cvt_f32_u32 per run (4.4 cycles each; kernel -2 .. -4.5 %).  Lanes are 112 bytes apart; the buffer resource is rebuilt per
    // wave and pass around the wave's own 7 KiB window, so no buffer size limit applies.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    auto pass_first_pair = [&](int ps) { return t0 + (int64_t)kRun * first_run(ps) - 8; }; // lane 0's
#pragma unroll 1
    for (int pass = 0; pass < K; pass++) {
        // the FIR's constants (fir_product): the vector ones written here, by instructions the compiler cannot hoist -- eight
        // registers that live from here to the end of the FIR, not across the plane arithmetic behind it; the scalar ones
        // made opaque INSIDE the loop, so that their exchanged and negated forms are operand modifiers, not hoisted copies
        FirConsts kc = {{{0, 0}, {0, 0}, {0, 0}, {0, 0}}, {{tap<12>(), tap<13>()}, {tap<10>(), tap<11>()}, {tap<8>(), tap<9>()}, {tap<6>(), tap<7>()}}};
#define ADSB_KC(b, T0, T1)                                                                                                          \
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"                                                                            \
                 : "=v"(kc.c2048[b].x), "=v"(kc.c2048[b].y)                                                                         \
                 : "i"(__builtin_bit_cast(uint32_t, 2048.0f * tap<T0>())), "i"(__builtin_bit_cast(uint32_t, 2048.0f * tap<T1>())))
        ADSB_KC(0, 12, 13);
        ADSB_KC(1, 10, 11);
        ADSB_KC(2, 8, 9);
        ADSB_KC(3, 6, 7);
#undef ADSB_KC
#pragma unroll
        for (int b = 0; b < 4; b++)
            asm volatile("" : "+s"(kc.taps[b]));
        const int v0 = first_run(pass); // first run of this wave in this pass
        const int v = v0 + lane;
        const int64_t wlo = pass_first_pair(pass);
        // wave-uniform: every pair this wave loads lies inside the buffer
        const bool interior = (wlo >= p_lo) && (wlo + kRun * 64 + 8 <= p_hi);
        f32x4 tl[17]; // tl[k]: pairs 2k+2, 2k+3 of the lane's 36 = slots 2k, 2k+1
        if (interior) {
            const uint64_t wbase = (uint64_t)(xin + (wlo - pbuf0));
            i32x4 rs;
            rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)wbase);
            rs.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(wbase >> 32) & 0xFFFFu)); // stride 0: raw buffer
            rs.z = 64 * kRun * 4 + 64;                                                        // bytes
            rs.w = (int)(4u | 5u << 3 | 6u << 6 | 7u << 9 /* dst_sel xyzw */ | 2u << 12 /* USCALED */ | 12u << 15 /* 16_16_16_16 */);
            const int voff = lane * (kRun * 4);
            asm volatile("buffer_load_format_xyzw %0, %17, %18, 0 offen offset:8\n\t"
                         "buffer_load_format_xyzw %1, %17, %18, 0 offen offset:16\n\t"
                         "buffer_load_format_xyzw %2, %17, %18, 0 offen offset:24\n\t"
                         "buffer_load_format_xyzw %3, %17, %18, 0 offen offset:32\n\t"
                         "buffer_load_format_xyzw %4, %17, %18, 0 offen offset:40\n\t"
                         "buffer_load_format_xyzw %5, %17, %18, 0 offen offset:48\n\t"
                         "buffer_load_format_xyzw %6, %17, %18, 0 offen offset:56\n\t"
                         "buffer_load_format_xyzw %7, %17, %18, 0 offen offset:64\n\t"
                         "buffer_load_format_xyzw %8, %17, %18, 0 offen offset:72\n\t"
                         "buffer_load_format_xyzw %9, %17, %18, 0 offen offset:80\n\t"
                         "buffer_load_format_xyzw %10, %17, %18, 0 offen offset:88\n\t"
                         "buffer_load_format_xyzw %11, %17, %18, 0 offen offset:96\n\t"
                         "buffer_load_format_xyzw %12, %17, %18, 0 offen offset:104\n\t"
                         "buffer_load_format_xyzw %13, %17, %18, 0 offen offset:112\n\t"
                         "buffer_load_format_xyzw %14, %17, %18, 0 offen offset:120\n\t"
                         "buffer_load_format_xyzw %15, %17, %18, 0 offen offset:128\n\t"
                         "buffer_load_format_xyzw %16, %17, %18, 0 offen offset:136"
                         "\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(tl[0]), "=&v"(tl[1]), "=&v"(tl[2]), "=&v"(tl[3]), "=&v"(tl[4]), "=&v"(tl[5]), "=&v"(tl[6]),
                           "=&v"(tl[7]), "=&v"(tl[8]), "=&v"(tl[9]), "=&v"(tl[10]), "=&v"(tl[11]), "=&v"(tl[12]),
                           "=&v"(tl[13]), "=&v"(tl[14]), "=&v"(tl[15]), "=&v"(tl[16])
                         : "v"(voff), "s"(rs)
                         : "memory");
        } else {
            // Stream start (the ring is zero-initialised, air.c:33: a missing pair is
            // 0x0800,0x0800 -> v = 0) and the ragged end of a buffer: plain loads, converted here
            int64_t pr0 = wlo + (int64_t)kRun * lane;
            asm volatile("" : "+v"(pr0)); // (opaque: keeps the 34 pair indices of this rare path from being precomputed per tile and spilled)
#pragma unroll
            for (int k = 0; k < 17; k++) {
                const int64_t pa = pr0 + 2 * k + 2, pb = pa + 1;
                const uint32_t d0 = (pa >= p_lo && pa < p_hi) ? xin[pa - pbuf0] : 0x08000800u;
                const uint32_t d1 = (pb >= p_lo && pb < p_hi) ? xin[pb - pbuf0] : 0x08000800u;
                tl[k] = f32x4{(float)(d0 & 0xFFFFu), (float)(d0 >> 16), (float)(d1 & 0xFFFFu), (float)(d1 >> 16)};
            }
        }
        // the run's pairs as (I, Q), slot s <-> pair s - 6 of the run: the converted samples as they are -- the - 2048 of
        // air.c:64-67 and the fs/4 sign of air.c:79-82 are inside the FIR's products (fir_product)
        f32x2 vv[34];
#pragma unroll
        for (int s = 0; s < 34; s++) {
            const f32x4 q = tl[s >> 1];
            vv[s] = (s & 1) ? f32x2{q.z, q.w} : f32x2{q.x, q.y};
        }

        // a[0..27]: this run; a[28..43]: the first 16 samples of the next run (next lane)
        float a[44];
        power_run<0>(vv, a, kc);
#pragma unroll
        for (int k = 0; k < 16; k++)
            a[28 + k] = from_next_lane(a[k]);

        // demod.c:102-105: every preamble sum is c[k] = (int)(a[k] + a[k+10]).  The
        // truncated value is kept as a float (v_trunc_f32 == the C conversion for the
        // magnitudes in the input domain); the integer comparisons `c > 2 c'` are
        // decided by the SIGN of 2 c' - c, which one fused multiply-add gives exactly
        // (a single rounding cannot change the sign of a non-zero difference and an
        // exact zero stays zero).
        //
        // Packing: every operation here combines index k with k + 5 or k + 10, so the
        // usual (k, k+1) register pairs cannot feed v_pk_* on both sides (5 is odd).
        // Pairs (k, k+2) for k mod 5 in {0, 1} can -- the partner set is closed under
        // +5 -- and leave k mod 5 == 4 as scalar operations: 4 of 5 values are packed.
        float c[34], dv[28], e1v[28], e2v[28];
#pragma unroll
        for (int k = 0; k < 33; k++) {
            if (k % 5 < 2) {
                const f32x2 lo = {a[k], a[k + 2]}, hi = {a[k + 10], a[k + 12]};
                const f32x2 sum = lo + hi;
                c[k] = __builtin_truncf(sum.x);
                c[k + 2] = __builtin_truncf(sum.y);
            } else if (k % 5 == 4) {
                c[k] = __builtin_truncf(a[k] + a[k + 10]);
            }
        }
#pragma unroll
        for (int m = 0; m < 28; m++) {
            if (m % 5 < 2 && m + 2 < 28) {
                const f32x2 am = {a[m], a[m + 2]}, an = {a[m + 5], a[m + 7]};
                const f32x2 cm = {c[m], c[m + 2]}, cn = {c[m + 5], c[m + 7]};
                const f32x2 two = {2.0f, 2.0f};
                const f32x2 dd = an - am;
                const f32x2 x1 = __builtin_elementwise_fma(cn, two, -cm);
                const f32x2 x2 = __builtin_elementwise_fma(cm, two, -cn);
                dv[m] = dd.x, dv[m + 2] = dd.y;
                e1v[m] = x1.x, e1v[m + 2] = x1.y;
                e2v[m] = x2.x, e2v[m + 2] = x2.y;
            } else if (m % 5 == 4 || (m % 5 < 2 && m + 2 >= 28)) {
                dv[m] = a[m + 5] - a[m];
                e1v[m] = __builtin_fmaf(c[m + 5], 2.0f, -c[m]);
                e2v[m] = __builtin_fmaf(c[m], 2.0f, -c[m + 5]);
            }
        }

        uint32_t d = 0, e1 = 0, e2 = 0;
#pragma unroll
        for (int m = 27; m >= 0; m--) { // bit m of each word <-> sample m of the run
            d = push_sign(d, __float_as_uint(dv[m]));    // a[m+5] - a[m] < 0:  a[m] > a[m+5]   (demod.c:34)
            e1 = push_sign(e1, __float_as_uint(e1v[m])); // 2 c[m+5] - c[m] < 0: c[m] > 2 c[m+5] (SN = 2, demod.c:83)
            e2 = push_sign(e2, __float_as_uint(e2v[m])); // 2 c[m] - c[m+5] < 0: c[m+5] > 2 c[m]
        }
        if (lane < kWaveRuns) { // lane 63 only feeds lane 62
            pl_d[v] = d;
            pl_e1[v] = e1;
            pl_e2[v] = e2;
        }
    }
}

// ------------------------------ Stage B ------------------------------
// Everything behind a tile's planes: gate, survivor queue, slicer + CRC, never-visited filter, ranking, finishing and
// the hand-off, by the tile's four waves between workgroup barriers.
// LDS of Stage B: queue[queue_cap], ctl[16] (qcount, qover, cl_n, cl_over, tile_n, tile_over, tile_base, try_base,
// tile_res, tile_fit, tile_chk[4], tile_lines, tile_sum), cl_rec[clist_cap * kCandWords].
template <bool kStats>
__device__ __forceinline__ void stage_b(const ScanArgs &args, const uint32_t tile, const int K, const int64_t t0, const int tid,
                                        const uint32_t *pl_d, const uint32_t *pl_e1, const uint32_t *pl_e2, uint32_t *queue,
                                        uint32_t *qcount, uint32_t *cl_rec, const int clist_cap, uint64_t &stamp_last)
{
    constexpr int NT = kThreads;
    constexpr int kFallbackChunks = 256 / NT; // a fallback round takes one bit position of 256 runs: <= 256 entries
    const uint32_t *__restrict__ xin = args.x;
    const int64_t pbuf0 = args.pbuf0, p_lo = args.p_lo, p_hi = args.p_hi;
    const int own = kPassRuns * K - kReachRuns;
    uint32_t *tile_n = qcount + 4;    // records this tile keeps (ranked into its hand-off range)
    uint32_t *tile_over = qcount + 5; // bit 0: some records had to go to the loose list; bit 1: tries went to the launch-wide list
    uint32_t *tile_base = qcount + 6; // granule index of the tile's marker in args.hand
    uint32_t *try_base = qcount + 7;  // first index of the round's range in args.tries (kStats)
    uint32_t *tile_res = qcount + 8;  // the tile has reserved its range of args.hand
    uint32_t *tile_fit = qcount + 9;  // ... and the whole range lies inside the array
    uint32_t *tile_chk = qcount + 10; // [4]: XOR of the granules the tile wrote there, word by word
    uint32_t *tile_lines = qcount + 14; // 64-byte lines of args.hand the tile has reserved (its marker says so: kMarkLinesShift)
    uint32_t *tile_sum = qcount + 15;   // rank-weighted sum over the records it wrote there (record_term: the marker's second summary)
    if (tid == 0) { // (the first barrier of the round loop below orders these)
        *tile_n = 0;
        *tile_over = 0;
        *tile_base = 0;
        *tile_res = 0;
        *tile_fit = 0;
        *tile_lines = 0;
        *tile_sum = 0;
        tile_chk[0] = tile_chk[1] = tile_chk[2] = tile_chk[3] = 0;
        qcount[2] = 0; // cl_n, cl_over: the staged list is the TILE's, whatever the number of rounds
        qcount[3] = 0;
    }
    // A finished record that cannot go through the hand-off stream (hand-off disabled,
    // staged list full, stream full) goes to the launch-wide loose list; the tile's marker
    // then tells the host to collect after completion.
    auto emit_loose = [&](uint32_t g_rel, uint32_t pw, const uint32_t (&wds)[4]) {
        if (args.hand)
            atomicOr(tile_over, 1u);
        const uint32_t slot = atomicAdd(&args.counters[0 * kCounterPad], 1u);
        if (slot < args.cand_cap) {
            uint32_t *rec = args.cands + (size_t)slot * kCandWords;
            rec[0] = g_rel;
            rec[1] = pw;
            rec[2] = wds[0];
            rec[3] = wds[1];
            rec[4] = wds[2];
            rec[5] = wds[3];
        }
    };
    const int64_t off_end64 = (int64_t)args.g_end - t0; // offsets of this tile that exist
    const int off_end = off_end64 > (int64_t)kRun * own ? kRun * own : off_end64 < 0 ? 0 : (int)off_end64;
    const uint32_t df18_mask = args.df18 ? ~0u : 0u;
    const int nchunks = (own + NT - 1) / NT;
    uint32_t *qover = qcount + 1;
    uint32_t *cl_n = qcount + 2;    // CRC-valid candidates of the tile staged in LDS
    uint32_t *cl_over = qcount + 3; // some were emitted directly: the staged list is incomplete
    const uint32_t qcap = (uint32_t)args.queue_cap;
    const uint32_t tile_rel = (uint32_t)(t0 - (int64_t)args.g_begin);

    // Normally ONE round: the survivors of the whole tile (~0.5 % of its offsets) fit the queue and all four waves
    // slice with dense lanes.  If they do not fit, the tile is redone in ranges of chunks [ch_lo, ch_hi) sized from
    // the count the failed round left in *qcount (a failed round costs its gate words only: ~3 us for a whole
    // tile); a single chunk that does not fit goes bit position by bit position (offset within the run): <= 256
    // entries per round, which cannot overflow (queue_cap >= 256).  The CRC-valid candidates of every round are
    // staged in the same list, which the filter behind the loop sees whole.
    // chunks whose gate words are computed together.  (Round 6 tried a whole tile's seven chunks in one batch with ONE queue
    // reservation per thread: equal on the sparse capture, 1.4 % slower on BASELINE configs[2] -- profiles/r6_ab_runs.txt.)
    constexpr int kGateBatch = 4;
    int ch_lo = 0, ch_hi = nchunks, grp = -1, width = nchunks;
    uint32_t try_fill = 0; // (kStats) try words in the tile's region so far: workgroup-uniform
    const bool stage_cands = !args.all_candidates;
    for (;;) {
        if (tid == 0) {
            *qcount = 0;
            *qover = 0;
        }
        __syncthreads();

#pragma unroll 1
        for (int base = ch_lo; base < ch_hi; base += kGateBatch) {
            // Two steps per batch of chunks.  First every gate word: plane reads and word-wide logic with no
            // dependence between chunks, so the LDS reads of a whole batch are in flight together (one chunk at
            // a time this loop took 3.2 us of a 46 us tile, most of it LDS latency) ...
            uint32_t gt[kGateBatch], gb1[kGateBatch], gb4[kGateBatch];
            // workgroup-uniform: every run of the batch exists and is complete (all but a tile's last batch, and
            // the last tiles of a launch): no per-lane range logic at all
            const bool full = grp < 0 && base + kGateBatch <= ch_hi && kRun * NT * (base + kGateBatch) <= off_end;
            auto gate_word = [&](int u, auto is_full) {
                const int vq = (base + u) * NT + tid;
                const int nvalid = off_end - kRun * vq; // <= 0: the run does not exist (vq >= own included: off_end <= 28 own)
                const int v = (decltype(is_full)::value || nvalid > 0) ? vq : 0; // planes are only read where they exist
                const uint32_t e2w[2] = {pl_e2[v + 1], pl_e2[v + 2]};
                const uint32_t dw[4] = {pl_d[v + 2], pl_d[v + 3], pl_d[v + 4], pl_d[v + 5]};
                // byte 0, bits 0..4 sit 80, 90, .., 120 samples after g (demod.c:109,46-81).  Bits 28..31 of
                // these words are whatever the funnel shift left there: the AND with E1 (28 valid bits) clears them.
                const uint32_t b0 = take28u<80 - 56>(dw), b1 = take28u<90 - 56>(dw), b2 = take28u<100 - 56>(dw),
                               b3 = take28u<110 - 56>(dw), b4 = take28u<120 - 56>(dw);
                // v_bitop3_b32: any function of three words in one instruction (the truth table is the immediate)
                const uint32_t hi = ADSB_BITOP3(b0, b1, b2, A & ~B & ~C);               // 10xxx
                const uint32_t lo = ADSB_BITOP3(b3, b4, df18_mask, (A ^ B) & (B | C));  // xxx01 (DF17, demod.c:64-67); xxx10 (DF18, :57-62) with -a
                const uint32_t h11 = ADSB_BITOP3(b0, b1, b2, ~A & B & ~C);              // 010xx
                const uint32_t m11 = ADSB_BITOP3(h11, b3, b4, A & B & C);               // 01011 (demod.c:70-77)
                const uint32_t df = ADSB_BITOP3(hi, lo, m11, (A & B) | C);
                // preamble: p1 > 2 s1 at g, p2 > 2 s2 <=> E2 at g + 30
                uint32_t gate = ADSB_BITOP3(pl_e1[v], take28u<30 - 28>(e2w), df, A & B & C);
                if constexpr (!decltype(is_full)::value) {
                    gate &= nvalid >= kRun ? ~0u : nvalid > 0 ? (1u << nvalid) - 1u : 0u;
                    if (base + u >= ch_hi)
                        gate = 0;
                    if (grp >= 0)
                        gate &= 1u << grp;
                }
                gt[u] = gate, gb1[u] = b1, gb4[u] = b4; // of a passing offset: b1 set <=> DF11; else b4 set <=> DF17
            };
            if (full) {
#pragma unroll
                for (int u = 0; u < kGateBatch; u++)
                    gate_word(u, std::true_type{});
            } else {
#pragma unroll
                for (int u = 0; u < kGateBatch; u++)
                    gate_word(u, std::false_type{});
            }
            // ... then the survivors (13 % of the lanes have one) go to the queue
#pragma unroll
            for (int u = 0; u < kGateBatch; u++) {
                uint32_t gate = gt[u];
                const int n = __popc(gate);
                if (n) {
                    const int v = (base + u) * NT + tid;
                    uint32_t slot = atomicAdd(qcount, (uint32_t)n);
                    if (slot + n <= qcap) {
                        while (gate) {
                            const int j = __ffs(gate) - 1;
                            gate &= gate - 1;
                            const uint32_t code = ((gb1[u] >> j) & 1u) ? 0u : ((gb4[u] >> j) & 1u) ? 1u : 2u;
                            queue[slot++] = ((uint32_t)v << 7) | ((uint32_t)j << 2) | code;
                        }
                    } else {
                        *qover = 1;
                    }
                }
            }
        }
        __syncthreads();
        ADSB_STAMP(2); // gate words + queue
        const bool over = *qover != 0;
        const int qtotal = (int)*qcount; // every survivor of the range, queued or not
        ADSB_COUNT(8, 1);
        ADSB_COUNT(9, over ? 0 : qtotal);
        const int qn = over ? 0 : qtotal;
        // valid.c:46,68: every DF-gate pass that is visited is a Try -- the queue entries ARE the tries.  The tile puts them
        // into its OWN region of args.tries (kTryRegion words, filled round by round; the count goes to args.try_counts[tile]
        // behind the rounds): no launch-wide reservation -- one more device-scope atomic per tile, awaited by the wave that
        // issued it at its next load, cost 22 % of the kernel (0.174 against 0.142 ms).  Only what does not fit the region
        // (a tile with more than 4 096 DF-gate passes: beyond 8 % of its offsets) and launches without regions (per-shard scans
        // hand a dense list to the host) reserve a range of the launch-wide list behind the regions; that round trip runs
        // under the slicer.  (Until round 6 a region held the whole-tile round only and every round behind a queue overflow
        // went through the list: on the adversarial capture 10 M words per launch, which the count pass then looked up one
        // binary search at a time -- 0.48 ms beside a 0.31 ms scan.)
        const bool try_region = kStats && args.try_counts && try_fill + (uint32_t)qn <= (uint32_t)kTryRegion; // workgroup-uniform
        uint32_t try_res = 0;
        if (kStats && tid == 0 && qn && !try_region) {
            if (args.hand)
                atomicOr(tile_over, 2u); // the launch-wide try list is in use: the count pass needs the launch's counters (kMarkTries)
            try_res = atomicAdd(&args.counters[1 * kCounterPad], (uint32_t)qn);
        }
#pragma unroll 1
        for (int q = tid; q < qn; q += NT) {
            const uint32_t ent = queue[q];
            const int sv = (int)(ent >> 7), sj = (int)((ent >> 2) & 31u);
            const uint32_t code = ent & 3u;
            const uint32_t g_rel = (uint32_t)(t0 - (int64_t)args.g_begin) + (uint32_t)(kRun * sv + sj);
            // Frame bit k = 14 b + c lies 80 + 10 k samples after g: the 14 column bytes come out of the D plane as word-wide
            // logic (slicer_bits.h: computed masks, nibble merges, one funnel shift -- ~165 instructions and 41 LDS reads per entry
            // where picking the 112 bits one by one took ~310 and 112; the same function runs on the CPU in
            // tests/cpp/slicer_bits.cpp against the definition).
            uint32_t cw[4];
            gather_columns(pl_d + sv, sj, cw);
            // short frames are bits 0..55 = rows b < 4; their syndromes are the long frame's 56 bits (4 rows) further
            // on: the low nibble of every column byte, in the high nibble's place
            uint32_t syn = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t iw = (code == 0) ? ((cw[j] & 0x0F0F0F0Fu) << 4) : cw[j];
#pragma unroll
                for (int k = 0; k < 4 && 4 * j + k < 14; k++)
                    syn ^= args.synd[(4 * j + k) * 256 + ((iw >> (8 * k)) & 0xFFu)];
            }
            uint32_t fixed = 0;
            if (syn != 0) {
                // valid.c:51,73: the reference rejects.  EXTENSION (off unless
                // cfg.fix_1bit): a long frame whose residual is the syndrome of ONE bit
                // k in [5,112) is repaired by flipping that bit in its column byte.
                if (!args.fix_tab || code == 0)
                    continue;
                const uint32_t e = args.fix_tab[(syn * args.fix_mul) >> 23];
                if ((e >> 8) != syn)
                    continue;
                const uint32_t k = e & 0xFFu, kc = k % 14u, kb = k / 14u;
#pragma unroll
                for (int wq = 0; wq < 4; wq++)
                    cw[wq] ^= ((kc >> 2) == (uint32_t)wq) ? (1u << (8 * (kc & 3u) + kb)) : 0u;
                fixed = 1;
            }

            // CRC-valid (about 1e-4 of the offsets).  Normally only staged here --
            // {g_rel, code, columns} -- and finished below with dense lanes.
            if (stage_cands) {
                const uint32_t ci = atomicAdd(cl_n, 1u);
                if (ci < (uint32_t)clist_cap) {
                    uint32_t *rec = cl_rec + ci * kCandWords;
                    rec[0] = g_rel;
                    rec[1] = code | (fixed << 8);
                    rec[2] = cw[0];
                    rec[3] = cw[1];
                    rec[4] = cw[2];
                    rec[5] = cw[3];
                    continue;
                }
                *cl_over = 1; // list full: this one is finished and emitted right here
            }
            uint32_t wds[4];
            columns_to_bytes(cw, code == 0, wds);
            wds[3] |= fixed << 24;
            const uint32_t pw = pw_at(xin, pbuf0, p_lo, p_hi, t0 + (int64_t)kRun * sv + sj);
            emit_loose(g_rel, pw, wds);
        }

        if (kStats && qn) { // (qn is workgroup-uniform)
            uint32_t tb = 0;
            if (!try_region) { // (workgroup-uniform) the range of the launch-wide list that thread 0 reserved
                if (tid == 0)
                    *try_base = try_res;
                __syncthreads(); // (the slicer only reads the queue)
                tb = *try_base;
            }
            uint32_t *dst = args.tries + (try_region ? (size_t)tile * kTryRegion + try_fill : (size_t)args.try_list_first + tb);
            const uint32_t room = try_region ? (uint32_t)qn : (tb < args.try_cap ? args.try_cap - tb : 0u);
            for (int q = tid; q < qn; q += NT) { // adjacent lanes, adjacent words
                const uint32_t ent = queue[q];
                if ((uint32_t)q < room)
                    dst[q] = ((tile_rel + (uint32_t)kRun * (ent >> 7) + ((ent >> 2) & 31u)) << 2) | (ent & 3u);
            }
            if (try_region)
                try_fill += (uint32_t)qn;
        }
        ADSB_STAMP(3); // slicer + CRC (+ try words)
        // next round (all of this is workgroup-uniform)
        if (grp >= 0) {
            if (++grp == kRun) { // chunk ch_lo is done bit by bit: on with the rest
                grp = -1;
                ch_lo = ch_hi;
                ch_hi = min(ch_lo + width, nchunks);
            }
        } else if (over) {
            // the same range again, cut to what the count says fits (and at least one chunk shorter)
            const int w = ch_hi - ch_lo;
            width = max(1, min(w - 1, (int)((uint32_t)w * qcap / (uint32_t)qtotal)));
            if (w == 1)
                grp = 0;
            else
                ch_hi = ch_lo + width;
        } else {
            if (2 * qn <= (int)qcap)
                width = min(2 * width, nchunks);
            ch_lo = ch_hi;
            ch_hi = min(ch_lo + width, nchunks);
        }
        if (ch_lo >= nchunks)
            break;
        __syncthreads(); // queue is rewritten
    }
    if (kStats && args.try_counts && tid == 0)
        args.try_counts[tile] = try_fill;
    if (stage_cands) {
        // Drop candidates the greedy scan (demod.c:89,128,134,141) can never visit.
        // Let c' be the closest candidate before c, with c inside c' (c.g < c'.g +
        // span').  The scan reaches c only by landing in (c'.g, c.g]: it cannot
        // walk there (it would visit c' first and jump past c), so some candidate
        // frame must END in (c'.g, c.g].  If the tile knows every candidate that
        // could (they start at >= c'.g - 1199, i.e. inside this tile, and the staged
        // list is complete), and none does, c is unreachable.  These are the +-1/2
        // sample shifted copies of every real frame: 3 of 4 records.
        __syncthreads();
        ADSB_STAMP(3);
        const int ncl = min((int)*cl_n, clist_cap); // <= kClistCap <= NT: one entry per thread
        ADSB_COUNT(10, ncl);
        const bool complete = *cl_over == 0;
        uint32_t res_need = 0, res_base = 0; // (the tile's reservation in the hand-off stream: below, once its records are counted)
        const bool reserves = tid == 0 && args.hand;
        bool keep = false;
        uint32_t rank = 0; // kept entries with a smaller offset: the record's place behind the tile's marker
        const uint32_t *ri = cl_rec + tid * kCandWords;
        const bool one_wave = ncl <= 64; // workgroup-uniform; the usual case (a tile stages ~20 candidates)
        uint32_t *any_nb = queue + kQueueCap - 8; // (a word of the survivor queue's LDS, free behind the rounds) do entries that stay sit next to each other?
        if (one_wave) {
            // Every entry sits in a lane of wave 0 and the all-pairs comparisons run on lane broadcasts
            // (v_readlane: the loop index is wave-uniform) instead of dependent LDS reads: measured with
            // per-phase stamps, the LDS loops below took 3.2 us of a 50 us tile, this takes 0.3.
            if (tid < 64) {
                const bool has = tid < ncl;
                const int gi = has ? (int)(ri[0] - tile_rel) : 0x3fffffff; // tile-local offset
                const bool lng = has && (ri[1] & 0xFFu) != 0;
                // key = 2 g + (long frame): ordered like g, and the closest predecessor's span comes with its key
                const int key = 2 * gi + (lng ? 1 : 0), g2 = 2 * gi;
                const int ei = has ? gi + (lng ? 1200 : 640) : 0x7fffffff; // where the candidate's frame ends
                // pk = the largest key below this one's = the closest candidate before it (with its length);
                // emax = the latest frame END that is not beyond this candidate: some frame ends in (pg, gi] <=> emax > pg.
                // Four entries per round, every round's broadcasts and compares independent of each other (lanes
                // beyond ncl hold neutral values): the lane -> scalar -> vector round trips overlap.
                int pk = -1, emax = -1;
                for (int j = 0; j < ncl; j += 4) {
                    int kj[4], ej[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        kj[u] = __builtin_amdgcn_readlane(key, j + u);
                        ej[u] = __builtin_amdgcn_readlane(ei, j + u);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        kj[u] = kj[u] < g2 ? kj[u] : -1;
                        ej[u] = ej[u] <= gi ? ej[u] : -1;
                    }
                    pk = max(max(pk, kj[0]), max(max(kj[1], kj[2]), kj[3]));
                    emax = max(max(emax, ej[0]), max(max(ej[1], ej[2]), ej[3]));
                }
                const int pg = pk >> 1, pspan = (pk & 1) ? 1200 : 640; // pk == -1: pg == -1, nothing precedes
                const bool drop = complete && pg >= ADSB_DECOFFSET_K - 1 && gi < pg + pspan && !(emax > pg);
                keep = has && !drop;
                const unsigned long long kept = __ballot(keep);
                const int kk = keep ? key : 0x7fffffff; // the entries that stay, as keys; the others never count
                bool nb = false; // an entry that stays sits one offset below this one (same length class): a possible copy
                for (int j = 0; j < ncl; j += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int kj = __builtin_amdgcn_readlane(kk, j + u);
                        rank += (uint32_t)(kj < g2);
                        nb |= kj == key - 2;
                    }
                }
                const unsigned long long nbs = __ballot(keep && nb);
                if (tid == 0) {
                    *tile_n = (uint32_t)__popcll(kept);
                    *any_nb = nbs != 0 ? 1u : 0u;
                }
            }
        } else {
            // More than a wave of entries (a tile of 48 k offsets full of 112-bit frames back to back stages ~130: BASELINE
            // configs[2]): the same all-pairs rule over two arrays of keys and frame ends in LDS, read four to a broadcast load.  Round 6: up to 128 entries are served by TWO threads each
            // (half of the pairs per thread, the two maxima merged by an LDS atomic: all four waves work instead of two), and
            // "the largest key below mine" is one subtraction and one unsigned maximum per pair -- k - mine wraps to a huge
            // number exactly for the keys below mine, and among those the largest k gives the largest difference -- where a
            // compare, a select and a signed maximum stood; the ranking pass is split the same way.  (Round 5: 32 rounds of 22
            // instructions per entry and pass on two waves, 6.7 us of such a tile's life.)
            uint32_t *fk = queue, *fe = queue + kClistCap, *kkv = queue + 2 * kClistCap; // (the survivor queue's words: free behind the rounds)
            uint32_t *rk = const_cast<uint32_t *>(pl_e1), *re = rk + kClistCap; // ... and the E1 plane's (>= 512 words: K >= 2)
            const bool has = tid < ncl;
            const int gi = has ? (int)(ri[0] - tile_rel) : 0x3fffffff; // tile-local offset
            const bool lng = has && (ri[1] & 0xFFu) != 0;
            const int key = 2 * gi + (lng ? 1 : 0), g2 = 2 * gi;
            fk[tid] = (uint32_t)key;                                          // (NT == kClistCap: one slot per thread)
            fe[tid] = (uint32_t)(has ? gi + (lng ? 1200 : 640) : 0x7fffffff); // where the candidate's frame ends
            rk[tid] = 0;
            re[tid] = 0;
            __syncthreads();
            const int n4 = (ncl + 3) & ~3;
            // entry `en` of this thread's part of the pairs: [j0, j1)
            const int parts = ncl <= NT / 2 ? 2 : 1, slots = NT / parts;
            const int en = tid & (slots - 1), part = tid / slots;
            const int per = ((n4 / 4 + parts - 1) / parts) * 4, j0 = part * per, j1 = min(n4, j0 + per);
            const uint32_t en_key = fk[en], en_g2 = en_key & ~1u, en_g1 = (en_key >> 1) + 1u; // (garbage for en >= ncl: never used)
            if (en < ncl) {
                uint32_t bk = 0, be = 0; // max over the pairs of (k - 2 g) and (end - (g + 1)), unsigned: >= 2^31 <=> some k < 2 g / some end <= g
                for (int j = j0; j < j1; j += 4) {
                    const u32x4 k4 = *reinterpret_cast<const u32x4 *>(fk + j), e4 = *reinterpret_cast<const u32x4 *>(fe + j);
                    bk = max(max(bk, k4.x - en_g2), max(k4.y - en_g2, max(k4.z - en_g2, k4.w - en_g2)));
                    be = max(max(be, e4.x - en_g1), max(e4.y - en_g1, max(e4.z - en_g1, e4.w - en_g1)));
                }
                if (parts > 1) {
                    atomicMax(&rk[en], bk);
                    atomicMax(&re[en], be);
                } else {
                    rk[en] = bk;
                    re[en] = be;
                }
            }
            __syncthreads();
            const uint32_t bk = rk[tid], be = re[tid];
            const int pk = bk >= 0x80000000u ? (int)(bk + (uint32_t)g2) : -1;       // the largest key below this one's
            const int emax = be >= 0x80000000u ? (int)(be + (uint32_t)gi + 1u) : -1; // the latest frame end that is not beyond this candidate
            const int pg = pk >> 1, pspan = (pk & 1) ? 1200 : 640; // pk == -1: pg == -1, nothing precedes
            const bool drop = complete && pg >= ADSB_DECOFFSET_K - 1 && gi < pg + pspan && !(emax > pg);
            keep = has && !drop;
            kkv[tid] = keep ? (uint32_t)key : 0x7fffffffu; // the entries that stay, as keys; the others never count
            rk[tid] = 0;                                    // (read above by this thread alone: now the rank's accumulator)
            const unsigned long long kept = __ballot(keep);
            if ((tid & 63) == 0 && kept)
                atomicAdd(tile_n, (uint32_t)__popcll(kept));
            if (tid == 0)
                *any_nb = 1; // (more than a wave of entries: frames back to back, copies everywhere)
            __syncthreads();
            if (en < ncl && kkv[en] != 0x7fffffffu) { // rank = kept entries with a smaller key: this thread's part of them
                uint32_t cnt = 0;
                for (int j = j0; j < j1; j += 4) {
                    const u32x4 k4 = *reinterpret_cast<const u32x4 *>(kkv + j);
                    cnt += (uint32_t)(k4.x < en_g2) + (uint32_t)(k4.y < en_g2) + (uint32_t)(k4.z < en_g2) + (uint32_t)(k4.w < en_g2);
                }
                if (parts > 1)
                    atomicAdd(&rk[en], cnt);
                else
                    rk[en] = cnt;
            }
            __syncthreads();
            rank = rk[tid];
        }
        // ---- from the staged entries that stay to the tile's records in the stream.
        // One record per RUN OF COPIES.  A frame decodes at two or three neighbouring offsets (the half-sample shifts); an
        // isolated frame's extra copies went with the filter above, but where frames stand back to back every copy is
        // reachable -- which one the scan lands on is decided by where the previous frame ended (demod.c:125-141) -- and a full
        // channel would cost three records per frame (310 k per 256 Mi samples at BASELINE configs[2]'s density, which the host
        // then has to read, check and walk: 3 x the kernel's time).  Entries r-1 and r (in ascending offset) are LINKED when
        // r's offset is the next one and code, repair flag and the four column words are the same; a chain of links is cut
        // into records of up to three offsets: the first one's {g_rel, pw}, the others' pw in the second granule's spare words,
        // their number in flags bits 1..2 (scan_kernel_format.h).  The host expands them where the greedy rule needs them.
        uint32_t st[kCandWords];
        if (keep) {
#pragma unroll
            for (int k = 0; k < kCandWords; k++)
                st[k] = ri[k];
            st[1] &= 0xFFFFu; // (the multi-wave filter's "kept" bit never existed here, but a link compares the whole word)
        }
        __syncthreads(); // every thread has read what it needs of the staged list (the filter's loops included)
        ADSB_STAMP(4); // never-visited filter + ranking
        const uint32_t nk = *tile_n;       // entries that stay
        const bool links = *any_nb != 0;   // workgroup-uniform: some of them are neighbours -- runs of copies are possible
        uint32_t *pwbuf = queue, *ext = queue + kClistCap, *wl = queue + kQueueCap - 4; // every entry's pw; the copies' pw (two per record); leaders per wave
        // e[]: the entry this thread finishes.  No neighbours (the usual tile: the filter has dropped an isolated frame's
        // copies): every entry that stays is a record of its own, finished by the thread that holds it, two barriers and the
        // re-ordering saved (the link logic cost the sparse launch 1.5 %).  Else: the entries replace the list in ascending
        // offset, thread r takes entry r, links are found and counted.
        uint32_t e[kCandWords] = {0, 0, 0, 0, 0, 0};
        bool act = keep, lead = keep, c1 = false, c2 = false;
        uint32_t nrec = nk, r2 = rank;
        if (!links) {
#pragma unroll
            for (int k = 0; k < kCandWords; k++)
                e[k] = st[k];
        } else {
            if (keep) {
                uint32_t *o = cl_rec + rank * kCandWords;
#pragma unroll
                for (int k = 0; k < kCandWords; k++)
                    o[k] = st[k];
            }
            __syncthreads();
            act = (uint32_t)tid < nk;
            lead = false;
            // Is this entry the previous entry's frame, one offset on?  One comparison per entry; the chains of links are then bit
            // logic on the four waves' ballots.  (Until round 6 every thread walked its chain backwards through the list -- up to
            // five evaluations of 12 dependent LDS reads each: 2.3 us of a full-channel tile's life.)
            bool lk = false;
            if (act) {
                const uint32_t *my = cl_rec + tid * kCandWords;
#pragma unroll
                for (int k = 0; k < kCandWords; k++)
                    e[k] = my[k];
                if (tid > 0) {
                    const uint32_t *p = my - kCandWords;
                    lk = e[0] == p[0] + 1u && e[1] == p[1] && e[2] == p[2] && e[3] == p[3] && e[4] == p[4] && e[5] == p[5];
                }
            }
            uint32_t *lmw = queue + kQueueCap - 16; // (eight free words of the survivor queue's LDS: the link masks, a wave each)
            const unsigned long long lm = __ballot(lk);
            if ((tid & 63) == 0) {
                lmw[2 * (tid >> 6)] = (uint32_t)lm;
                lmw[2 * (tid >> 6) + 1] = (uint32_t)(lm >> 32);
            }
            __syncthreads();
            if (act) {
                auto mask_of = [&](int w) { return (unsigned long long)lmw[2 * w] | (unsigned long long)lmw[2 * w + 1] << 32; };
                auto linked = [&](uint32_t a) { return a < (uint32_t)NT && ((mask_of((int)(a >> 6)) >> (a & 63u)) & 1ull) != 0; }; // (bits of entries >= nk are 0)
                // links behind this entry = the run of set bits that ends at its own bit: leading zeros of the inverted mask,
                // shifted so that its bit is the top one (the zeros shifted in below bit 0 invert to ones: the count stops there)
                int w = tid >> 6;
                const uint32_t b = (uint32_t)tid & 63u;
                const unsigned long long inv0 = ~(mask_of(w) << (63u - b));
                uint32_t back = inv0 ? (uint32_t)__clzll((long long)inv0) : 64u;
                if (back == b + 1u) // the run reaches the wave's first entry: on into the waves before
                    while (w-- > 0) {
                        const unsigned long long inv = ~mask_of(w);
                        const uint32_t more = inv ? (uint32_t)__clzll((long long)inv) : 64u;
                        back += more;
                        if (more < 64u)
                            break;
                    }
                lead = back % 3u == 0;
                c1 = lead && linked((uint32_t)tid + 1);
                c2 = c1 && linked((uint32_t)tid + 2);
            }
            const unsigned long long leaders = __ballot(lead);
            if ((tid & 63) == 0)
                wl[tid >> 6] = (uint32_t)__popcll(leaders);
            __syncthreads();
            nrec = 0;
            r2 = (uint32_t)__popcll(leaders & ((1ull << (tid & 63)) - 1ull)); // this leader's place among the records
#pragma unroll
            for (int w = 0; w < kWaves; w++) {
                r2 += w < (tid >> 6) ? wl[w] : 0u;
                nrec += wl[w];
            }
        }
        // The tile reserves its range of the hand-off stream -- one marker granule plus two per record, in whole 64-byte
        // lines -- with one device-scope atomic whose answer takes ~2 us under the scan's traffic: the round trip runs beside
        // the finishing below.  (Until round 5 the tile reserved for every entry that stayed, BEFORE it knew its records:
        // with runs of copies that left two thirds of every range unwritten, and the host, which reads the stream
        // sequentially, lost its prefetcher at every tile: 1.0 ms per 2 800 tiles.)
        ADSB_STAMP(5); // ascending order, links, leaders
        ADSB_COUNT(11, nrec);
        ADSB_COUNT(12, nk);
        if (reserves) { // the result is not looked at before this thread's own record is finished
            res_need = stream_granules(nrec);
            res_base = atomicAdd(&args.counters[2 * kCounterPad], res_need);
        }
        // finish the entry: bytes in order, pw (demod.c:127,133) -- every offset has a pw of its own
        uint32_t fin[6] = {0, 0, 0, 0, 0, 0};
        if (act) {
            const uint32_t cw[4] = {e[2], e[3], e[4], e[5]};
            uint32_t wds[4];
            columns_to_bytes(cw, (e[1] & 0xFFu) == 0, wds);
            wds[3] |= ((e[1] >> 8) & 1u) << 24; // repaired-by-extension flag
            const uint32_t pw = pw_at(xin, pbuf0, p_lo, p_hi, (int64_t)args.g_begin + e[0]);
            fin[0] = e[0], fin[1] = pw, fin[2] = wds[0], fin[3] = wds[1], fin[4] = wds[2], fin[5] = wds[3];
            if (links)
                pwbuf[tid] = pw;
        }
        if (reserves) {
            *tile_base = res_base;
            *tile_fit = (res_base < args.hand_cap && res_need <= args.hand_cap - res_base) ? 1u : 0u;
            *tile_lines = res_need >> 2;
            *tile_res = 1;
        }
        __syncthreads(); // tile_base / tile_fit are in, every entry has been read, every pw is known
        ADSB_STAMP(6); // bytes in order, pw, the reservation's round trip
        const bool to_stream = args.hand && *tile_fit; // workgroup-uniform
        if (act) {
            if (!to_stream) {
                const uint32_t wds[4] = {fin[2], fin[3], fin[4], fin[5]};
                emit_loose(fin[0], fin[1], wds); // (one by one: records of the loose list never carry copies)
            } else if (lead) {
                fin[5] |= ((c1 ? 1u : 0u) + (c2 ? 1u : 0u)) << kRecCopiesShift;
                const uint32_t pw1 = c1 ? pwbuf[tid + 1] : 0u, pw2 = c2 ? pwbuf[tid + 2] : 0u;
                ext[2 * r2] = pw1;
                ext[2 * r2 + 1] = pw2;
                atomicXor(&tile_chk[0], fin[0] ^ fin[4]); // word-wise XOR of its two granules
                atomicXor(&tile_chk[1], fin[1] ^ fin[5]);
                atomicXor(&tile_chk[2], fin[2] ^ pw1);
                atomicXor(&tile_chk[3], fin[3] ^ pw2);
                atomicAdd(tile_sum, record_term(r2, fin[0], fin[1]));
                // the finished record goes back into the list, in its place among the records (the list was last READ in
                // front of the barrier above: by its holders, or by the threads that took the entries in ascending offset)
                uint32_t *o = cl_rec + r2 * kCandWords;
#pragma unroll
                for (int k = 0; k < 6; k++)
                    o[k] = fin[k];
            }
        }
        if (to_stream) {
            if (tid == 0)
                *tile_n = nrec; // (the marker's count)
            // the tile's range {marker, records} leaves as one store of adjacent lanes
            __syncthreads();
            for (uint32_t L = tid; L < 1u + 2u * nrec; L += NT) {
                u32x4 gv;
                if (L == 0) {
                    const uint32_t nf = nrec | ((*tile_over & 1u) ? kMarkOver : 0u) | ((*tile_over & 2u) ? kMarkTries : 0u) | (*tile_lines << kMarkLinesShift);
                    uint32_t lo, hi;
                    marker_check(tile, nf, args.gen, tile_chk[0], tile_chk[1], tile_chk[2], tile_chk[3], *tile_sum, lo, hi);
                    gv = u32x4{tile, nf, lo, hi};
                    *tile_res = 2; // marker written
                } else {
                    const uint32_t ri2 = (L - 1u) >> 1;
                    const uint32_t *r = cl_rec + ri2 * kCandWords;
                    gv = ((L - 1u) & 1u) ? u32x4{r[4], r[5], ext[2 * ri2], ext[2 * ri2 + 1]} : u32x4{r[0], r[1], r[2], r[3]};
                }
                store_granule_through(args.hand, *tile_base + L, gv);
            }
        }
    }


    ADSB_STAMP(7); // records back into the list, check words, the store
    if (args.hand) {
        // Publish the tile: its marker granule {tile, count | flags, checksum} in front of
        // its records.  No fence: a system-scope release in every thread writes back the
        // L2 per tile (measured: 4.5x slower kernel), and without one nothing orders these
        // stores on their way to host memory (measured: a flag does overtake the records)
        // -- which is why the marker carries a checksum of the records (scan_kernel.h).
        __syncthreads();
        if (tid == 0 && *tile_res != 2) {
            uint32_t b = *tile_base, fit = *tile_fit;
            uint32_t lines = *tile_lines;
            if (!*tile_res) { // nothing was staged (all_candidates)
                b = atomicAdd(&args.counters[2 * kCounterPad], stream_granules(0));
                fit = b < args.hand_cap;
                lines = stream_granules(0) >> 2;
            }
            if (b < args.hand_cap) {
                const uint32_t nf = *tile_n | ((*tile_over & 1u) ? kMarkOver : 0u) | ((*tile_over & 2u) ? kMarkTries : 0u) | (fit ? 0u : kMarkNoFit) |
                                    (lines << kMarkLinesShift);
                uint32_t lo, hi;
                marker_check(tile, nf, args.gen, tile_chk[0], tile_chk[1], tile_chk[2], tile_chk[3], *tile_sum, lo, hi);
                store_granule_through(args.hand, b, u32x4{tile, nf, lo, hi});
            }
        }
    }

}

// The classic kernel: one workgroup of four waves per tile; Stage A, a barrier, Stage B between barriers.
template <bool kStats>
__global__ __launch_bounds__(kThreads, kMinWaves) void scan_kernel(const ScanArgs args)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int K = tile_passes(blockIdx.x, args.big_tiles, args.passes);
    const int nplane = kPassRuns * K + kPlanePad;
    uint32_t *pl_d = smem;
    uint32_t *pl_e1 = smem + nplane;
    uint32_t *pl_e2 = smem + 2 * nplane;
    uint32_t *queue = smem + 3 * nplane;
    uint32_t *qcount = queue + kQueueCap;
    uint32_t *cl_rec = qcount + 16; // kClistCap records of kCandWords

    const int tid = threadIdx.x;
    // The kMinWaves workgroups that start together on a CU at the head of a large launch (blocks b, b + 256, b + 512,
    // ... with the observed round-robin placement; nothing depends on it) begin 0, 1, 2, ... x kSleepStagger x 64 cycles
    // apart (2.6 us steps), so that their load phases do not coincide from the first pass on.  Measured in bench.py
    // with four per CU: -3.2 .. -4.0 % kernel time on one MI355X box, +-0.5 % on another; steps of 1.2 us did nothing,
    // steps of 3.7 us and more were worse than 2.6.  With five per CU (96 VGPRs since round 4) and only the first four
    // staggered the step was 3 % slower than with all five (profiles/r4_ab_runs.txt section 7).
    if (gridDim.x >= 256u * kMinWaves * 4 / kWaves && blockIdx.x < 256u * kMinWaves * 4 / kWaves) {
        const uint32_t slot = blockIdx.x >> 8;
        for (uint32_t i = 0; i < slot; i++)
            __builtin_amdgcn_s_sleep(kSleepStagger);
    }
    // cfg.profile: the launch's duration is (latest tile end) - (earliest tile start)
    const uint64_t prof_begin = args.profile ? __builtin_amdgcn_s_memrealtime() : 0;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t t0 = // first owned offset
        (int64_t)args.g_begin + (int64_t)kRun * (int64_t)tile_first_run(blockIdx.x, args.big_tiles, args.passes);

    // plane words past the last computed run are read (never used) by Stage B
    if (tid < kPlanePad) {
        pl_d[kPassRuns * K + tid] = 0;
        pl_e1[kPassRuns * K + tid] = 0;
        pl_e2[kPassRuns * K + tid] = 0;
    }

    // kernel arguments are only ever used by value (taking their address would
    // demote the sample pointer to a flat/scratch access)
    ADSB_STAMP_BEGIN();
    stage_a(args.x, args.pbuf0, args.p_lo, args.p_hi, t0, K, wave, lane, pl_d, pl_e1, pl_e2);
    __syncthreads();
    ADSB_STAMP(1); // Stage A
    ADSB_COUNT(0, 1);
    stage_b<kStats>(args, blockIdx.x, K, t0, tid, pl_d, pl_e1, pl_e2, queue, qcount, cl_rec, args.clist_cap, stamp_last);
    ADSB_STAMP_END(13); // the marker

    if (args.profile) { // the launch's duration is (latest tile end) - (earliest tile start)
        __syncthreads();
        if (tid == 64) { // not wave 0: that one has just issued the tile's write-through stores
            unsigned long long *c64 = reinterpret_cast<unsigned long long *>(args.counters);
            atomicMax(&c64[2 * kCounterPad], ~(unsigned long long)prof_begin); // = counters 4 and 5
            atomicMax(&c64[(5 * kCounterPad) / 2], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        }
    }
}

// Behind every scan launch, one wave: hand the launch counters to the host (two
// write-through granules into pinned memory) and leave them zero for the slot's next
// launch.  This replaces a runtime copy kernel, a fill kernel and the gap between them
// (measured: 16 us per launch of a multi-launch stream, against ~4 us).  Letting the last
// tile to finish do it instead -- a `done` counter -- was measured too: every tile then waits
// for its atomics to return before it can retire, and the kernel ran 25 % slower.
__global__ __launch_bounds__(64) void report_kernel(uint32_t *counters, uint32_t *report, uint32_t gen)
{
    if (threadIdx.x != 0)
        return;
    unsigned long long *c64 = reinterpret_cast<unsigned long long *>(counters);
    const uint32_t c0 = atomicExch(&counters[0], 0u), c1 = atomicExch(&counters[1 * kCounterPad], 0u),
                   c2 = atomicExch(&counters[2 * kCounterPad], 0u);
    atomicExch(&counters[3 * kCounterPad], 0u);
    const unsigned long long nb = atomicExch(&c64[2 * kCounterPad], 0ull), en = atomicExch(&c64[(5 * kCounterPad) / 2], 0ull);
    store_granule_through(report, 0, u32x4{c0, c1, c2, gen});
    store_granule_through(report, 1, u32x4{(uint32_t)nb, (uint32_t)(nb >> 32), (uint32_t)en, (uint32_t)(en >> 32)});
}

// Is g inside an accepted frame that starts before it?  (The frame's own offset is a visited Try: strict
// inequality.)  Binary search over the whole frame array: the list and carry entries, and tiles whose window
// of frames does not fit a wave.
__device__ __forceinline__ bool try_shadowed(const TryCountArgs &a, uint64_t g)
{
    uint32_t lo = 0, hi = a.n_frames;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.frames[mid].g < g)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo > 0 && g < a.frames[lo - 1].g + a.frames[lo - 1].span;
}

// Blocks [0, ceil(n_tiles / 4)): one WAVE per tile region.  The accepted frames that can shadow a try of the
// tile start inside (t0 - 1200, t0 + tile offsets): the wave finds the first of them with a 64-way search (three
// rounds of one load per lane for 13 k frames, instead of 14 dependent loads per try), keeps the window one
// frame per lane, and tests every try against it on lane broadcasts.  Measured: the per-try binary search took
// 34 us per 256 Mi samples with 8 192 resident waves, and slowed the scan kernel running beside it by as much.
// The remaining blocks: the launch-wide list and the carry of earlier passes, grid-stride.
constexpr int kCountThreads = 256;
// Blocks that walk the tile regions, four tiles at a time each (grid-stride).  The pass runs beside the NEXT launch's scan, which
// is bound by VALU issue: with one block per four tiles (697 for a 256 Mi-sample launch) that scan took 0.156 ms, with 256 or 64
// blocks 0.149 ms and the statistics step 2 % less; with 16 the pass itself (0.33 ms) became the step, and with 64 a stream of
// eight launches (2 Gi samples) ended 0.3 ms later, its passes queued up behind the scans (profiles/r4_ab_runs.txt section 8).
constexpr int kCountRegionGrid = 256;
// (32 VGPRs: what five resident scan waves of 96 leave free on a SIMD of 512.  Round 6's first version of the four-words-per-lane
// loop took 40 and no longer fitted beside the scan it runs next to: the pass took 72 us instead of 53 on the sparse capture and
// the scan beside it 3 % longer -- profiles/r6_ab_runs.txt section 5.)
__global__ __launch_bounds__(kCountThreads) void count_tries_kernel(const TryCountArgs a)
{
    uint32_t cnt[3] = {0, 0, 0};
    const uint32_t region_blocks = a.regions ? (a.n_tiles + 3u) / 4u : 0u;
    auto carry = [&](uint64_t g, uint32_t code) { // the scan has not got there yet
        if (a.final)
            return;
        const uint32_t slot = atomicAdd(a.n_carry_out, 1u);
        if (slot < a.carry_cap)
            a.carry_out[slot] = (g << 2) | code;
        else
            a.acc[3] = 1; // reported when the statistics are read
    };
    const uint32_t region_grid = min(region_blocks, (uint32_t)kCountRegionGrid);
    if (blockIdx.x < region_grid) {
      for (uint32_t rb = blockIdx.x; rb < region_blocks; rb += region_grid) {
        const uint32_t tile = rb * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
        const uint32_t n = tile < a.n_tiles ? min(a.region_counts[tile], (uint32_t)kTryRegion) : 0u;
        if (n) { // wave-uniform
            const uint64_t t0 = a.g_base + (uint64_t)kRun * tile_first_run(tile, a.big_tiles, a.passes);
            const uint64_t t1 = t0 + (uint64_t)kRun * (uint64_t)owned_runs(tile_passes(tile, a.big_tiles, a.passes));
            const uint64_t key = t0 >= (uint64_t)(ADSB_DECOFFSET_K - 1) ? t0 - (ADSB_DECOFFSET_K - 1) : 0; // frames below cannot reach t0
            // first frame with g >= key: the answer lies in [lo, hi]
            uint32_t lo = 0, hi = a.n_frames;
            while (lo < hi) {
                const uint32_t step = (hi - lo + 63u) / 64u, idx = lo + lane * step;
                const bool below = idx < hi && a.frames[idx].g < key; // true for a prefix of the lanes
                const uint32_t k = (uint32_t)__popcll(__ballot(below));
                if (k == 0) {
                    hi = lo;
                } else {
                    const uint32_t nlo = lo + (k - 1u) * step + 1u;
                    hi = min(lo + k * step, hi);
                    lo = nlo;
                }
            }
            // the window, one frame per lane, relative to key (everything fits 32 bits)
            const uint32_t fi = lo + lane;
            const bool have = fi < a.n_frames && a.frames[fi].g < t1;
            const uint32_t fr = have ? (uint32_t)(a.frames[fi].g - key) : 0xFFFFFFFFu;
            const uint32_t fe = have ? fr + a.frames[fi].span : 0u;
            const uint32_t w = (uint32_t)__popcll(__ballot(have));
            const bool fits = w < 64u || lo + 64u >= a.n_frames || a.frames[lo + 64u].g >= t1; // wave-uniform
            const uint32_t *reg = a.regions + (size_t)tile * kTryRegion;
            const uint32_t base_minus_key = (uint32_t)(a.g_base - key); // (mod 2^32: key may lie up to 1199 offsets below the launch's base)
            const uint32_t hi_rel = a.hi <= a.g_base ? 0u : (uint32_t)std::min<uint64_t>(a.hi - a.g_base, 0xFFFFFFFFull); // tries at or beyond are carried
            // kW tries per lane and round: the loads are in flight together (a region of the adversarial capture holds 3 600
            // words: one load per round and lane was 57 dependent round trips per tile), and a broadcast of the window serves kW
            // comparisons.  kW = 2, not 4: the pass has to fit the 32 registers that five resident scan waves leave free on a SIMD
            // (with 4 it took 40, ran 72 us instead of 53 on the sparse capture and cost the scan beside it 3 %).
            constexpr int kW = 2;
            for (uint32_t i0 = 0; i0 < n; i0 += 64u * kW) {
                uint32_t word[kW];
                bool live[kW];
#pragma unroll
                for (int u = 0; u < kW; u++) {
                    const uint32_t i = i0 + 64u * (uint32_t)u + lane;
                    live[u] = i < n;
                    word[u] = live[u] ? reg[i] : 0u;
                }
                // (32-bit arithmetic relative to the launch's base and to the window's key: the pass has 32 registers)
                uint32_t rt[kW];
                bool shadowed[kW] = {};
#pragma unroll
                for (int u = 0; u < kW; u++) {
                    const uint32_t gr = word[u] >> 2;
                    rt[u] = gr + base_minus_key; // (only looked at for tries below a.hi: those lie inside the tile)
                    if (live[u] && gr >= hi_rel) {
                        carry(a.g_base + gr, word[u] & 3u);
                        live[u] = false;
                    }
                }
                if (fits) {
                    for (uint32_t j = 0; j < w; j++) { // (w is wave-uniform)
                        const uint32_t fj = __builtin_amdgcn_readlane(fr, j), ej = __builtin_amdgcn_readlane(fe, j);
#pragma unroll
                        for (int u = 0; u < kW; u++)
                            shadowed[u] |= fj < rt[u] && rt[u] < ej;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kW; u++)
                        shadowed[u] = live[u] && try_shadowed(a, a.g_base + (word[u] >> 2));
                }
#pragma unroll
                for (int u = 0; u < kW; u++)
                    if (live[u] && !shadowed[u]) {
                        const uint32_t code = word[u] & 3u;
                        cnt[code < 3 ? code : 2]++;
                    }
            }
        }
      }
    } else {
        const uint32_t n_carry = min(*a.n_carry, a.carry_cap);
        if (blockIdx.x == region_grid && threadIdx.x == 0)
            *a.n_carry_next = 0; // three counts in rotation: nobody reads or appends to this one during this pass
        const uint32_t total = a.n_tries + n_carry;
        const uint32_t nb = gridDim.x - region_grid;
        for (uint32_t i = (blockIdx.x - region_grid) * blockDim.x + threadIdx.x; i < total; i += nb * blockDim.x) {
            uint64_t g;
            uint32_t code;
            if (i < a.n_tries) {
                const uint32_t w = a.tries[i];
                g = a.g_base + (w >> 2);
                code = w & 3u;
            } else {
                const uint64_t w = a.carry_in[i - a.n_tries];
                g = w >> 2;
                code = (uint32_t)w & 3u;
            }
            if (g >= a.hi) {
                carry(g, code);
                continue;
            }
            if (!try_shadowed(a, g))
                cnt[code < 3 ? code : 2]++;
        }
    }
    // one atomic per block and counter: same-address atomics serialise in L2
    __shared__ uint32_t part[3];
    if (threadIdx.x < 3)
        part[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t v = cnt[k];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off);
        if ((threadIdx.x & 63) == 0 && v)
            atomicAdd(&part[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 3 && part[threadIdx.x])
        atomicAdd(&a.acc[threadIdx.x], (unsigned long long)part[threadIdx.x]);
}

hipError_t launch_count_tries(const TryCountArgs &args, hipStream_t stream)
{
    // the carry count lives on the device: a fixed number of blocks, grid-stride over whatever there is
    const uint32_t region_blocks = args.regions ? (args.n_tiles + 3u) / 4u : 0u;
    const uint32_t guess = args.n_tries + 65536u;
    // (beside the tile regions the list is what did not fit them -- rare, short -- and the carry: 64 blocks, so that the pass stays
    // small beside the next scan; a LONG list -- a launch forced onto it, tiles beyond 4 096 passes -- is a latency-bound chain
    // of look-ups per thread and needs the threads: one block per 2 048 words, up to 1 024)
    const uint32_t cap = args.regions ? std::min<uint32_t>(1024u, std::max<uint32_t>(64u, args.n_tries / 2048u)) : 2048u;
    const unsigned list_blocks = (unsigned)std::min<uint32_t>((guess + kCountThreads - 1u) / kCountThreads, cap);
    const uint32_t region_grid = std::min<uint32_t>(region_blocks, (uint32_t)kCountRegionGrid);
    hipLaunchKernelGGL(count_tries_kernel, dim3(region_grid + list_blocks), dim3(kCountThreads), 0, stream, args);
    return hipGetLastError();
}

void make_syndrome_table(uint32_t *out)
{
    // S[k] = x^(111-k) mod G, G = x^24 + 0xFFF409 (crc.h): the residual of
    // valid.c:49-51,71-73 is the XOR of S[k] over the set frame bits k.
    uint32_t s[112];
    uint32_t r = 1;
    for (int e = 0; e < 112; e++) {
        s[111 - e] = r;
        r <<= 1;
        if (r & 0x1000000u)
            r ^= 0x1FFF409u;
    }
    for (int c = 0; c < 14; c++)
        for (int v = 0; v < 256; v++) {
            uint32_t acc = 0;
            for (int b = 0; b < 8; b++)
                if (v & (1 << b))
                    acc ^= s[14 * b + c];
            out[c * 256 + v] = acc;
        }
}

uint32_t make_fix_table(uint32_t *tab)
{
    uint32_t syn[112];
    uint32_t r = 1;
    for (int e = 0; e < 112; e++) {
        syn[111 - e] = r;
        r <<= 1;
        if (r & 0x1000000u)
            r ^= 0x1FFF409u;
    }
    for (uint32_t mul = 0x9E3779B1u;; mul += 2) {
        for (int i = 0; i < kFixSlots; i++)
            tab[i] = 0;
        bool ok = true;
        for (int k = 5; k < 112 && ok; k++) {
            uint32_t &slot = tab[(uint32_t)(syn[k] * mul) >> 23];
            ok = slot == 0;
            slot = (syn[k] << 8) | (uint32_t)k;
        }
        if (ok)
            return mul;
    }
}

int choose_passes(uint64_t n_offsets, int cus, bool dense)
{
    if (cus <= 0)
        cus = 256;
    // Measured, not modelled (round 6: launches of every size interleaved in one process, K = 2..7 side by side on the
    // kernel's own clock: tools/ab_interleaved.py, profiles/r6_ab_passes_sizes.txt; rounds 1-5 chose by a cost model of
    // "rounds of resident workgroups" that the same measurement refutes -- a large launch's time is 10.7 us + 46 ns per tile,
    // no steps -- and that was up to 14 % off the best K at 16 Mi offsets).  Below ~64 Mi offsets the best K is not monotonic:
    // the launch is one or two device-fulls of tiles and the fill of the last one decides.
    //   offsets (Mi)      1     2     4     8     16    24    32    48    64    80    88   128
    //   best K            2     2     3     3     5     5     3     3     4     6     6     7
    //   next best, +%   3:17  3:13  2:7   2:4   2:10  4:5   7:2   5:2   5:1   7:2   5:2   6:1
    // From K = 8 on a CU's LDS holds four workgroups, not five (+7 % and more).  At 128 Mi offsets, K = 4 / 5 / 6 / 7 / 8:
    // sparse 1.045 / 1.036 / 1.007 / 1 / 1.067, noise the same; BASELINE configs[2] at its stated density 0.970 / 0.981 /
    // 0.972 / 1 / 1.093, with the Try/Ok table 0.961 / 0.939 / 0.938 / 1: a tile of frames back to back stages fewer
    // candidates for its all-pairs filter and overflows its survivor queue less often.  `dense` = the handle's previous
    // launch handed over a record per 2 048 offsets or more.
    const uint64_t n = n_offsets * 256u / (uint64_t)cus; // (the table is a 256-CU device's)
    constexpr struct {
        uint32_t below_mi;
        int k;
    } table[] = {{3, 2}, {12, 3}, {28, 5}, {56, 3}, {72, 4}, {96, 6}};
    for (const auto &row : table)
        if (n < ((uint64_t)row.below_mi << 20))
            return row.k;
    return dense ? 6 : 7;
}

// The staging buffer's unscanned tail (a few KB) moved to the other buffer, by a kernel of the library's own: the
// runtime's device-to-device copy loads its blit kernels on first use -- 7 ms inside the first such hipMemcpyAsync of a
// process, a quarter of what the C host program needs to decode a 510 MiB capture (profiles/r5_cli_timing.txt).  src and dst are
// 16-byte aligned (the tail starts on a 128-byte line of one buffer and goes to the first sample of the other).
__global__ __launch_bounds__(256) void copy_samples_kernel(uint16_t *__restrict__ dst, const uint16_t *__restrict__ src, size_t n)
{
    const size_t n8 = n / 8, stride = (size_t)gridDim.x * blockDim.x;
    const u32x4 *s4 = reinterpret_cast<const u32x4 *>(src);
    u32x4 *d4 = reinterpret_cast<u32x4 *>(dst);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride)
        d4[i] = s4[i];
    if (blockIdx.x == 0 && threadIdx.x < n - 8 * n8)
        dst[8 * n8 + threadIdx.x] = src[8 * n8 + threadIdx.x];
}

hipError_t launch_copy_samples(uint16_t *dst, const uint16_t *src, size_t n, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) // (not the tail's case: let the runtime do it)
        return hipMemcpyAsync(dst, src, n * sizeof(uint16_t), hipMemcpyDeviceToDevice, stream);
    const unsigned blocks = (unsigned)std::min<size_t>(1024, (n / 8 + 255) / 256 + 1);
    hipLaunchKernelGGL(copy_samples_kernel, dim3(blocks), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs &args, bool stats, hipStream_t stream)
{
    if (args.g_end <= args.g_begin)
        return hipSuccess;
    const unsigned blocks = tile_count(args.g_end - args.g_begin, args.big_tiles, args.passes);
    const size_t lds = lds_bytes(args.passes);
    if (stats)
        hipLaunchKernelGGL(scan_kernel<true>, dim3(blocks), dim3(kThreads), lds, stream, args);
    else
        hipLaunchKernelGGL(scan_kernel<false>, dim3(blocks), dim3(kThreads), lds, stream, args);
    if (args.report)
        hipLaunchKernelGGL(report_kernel, dim3(1), dim3(64), 0, stream, args.counters, args.report, args.gen);
    return hipGetLastError();
}

} // namespace adsb
