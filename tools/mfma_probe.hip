// mfma_probe.hip -- can the matrix pipe serve as a second array of ROUNDED f32 multipliers?
//
// v_mfma_f32_4x4x1_16b_f32 computes, for 16 independent blocks, D[i][j] = fma(A[i], B[j], C[i][j])
// with K = 1: one fused multiply-add per element, no accumulation chain.  With C = 0 that is
// RN(A[i] * B[j]) -- the separately rounded product the reference's FIR needs (air.c:72-73) -- and
// each lane gets ITS OWN B value times four A values.  This probe checks (1) the operand layout and
// that the products are bit-identical to v_mul_f32, (2) how many cycles such an MFMA takes on one
// SIMD and (3) how much VALU issue it costs when both pipes are busy (same wave / different waves).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void layout_kernel(const float *a, const float *b, float *d, int cbsz, int abid)
{
    const int l = threadIdx.x;
    f32x4 r;
    const f32x4 z = {0, 0, 0, 0};
    if (cbsz == 0) r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], z, 0, 0, 0);
    else r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], z, 4, 0, 0); // A of block `abid`=0 broadcast to all 16
    (void)abid;
    d[4 * l + 0] = r.x; d[4 * l + 1] = r.y; d[4 * l + 2] = r.z; d[4 * l + 3] = r.w;
}

// MODE 0: MFMA only   1: pk_add only   2: per iteration 8 MFMA + 8*R pk_add interleaved (R = RATIO)
// 3: waves with odd blockIdx run MFMAs, even run pk_adds
template <int MODE, int RATIO>
__global__ __launch_bounds__(256) void rate_kernel(float *out, int iters, float seed)
{
    f32x4 acc[8];
    f32x2 p[16];
    float av = seed + (threadIdx.x & 3), bv = seed * 3.0f + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = f32x2{seed + i, seed - i + threadIdx.x};
    const f32x2 c2 = {seed, seed + 1.0f};
    const bool mfma_wave = (MODE == 0) || (MODE == 2) || (MODE == 3 && (blockIdx.x & 1));
    const bool valu_wave = (MODE == 1) || (MODE == 2) || (MODE == 3 && !(blockIdx.x & 1));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (mfma_wave)
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(acc[i]) : "v"(av), "v"(bv));
            if (valu_wave) {
#pragma unroll
                for (int r = 0; r < (MODE == 2 ? RATIO : 2); r++)
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i * 2 + r) & 15]) : "v"(c2));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 16; i++) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int RATIO>
int run(const char *name, float *out, double mfma_per_iter, double valu_per_iter)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = 256 * wps;
        rate_kernel<MODE, RATIO><<<blocks, 256>>>(out, 100, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); rate_kernel<MODE, RATIO><<<blocks, 256>>>(out, iters, 1.0f); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double share = MODE == 3 ? 0.5 : 1.0; // mode 3: half of the waves per role
        const double ns_iter = ms * 1e6 / iters / (wps * share);
        printf("%-34s %d waves/SIMD: %.3f ms; per SIMD per iteration-of-one-wave %.1f ns = %.1f cyc@2.4GHz (%.0f MFMA + %.0f pk_add)\n", name, wps, ms,
               ns_iter, ns_iter * 2.4, mfma_per_iter, valu_per_iter);
    }
    return 0;
}

int main()
{
    // ---- layout + exactness
    float ha[64], hb[64], hd[256];
    float *a, *b, *d;
    CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024));
    const double taps[7] = {0.012627, 0.037881, 0.063135, 0.088388, 0.075761, 0.050508, 0.025254};
    srand(1);
    int bad_layout = 0, bad_bits = 0, bad_bcast = 0;
    for (int rep = 0; rep < 2000; rep++) {
        for (int l = 0; l < 64; l++) {
            ha[l] = (float)taps[(l + rep) % 7] * (rep % 5 == 4 ? -1.0f : 1.0f);
            hb[l] = (float)((rand() % 65536) - 2048) * ((rep % 3 == 2 && l % 7 == 0) ? 0.0f : 1.0f);
        }
        CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
        for (int cb = 0; cb < 2; cb++) {
            layout_kernel<<<1, 64>>>(a, b, d, cb, 0);
            CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; l++)
                for (int i = 0; i < 4; i++) {
                    const int blk = cb ? 0 : l / 4;
                    volatile float want = ha[4 * blk + i] * hb[l]; // host IEEE binary32 multiply
                    float w = want, g = hd[4 * l + i];
                    if (memcmp(&w, &g, 4) != 0) {
                        if (w == g) bad_bits++; // e.g. sign of zero
                        else (cb ? bad_bcast : bad_layout)++;
                        if (bad_layout + bad_bcast + bad_bits < 6)
                            printf("  rep %d cbsz %d lane %d i %d: want %a got %a\n", rep, cb, l, i, w, g);
                    }
                }
        }
    }
    printf("layout/exactness over 2000 x 64 x 4 x 2 products: wrong value (no broadcast) %d, wrong value (cbsz=4 broadcast) %d, equal but different bits %d\n",
           bad_layout, bad_bcast, bad_bits);
    float *out; CK(hipMalloc(&out, 4));
    run<0, 0>("MFMA 4x4x1 only", out, 8, 0);
    run<1, 0>("pk_add only (16/iter)", out, 0, 16);
    run<2, 1>("same wave: 8 MFMA + 8 pk_add", out, 8, 8);
    run<2, 2>("same wave: 8 MFMA + 16 pk_add", out, 8, 16);
    run<2, 4>("same wave: 8 MFMA + 32 pk_add", out, 8, 32);
    run<3, 0>("split waves: MFMA | 16 pk_add", out, 8, 16);
    return 0;
}
