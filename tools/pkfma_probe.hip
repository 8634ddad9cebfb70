// pkfma_probe.hip -- does v_pk_fma_f32 with three distinct 64-bit sources issue at v_pk_mul_f32's rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    f32x2 p[16], q[16], c[4];
    for (int i = 0; i < 16; i++) { p[i] = f32x2{seed + i, seed - i + threadIdx.x}; q[i] = f32x2{seed * i, seed + threadIdx.x * i}; }
    for (int i = 0; i < 4; i++) c[i] = f32x2{seed + 3 * i, seed - 7 * i};
    f32x2 ks = {seed * 0.5f, seed * 0.25f}; // uniform -> SGPR pair
    ks.x = __builtin_amdgcn_readfirstlane(__float_as_int(ks.x)) ? ks.x : 1.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(q[i]), "s"(ks));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "s"(ks), "v"(q[i]), "v"(c[i & 3]));
                if (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "v"(q[(i + 5) & 15]), "v"(q[i]), "v"(c[i & 3]));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,1] neg_hi:[1,0,1]" : "=v"(p[i]) : "s"(ks), "v"(q[i]), "v"(c[i & 3]));
                if (OP == 4) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[i]) : "v"(q[i]), "v"(c[i & 3]));
            }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int OP> int run(const char *name, float *out)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, wps = 4, blocks = 256 * wps;
    k<OP><<<blocks, 256>>>(out, 100, 1.0f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(out, iters, 1.0f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s 4 waves/SIMD: %.2f cyc@2.4GHz per wave-instruction per SIMD\n", name, ms * 1e6 / ((double)iters * 64 * wps) * 2.4);
    return 0;
}
int main()
{
    float *out; CK(hipMalloc(&out, 4));
    run<0>("v_pk_mul_f32 v, v, s[pair]", out);
    run<4>("v_pk_add_f32 v, v, v", out);
    run<1>("v_pk_fma_f32 v, s[pair], v, v", out);
    run<3>("v_pk_fma_f32 v, s[pair], v, v (neg mods)", out);
    run<2>("v_pk_fma_f32 v, v, v, v (3 distinct)", out);
    return 0;
}
