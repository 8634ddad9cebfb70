// valu_bench.hip -- cycles per wave64 VALU instruction on one SIMD at 1..8 waves/SIMD
// (tuning aid: what do v_add_f32 / v_pk_add_f32 / v_pk_mul_f32 / v_alignbit / v_cvt cost?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    // 16 independent chains so that no instruction waits for its predecessor
    float a[16];
    f32x2 p[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        a[i] = seed + i + threadIdx.x;
        p[i] = f32x2{seed + i, seed - i + threadIdx.x};
        u[i] = (uint32_t)(i * 977 + threadIdx.x);
    }
    const float c = seed * 0.5f + 1.0f;
    const f32x2 c2 = {c, c + 1.0f};
    f32x2 c3 = {c + 2.0f, c + 3.0f};
    asm volatile("" : "+v"(c3));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (OP == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (OP == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
                if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 7) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i]));
                if (OP == 8) asm volatile("v_trunc_f32 %0, %0" : "+v"(a[i]));
                if (OP == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 15]) : "vcc");
                if (OP == 11) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : : "vcc");
                if (OP == 12) { // plane bit: compare + shift-in
                    asm volatile("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(u[i]) : "v"(a[i]), "v"(a[(i + 1) & 15]) : "vcc");
                }
                if (OP == 13) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 14) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
                if (OP == 15) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 16) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 18) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
                if (OP == 19) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
                if (OP == 20) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 21) asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 22) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 23) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 24) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 25) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 26) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 27) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(a[(i + 1) & 15]));
                if (OP == 28) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));
                if (OP == 29) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                if (OP == 30) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 15]) : "s20", "s21");
                if (OP == 40) { f32x2 t; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p[(i + 1) & 15]), "s"(c2)); asm volatile("" : : "v"(t)); }
                if (OP == 41) { // round 5's FIR step: product with a scalar tap pair, then the accumulation
                    f32x2 t;
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p[(i + 1) & 15]), "s"(c2));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t));
                }
                if (OP == 42) { // round 6's: the product is a fused multiply-add with a vector constant, modifiers on
                    f32x2 t;
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(p[(i + 1) & 15]), "s"(c2), "v"(c3));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t));
                }
                if (OP == 43) { f32x2 t; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(p[(i + 1) & 15]), "s"(c2), "v"(c3)); asm volatile("" : : "v"(t)); }
                if (OP == 44) { f32x2 t; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(p[(i + 1) & 15]), "s"(c2), "v"(c3)); asm volatile("" : : "v"(t)); }
                if (OP == 45) { f32x2 t; asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(p[(i + 1) & 15]), "v"(c2), "v"(c3)); asm volatile("" : : "v"(t)); }
                if (OP == 46) { f32x2 t; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p[(i + 1) & 15]), "v"(c2)); asm volatile("" : : "v"(t)); }
                if (OP == 47) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 5) & 15]));
                if (OP == 9) { // the FIR's dependent pair: pk_mul into a temp, pk_add accumulate (2 instr)
                    f32x2 t;
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p[(i + 1) & 15]), "v"(c2));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(t));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + p[i].x + p[i].y + (float)u[i];
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
int run(const char *name, float *out, int per_iter)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps : {1, 4, 5}) { // waves per SIMD: blocks of 256 threads = 1 wave on each of 4 SIMDs
        const int blocks = 256 * wps;
        k<OP><<<blocks, 256>>>(out, 100, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(out, iters, 1.0f); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * per_iter * wps;
        printf("%-22s %d waves/SIMD: %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    return 0;
}

int main()
{
    float *out; CK(hipMalloc(&out, 4));
    if (getenv("VB_FIR")) {
        run<46>("pk_mul v,v,v", out, 64);
        run<40>("pk_mul v,v,s", out, 64);
        run<47>("pk_add v,v,v", out, 64);
        run<44>("pk_fma v,v,s,v", out, 64);
        run<45>("pk_fma v,v,v,v", out, 64);
        run<43>("pk_fma v,v,s,v mods", out, 64);
        run<41>("pk_mul(s)+pk_add", out, 128);
        run<42>("pk_fma(s,v,mods)+pk_add", out, 128);
        return 0;
    }
    run<0>("v_add_f32", out, 64);
    run<6>("v_mul_f32", out, 64);
    run<4>("v_fma_f32", out, 64);
    run<1>("v_pk_add_f32", out, 64);
    run<2>("v_pk_mul_f32", out, 64);
    run<5>("v_pk_fma_f32", out, 64);
    run<3>("v_alignbit_b32", out, 64);
    run<7>("v_cvt_f32_u32_sdwa", out, 64);
    run<8>("v_trunc_f32", out, 64);
    run<9>("pk_mul+pk_add pair", out, 128);
    run<10>("v_cmp_gt_f32 vcc", out, 64);
    run<30>("v_cmp_gt_f32_e64 sgpr", out, 64);
    run<11>("v_addc_co_u32", out, 64);
    run<12>("cmp+addc pair", out, 128);
    run<13>("v_sub_f32", out, 64);
    run<24>("v_max_f32", out, 64);
    run<25>("v_fmac_f32", out, 64);
    run<14>("v_lshlrev_b32", out, 64);
    run<15>("v_or_b32", out, 64);
    run<16>("v_and_b32", out, 64);
    run<17>("v_add_u32", out, 64);
    run<18>("v_cvt_f32_u32", out, 64);
    run<19>("v_cvt_i32_f32", out, 64);
    run<28>("v_cvt_f32_ubyte1", out, 64);
    run<20>("v_mov_b32", out, 64);
    run<21>("v_mov_b32_dpp", out, 64);
    run<27>("v_add_f32_dpp", out, 64);
    run<22>("v_lshl_or_b32", out, 64);
    run<23>("v_and_or_b32", out, 64);
    run<29>("v_perm_b32", out, 64);
    run<26>("v_cndmask_b32", out, 64);
    return 0;
}
