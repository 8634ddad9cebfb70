// bank_probe.hip -- does a packed-f32 instruction with TWO vector-register pair sources cost more when the pairs share a
// register bank (index mod 4)?  Fixed physical registers, 16 independent destinations, 1 / 4 / 5 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP16(INS, A, B)                                                                                        \
    INS " v[40:41], " A ", " B "\n\t" INS " v[42:43], " A ", " B "\n\t" INS " v[44:45], " A ", " B "\n\t"       \
    INS " v[46:47], " A ", " B "\n\t" INS " v[48:49], " A ", " B "\n\t" INS " v[50:51], " A ", " B "\n\t"       \
    INS " v[52:53], " A ", " B "\n\t" INS " v[54:55], " A ", " B "\n\t" INS " v[56:57], " A ", " B "\n\t"       \
    INS " v[58:59], " A ", " B "\n\t" INS " v[60:61], " A ", " B "\n\t" INS " v[62:63], " A ", " B "\n\t"       \
    INS " v[64:65], " A ", " B "\n\t" INS " v[66:67], " A ", " B "\n\t" INS " v[68:69], " A ", " B "\n\t"       \
    INS " v[70:71], " A ", " B
#define CLOB "v4","v5","v6","v7","v8","v9","v10","v11","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","s20","s21"

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    asm volatile("v_mov_b32 v4, 1.0\n\tv_mov_b32 v5, 2.0\n\tv_mov_b32 v6, 1.0\n\tv_mov_b32 v7, 2.0\n\tv_mov_b32 v8, 0.5\n\tv_mov_b32 v9, 0.5\n\t"
                 "v_mov_b32 v10, 0.5\n\tv_mov_b32 v11, 0.5\n\ts_mov_b32 s20, 1.0\n\ts_mov_b32 s21, 2.0" ::: CLOB);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (OP == 0) asm volatile(REP16("v_pk_add_f32", "v[4:5]", "v[8:9]") ::: CLOB);          // banks {0,1} + {0,1}
            if (OP == 1) asm volatile(REP16("v_pk_add_f32", "v[4:5]", "v[10:11]") ::: CLOB);        // {0,1} + {2,3}
            if (OP == 2) asm volatile(REP16("v_pk_fma_f32", "v[4:5], s[20:21]", "v[8:9]") ::: CLOB);
            if (OP == 3) asm volatile(REP16("v_pk_fma_f32", "v[4:5], s[20:21]", "v[10:11]") ::: CLOB);
            if (OP == 4) asm volatile(REP16("v_pk_mul_f32", "v[4:5]", "s[20:21]") ::: CLOB);
            if (OP == 5) asm volatile(REP16("v_pk_fma_f32", "v[4:5], v[8:9]", "v[10:11]") ::: CLOB); // three vector pairs
            if (OP == 6) asm volatile(REP16("v_pk_fma_f32", "v[4:5], v[6:7]", "v[10:11]") ::: CLOB);
            if (OP == 7) asm volatile(REP16("v_pk_add_f32", "v[4:5]", "v[4:5]") ::: CLOB);          // the same pair twice
        }
    }
    if (iters < 0) out[0] = 1;
}

template <int OP>
int run(const char *name, float *out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int wps : {1, 4, 5}) {
        const int blocks = 256 * wps;
        k<OP><<<blocks, 256>>>(out, 100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k<OP><<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)iters * 64 * wps;
        printf("%-44s %d waves/SIMD: %.2f cycles per instruction at 2.4 GHz\n", name, wps, ms * 1e6 / n * 2.4);
    }
    return 0;
}

int main()
{
    float *out; CK(hipMalloc(&out, 4));
    run<4>("pk_mul v[4:5], s", out);
    run<0>("pk_add v[4:5], v[8:9]   (same banks)", out);
    run<1>("pk_add v[4:5], v[10:11] (other banks)", out);
    run<7>("pk_add v[4:5], v[4:5]", out);
    run<2>("pk_fma v[4:5], s, v[8:9]   (same banks)", out);
    run<3>("pk_fma v[4:5], s, v[10:11] (other banks)", out);
    run<5>("pk_fma v[4:5], v[8:9], v[10:11]", out);
    run<6>("pk_fma v[4:5], v[6:7], v[10:11]", out);
    return 0;
}
