// tools/ubench_valu.hip — measurement tool (not product code): issue rate of v_fma_f32 vs
// v_pk_fma_f32 on gfx950, with VGPR and with SGPR multiplicands, at 1..8 waves per SIMD.
// Decides whether the sphere sweep should pair two spheres per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float sa, float sb)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float m = 0.999f + blockIdx.x * 1e-9f;
    v2f m2 = {m, m};
    v2f s2 = {sa, sb};
    for (int i = 0; i < iters; ++i)
    {
        if (MODE == 0)
        {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "v"(m), "v"(a0));)
        }
        else if (MODE == 1)
        {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                              : "v"(m2), "v"(p0));)
        }
        else if (MODE == 2)
        {
            REP8(asm volatile("v_fma_f32 %0, %8, %0, %9\n v_fma_f32 %1, %8, %1, %9\n v_fma_f32 %2, %8, %2, %9\n v_fma_f32 %3, %8, %3, %9\n"
                              "v_fma_f32 %4, %8, %4, %9\n v_fma_f32 %5, %8, %5, %9\n v_fma_f32 %6, %8, %6, %9\n v_fma_f32 %7, %8, %7, %9"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                              : "s"(sa), "v"(a0));)
        }
        else if (MODE >= 4)
        {
            // one-operand-pattern probes of the instructions the tree kernel's node test is made of
#define R1_PROBE(INS, T, C)                                                                                                              \
    REP8(asm volatile(INS " %0, %0, %8\n " INS " %1, %1, %8\n " INS " %2, %2, %8\n " INS " %3, %3, %8\n " INS " %4, %4, %8\n " INS           \
                          " %5, %5, %8\n " INS " %6, %6, %8\n " INS " %7, %7, %8"                                                          \
                      : "+v"(T##0), "+v"(T##1), "+v"(T##2), "+v"(T##3), "+v"(T##4), "+v"(T##5), "+v"(T##6), "+v"(T##7)                    \
                      : "v"(C));)
            if (MODE == 4)
            {
                R1_PROBE("v_pk_add_f32", p, m2)
            }
            else if (MODE == 5)
            {
                R1_PROBE("v_pk_mul_f32", p, m2)
            }
            else if (MODE == 6)
            {
                R1_PROBE("v_add_f32", a, m)
            }
            else if (MODE == 7)
            {
                REP8(asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n"
                                  "v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                  : "v"(m));)
            }
            else if (MODE == 8)
            {
                REP8(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                                  "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                  : "v"(m), "v"(sa));)
            }
            else
            {
                REP8(asm volatile("v_cvt_f32_f16_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                  "v_cvt_f32_f16_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                  "v_cvt_f32_f16_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                  "v_cvt_f32_f16_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                  "v_cvt_f32_f16_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                  "v_cvt_f32_f16_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                  "v_cvt_f32_f16_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
                                  "v_cvt_f32_f16_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                                  : "v"(m));)
            }
        }
        else
        {
            REP8(asm volatile("v_pk_fma_f32 %0, %8, %0, %9\n v_pk_fma_f32 %1, %8, %1, %9\n v_pk_fma_f32 %2, %8, %2, %9\n v_pk_fma_f32 %3, %8, %3, %9\n"
                              "v_pk_fma_f32 %4, %8, %4, %9\n v_pk_fma_f32 %5, %8, %5, %9\n v_pk_fma_f32 %6, %8, %6, %9\n v_pk_fma_f32 %7, %8, %7, %9"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                              : "s"(s2), "v"(p0));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 12345.678f)
        out[0] = r;
}

template <int MODE>
static void run(const char *name, int cus, float *d)
{
    const int iters = 4000;
    for (int wps = (MODE >= 4 ? 4 : 1); wps <= 8; wps *= 2)
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(cus * wps), dim3(256), 0, 0, d, 10, 0.999f, 1.001f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(cus * wps), dim3(256), 0, 0, d, iters, 0.999f, 1.001f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        double wave_instr_per_simd = (double)iters * 64 * wps; // each wave issues iters*64 instrs; wps waves per SIMD
        double ns_per_instr = ms * 1e6 / wave_instr_per_simd;
        double lanes_fma = (double)cus * 4 * wps * 64 * iters * 64 * ((MODE & 1) ? 2 : 1);
        printf("%-28s waves/SIMD %d: %8.3f ms  %6.3f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)  %7.1f TFLOP/s\n", name, wps, ms,
               ns_per_instr, ns_per_instr * 2.4, lanes_fma * 2 / (ms * 1e-3) / 1e12);
    }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    float *d;
    hipMalloc(&d, 4);
    run<0>("v_fma_f32 (vgpr)", prop.multiProcessorCount, d);
    run<1>("v_pk_fma_f32 (vgpr)", prop.multiProcessorCount, d);
    run<2>("v_fma_f32 (sgpr src0)", prop.multiProcessorCount, d);
    run<3>("v_pk_fma_f32 (sgpr-pair src0)", prop.multiProcessorCount, d);
    run<4>("v_pk_add_f32", prop.multiProcessorCount, d);
    run<5>("v_pk_mul_f32", prop.multiProcessorCount, d);
    run<6>("v_add_f32", prop.multiProcessorCount, d);
    run<7>("v_cvt_f32_f16", prop.multiProcessorCount, d);
    run<8>("v_max3_f32", prop.multiProcessorCount, d);
    run<9>("v_cvt_f32_f16_sdwa (high half)", prop.multiProcessorCount, d);
    return 0;
}
