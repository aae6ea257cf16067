// tools/ubench_halfwave.hip — does a wave64 VALU instruction whose upper (or lower) 32 lanes are all inactive issue
// faster on gfx950?  One kernel, a long chain-free stream of v_fma_f32 under three exec masks: all 64 lanes, lanes
// 0..31 only, every second lane.  Prints ns per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    const int lane = threadIdx.x & 63;
    const bool on = MODE == 0 ? true : (MODE == 1 ? lane < 32 : (MODE == 2 ? (lane & 1) == 0 : lane < 16));
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    if (on)
    {
        for (int i = 0; i < iters; ++i)
        {
#pragma unroll
            for (int u = 0; u < 16; ++u)
            {
                a0 = __builtin_fmaf(a0, b, c), a1 = __builtin_fmaf(a1, b, c), a2 = __builtin_fmaf(a2, b, c), a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c), a5 = __builtin_fmaf(a5, b, c), a6 = __builtin_fmaf(a6, b, c), a7 = __builtin_fmaf(a7, b, c);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
static void run(const char *name, float *d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = (double)blocks * 4 / 1024.0; // 256 CUs x 4 SIMDs
    const double instr = (double)iters * 128.0 * waves_per_simd;
    printf("%-28s %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, ms * 1e6 / instr, ms * 1e6 / instr * 2.4);
}

int main()
{
    float *d;
    const int blocks = 256 * 8; // 8 workgroups per CU = 8 waves per SIMD
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    run<0>("v_fma_f32, 64 lanes", d, blocks, 4000);
    run<1>("v_fma_f32, lanes 0..31", d, blocks, 4000);
    run<2>("v_fma_f32, even lanes", d, blocks, 4000);
    run<3>("v_fma_f32, lanes 0..15", d, blocks, 4000);
    return 0;
}
