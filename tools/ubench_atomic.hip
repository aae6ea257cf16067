// tools/ubench_atomic.hip — throughput of returning atomicAdd on ONE address vs Q addresses, issued by
// one lane per wave from a full persistent grid (the access pattern of the trace kernel's sample queue).
// Measurement tool, not product code.   hipcc --offload-arch=gfx950 -O2 -o ubench_atomic tools/ubench_atomic.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) k(uint32_t *ctr, int nq, int stride_words, int per_wave, uint32_t *sink)
{
    uint32_t acc = 0;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    uint32_t *c = ctr + (size_t)(wave % nq) * stride_words;
    for (int i = 0; i < per_wave; ++i)
    {
        uint32_t v = 0;
        if ((threadIdx.x & 63) == 0)
            v = atomicAdd(c, 64u);
        v = __builtin_amdgcn_readfirstlane(v);
        acc += v;
        // a little dependent work between fetches, like a wave consuming its chunk
        for (int j = 0; j < 64; ++j)
            acc = acc * 1664525u + 1013904223u;
    }
    if (acc == 0x12345678u)
        sink[0] = acc;
}

int main()
{
    uint32_t *ctr, *sink;
    hipMalloc(&ctr, 1 << 20);
    hipMalloc(&sink, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 1536, per_wave = 64;
    const long total = (long)blocks * 4 * per_wave;
    int nqs[] = {1, 2, 4, 8, 16, 64, 6144};
    int strides[] = {1, 32, 1024};
    for (int s : strides)
        for (int nq : nqs)
        {
            if ((size_t)nq * s * 4 > (1u << 20))
                continue;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep)
            {
                hipMemset(ctr, 0, 1 << 20);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, nq, s, per_wave, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("queues %5d stride %5d words: %8.3f ms for %ld atomics = %7.1f ns per atomic (aggregate), %6.1f M/s\n", nq, s, best, total,
                   best * 1e6 / total, total / best / 1e3);
        }
    return 0;
}
