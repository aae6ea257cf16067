// tools/ubench_xcd_visibility.hip — what does it take for data written by one wave to be read correctly by a wave on ANOTHER XCD
// (its own L2) inside one kernel, and what does it cost?  Behind it: resolving a tile inside the trace kernel —
// many waves on all XCDs store 16-byte sample records, count them into a per-tile counter with a device-scope atomic, and the
// wave whose add completes the tile reads every record of the tile.
//
// Part 1 (correctness).  G groups of 16 single-wave workgroups (consecutive workgroups sit on different XCDs).  Round r: workgroup w
// stores 64 x 16 bytes of pattern(w, r) into region [r][w], runs the FENCE variant, then lane 0 adds 1 to count[r][group]; the
// workgroup that sees 15 reads the 16 regions of the group with the LOAD variant and counts wrong words.  The whole thing runs twice
// over the same memory with different patterns, so the L2s hold stale lines of the first pass.
//   STORE: 0 plain global_store_dwordx4   1 two 64-bit relaxed agent-scope atomic stores (global_store_dwordx2 sc1)   2 nontemporal
//   FENCE: 0 none   1 s_waitcnt vmcnt(0)   2 __atomic_thread_fence(release, agent) (buffer_wbl2 sc1 + waits)
//   LOAD : 0 plain   1 64-bit relaxed agent-scope atomic loads (sc1)   2 acquire fence (buffer_inv sc1) + plain loads
// Part 2 (cost).  Every wave of a chip-filling grid runs 2000 x {store 16 B per lane; FENCE; one lane: atomic add}: time per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t pattern(uint32_t w, uint32_t r, uint32_t lane, uint32_t c, uint32_t salt)
{
    uint32_t x = (w * 0x9E3779B1u) ^ (r * 0x85EBCA77u) ^ (lane * 0xC2B2AE3Du) ^ (c * 0x27D4EB2Fu) ^ salt;
    x ^= x >> 15, x *= 0x2C1B3C6Du, x ^= x >> 12;
    return x;
}

template <int STORE>
__device__ __forceinline__ void put(uint4 *p, uint4 v)
{
    if (STORE == 0)
        *p = v;
    else if (STORE == 1)
    {
        unsigned long long *q = (unsigned long long *)p;
        __hip_atomic_store(q, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4 t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, (u4 *)p);
    }
}
template <int FENCE>
__device__ __forceinline__ void fence()
{
    if (FENCE == 1)
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt / lgkmcnt left alone
}
template <>
__device__ __forceinline__ void fence<2>()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
template <int LOAD>
__device__ __forceinline__ uint4 get(const uint4 *p)
{
    if (LOAD == 1)
    {
        const unsigned long long *q = (const unsigned long long *)p;
        const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
    }
    return *p;
}

template <int STORE, int FENCE, int LOAD>
__global__ void __launch_bounds__(64) k_visibility(uint4 *data, uint32_t *count, unsigned long long *wrong, uint32_t rounds, uint32_t salt)
{
    const uint32_t w = blockIdx.x, lane = threadIdx.x, nw = gridDim.x, group = w >> 4;
    for (uint32_t r = 0; r < rounds; ++r)
    {
        uint4 *mine = data + ((size_t)r * nw + w) * 64;
        put<STORE>(mine + lane, make_uint4(pattern(w, r, lane, 0, salt), pattern(w, r, lane, 1, salt), pattern(w, r, lane, 2, salt), pattern(w, r, lane, 3, salt)));
        fence<FENCE>();
        uint32_t old = 0;
        if (lane == 0)
            old = atomicAdd(&count[(size_t)r * (nw >> 4) + group], 1u);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == 15u) // this wave completes the group for round r: read everything the sixteen wrote
        {
            if (LOAD == 2)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            uint32_t bad = 0;
            for (uint32_t m = 0; m < 16; ++m)
            {
                const uint32_t ww = (group << 4) + m;
                const uint4 v = get<LOAD>(data + ((size_t)r * nw + ww) * 64 + lane);
                bad += (v.x != pattern(ww, r, lane, 0, salt)) + (v.y != pattern(ww, r, lane, 1, salt)) + (v.z != pattern(ww, r, lane, 2, salt)) +
                       (v.w != pattern(ww, r, lane, 3, salt));
            }
            if (bad)
                atomicAdd(wrong, (unsigned long long)bad);
        }
    }
}

template <int STORE, int FENCE>
__global__ void __launch_bounds__(256) k_cost(uint4 *data, uint32_t *count, uint32_t iters)
{
    const uint32_t gt = blockIdx.x * 256 + threadIdx.x, wave = gt >> 6;
    for (uint32_t i = 0; i < iters; ++i)
    {
        put<STORE>(data + ((size_t)(i & 63u) * gridDim.x * 256 + gt), make_uint4(i, gt, 2u, 3u));
        fence<FENCE>();
        if ((threadIdx.x & 63u) == 0)
            atomicAdd(&count[(wave & 1023u) * 32u], 1u);
    }
}

int main()
{
    const uint32_t NW = 4096, ROUNDS = 48;
    uint4 *data = nullptr;
    uint32_t *count = nullptr;
    unsigned long long *wrong = nullptr;
    const size_t data_bytes = (size_t)ROUNDS * NW * 64 * 16;
    CK(hipMalloc((void **)&data, data_bytes > (size_t)64 * 2048 * 256 * 16 ? data_bytes : (size_t)64 * 2048 * 256 * 16));
    CK(hipMalloc((void **)&count, (size_t)ROUNDS * (NW / 16) * 4 + 1024 * 128));
    CK(hipMalloc((void **)&wrong, 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    printf("part 1: %u single-wave workgroups in groups of 16, %u rounds, two passes over the same memory (stale lines in every L2)\n", NW, ROUNDS);
    printf("%-28s %-26s %-22s %12s %10s\n", "store", "fence before the count", "load", "wrong words", "ms");
    const char *sn[3] = {"plain dwordx4", "2 x 64-bit agent (sc1)", "nontemporal dwordx4"};
    const char *fn[3] = {"none", "s_waitcnt vmcnt(0)", "release fence, agent (wbl2)"};
    const char *ln[3] = {"plain", "64-bit agent loads (sc1)", "acquire fence (inv) + plain"};
#define RUN(S, F, L)                                                                                                   \
    {                                                                                                                  \
        unsigned long long total = 0;                                                                                  \
        float ms_sum = 0;                                                                                              \
        for (uint32_t pass = 0; pass < 3; ++pass)                                                                      \
        {                                                                                                              \
            CK(hipMemset(count, 0, (size_t)ROUNDS * (NW / 16) * 4));                                                   \
            CK(hipMemset(wrong, 0, 8));                                                                                \
            CK(hipEventRecord(a));                                                                                     \
            hipLaunchKernelGGL((k_visibility<S, F, L>), dim3(NW), dim3(64), 0, 0, data, count, wrong, ROUNDS, 0x1234567u * (pass + 1));                 \
            CK(hipEventRecord(b));                                                                                     \
            CK(hipDeviceSynchronize());                                                                                \
            unsigned long long h = 0;                                                                                  \
            CK(hipMemcpy(&h, wrong, 8, hipMemcpyDeviceToHost));                                                        \
            float ms = 0;                                                                                              \
            CK(hipEventElapsedTime(&ms, a, b));                                                                        \
            total += h, ms_sum += ms;                                                                                  \
        }                                                                                                              \
        printf("%-28s %-26s %-22s %12llu %10.3f\n", sn[S], fn[F], ln[L], total, ms_sum / 3);                           \
    }
    RUN(0, 0, 0) RUN(0, 1, 0) RUN(0, 2, 0) RUN(0, 2, 1) RUN(0, 2, 2)
    RUN(1, 0, 1) RUN(1, 1, 0) RUN(1, 1, 1) RUN(1, 1, 2) RUN(1, 2, 1)
    RUN(2, 1, 1) RUN(2, 2, 2)
#undef RUN
    printf("\npart 2: 2048 workgroups x 4 waves, 2000 x {16-byte store per lane; fence; one atomic per wave}\n");
    printf("%-28s %-30s %12s\n", "store", "fence", "ns / iteration");
#define COST(S, F)                                                                                                     \
    {                                                                                                                  \
        float best = 1e9f;                                                                                             \
        for (int t = 0; t < 3; ++t)                                                                                    \
        {                                                                                                              \
            CK(hipEventRecord(a));                                                                                     \
            hipLaunchKernelGGL((k_cost<S, F>), dim3(2048), dim3(256), 0, 0, data, count + ROUNDS * (NW / 16), 2000u);  \
            CK(hipEventRecord(b));                                                                                     \
            CK(hipDeviceSynchronize());                                                                                \
            float ms = 0;                                                                                              \
            CK(hipEventElapsedTime(&ms, a, b));                                                                        \
            best = ms < best ? ms : best;                                                                              \
        }                                                                                                              \
        printf("%-28s %-30s %12.1f\n", sn[S], fn[F], best * 1e6 / 2000.0);                                             \
    }
    COST(0, 0) COST(0, 1) COST(0, 2) COST(1, 0) COST(1, 1) COST(1, 2) COST(2, 1)
#undef COST
    return 0;
}
