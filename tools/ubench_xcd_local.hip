// tools/ubench_xcd_local.hip — can work be kept INSIDE one XCD, so that its data never has to cross between the eight L2s?
// (1) HW_REG_XCC_ID: which XCD a workgroup runs on; how workgroups are dealt to XCDs.
// (2) atomics WITHOUT the sc1 bit (workgroup scope) are performed in the issuing XCD's L2: are they coherent among the workgroups of
//     that XCD (each XCD counting on a line of its own), and how fast are they on one line compared with device-scope atomics?
// (3) plain stores + s_waitcnt vmcnt(0) + an L2-local atomic flag, read on the same XCD with sc0 loads: does the consumer see the data?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; } // HW_REG_XCC_ID[3:0]

__global__ void k_ids(uint32_t *ids)
{
    if (threadIdx.x == 0)
        ids[blockIdx.x] = xcc_id();
}

// every workgroup adds 1 (per wave) to ITS XCD's counter, `iters` times, with an L2-local atomic (LOCAL) or a device-scope one
template <bool LOCAL>
__global__ void __launch_bounds__(256) k_count(uint32_t *counters /* [16][32] */, uint32_t iters)
{
    const uint32_t x = xcc_id();
    uint32_t *c = counters + 32u * x;
    for (uint32_t i = 0; i < iters; ++i)
        if ((threadIdx.x & 63u) == 0)
        {
            if (LOCAL)
                (void)__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                (void)__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
}

// per XCD: producers (all workgroups but the first to arrive on the XCD) write 64 x 16 B per round into their slot, wait for the
// stores, add 1 to the XCD's round counter (L2-local); the consumer (first workgroup on the XCD, elected with an L2-local atomic) waits
// for the counter to reach the number of producers that have arrived ... simplified: producers write ROUNDS regions and count; at the end the LAST
// workgroup to finish on each XCD (L2-local countdown of finished workgroups) reads everything its XCD's workgroups wrote and verifies.
__global__ void __launch_bounds__(64) k_visible(uint4 *data, uint32_t *state /* [16][32]: +0 arrived, +1 finished */, uint32_t *owner /* [grid] */,
                                                  unsigned long long *wrong, uint32_t rounds, uint32_t salt, int sc0_loads)
{
    const uint32_t x = xcc_id(), w = blockIdx.x, lane = threadIdx.x;
    uint32_t *st = state + 32u * x;
    if (lane == 0)
    {
        owner[w] = x;
        (void)__hip_atomic_fetch_add(st, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (uint32_t r = 0; r < rounds; ++r)
        data[((size_t)r * gridDim.x + w) * 64 + lane] = make_uint4(w ^ salt, r, lane, x);
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the stores are performed in this XCD's L2
    uint32_t fin = 0;
    if (lane == 0)
        fin = __hip_atomic_fetch_add(st + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1u;
    fin = __builtin_amdgcn_readfirstlane(fin);
    uint32_t arrived = 0;
    if (lane == 0)
        arrived = __hip_atomic_load(st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (fin != arrived)
        return; // somebody on this XCD is still writing (or has not started: then a later one is last)
    // last on this XCD so far: verify what every finished workgroup of this XCD wrote (owner[] is written with plain stores too)
    uint32_t bad = 0, seen = 0;
    for (uint32_t ww = 0; ww < gridDim.x; ++ww)
    {
        const uint32_t ow = sc0_loads ? __hip_atomic_load(owner + ww, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : owner[ww];
        if (ow != x)
            continue;
        ++seen;
        for (uint32_t r = 0; r < rounds; ++r)
        {
            const uint4 *p = data + ((size_t)r * gridDim.x + ww) * 64 + lane;
            uint4 v;
            if (sc0_loads)
            {
                const unsigned long long a = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned long long b = __hip_atomic_load((const unsigned long long *)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                v = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
            }
            else
                v = *p;
            bad += (v.x != (ww ^ salt)) + (v.y != r) + (v.z != lane) + (v.w != x);
        }
    }
    if (bad)
        atomicAdd(wrong, (unsigned long long)bad);
    if (lane == 0)
        atomicAdd(wrong + 1, (unsigned long long)seen);
}

int main()
{
    const uint32_t G = 2048;
    uint32_t *ids = nullptr, *counters = nullptr, *owner = nullptr;
    uint4 *data = nullptr;
    unsigned long long *wrong = nullptr;
    CK(hipMalloc((void **)&ids, G * 4));
    CK(hipMalloc((void **)&counters, 16 * 128));
    CK(hipMalloc((void **)&owner, G * 4));
    CK(hipMalloc((void **)&wrong, 16));
    const uint32_t ROUNDS = 16;
    CK(hipMalloc((void **)&data, (size_t)ROUNDS * G * 64 * 16));
    std::vector<uint32_t> h(G);
    hipLaunchKernelGGL(k_ids, dim3(G), dim3(64), 0, 0, ids);
    CK(hipMemcpy(h.data(), ids, G * 4, hipMemcpyDeviceToHost));
    printf("(1) HW_REG_XCC_ID of workgroups 0..31:");
    for (int i = 0; i < 32; ++i)
        printf(" %u", h[i]);
    uint32_t hist[16] = {0}, rr = 0;
    for (uint32_t i = 0; i < G; ++i)
        hist[h[i] & 15]++, rr += (h[i] == (i & 7u));
    printf("\n    histogram over %u workgroups:", G);
    for (int i = 0; i < 16; ++i)
        if (hist[i])
            printf(" xcc%d=%u", i, hist[i]);
    printf("   workgroup i on XCD i %% 8: %u of %u\n", rr, G);

    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    printf("(2) 2048 workgroups x 4 waves x 2000 atomic adds, each XCD on a line of its own\n");
    for (int local = 1; local >= 0; --local)
    {
        CK(hipMemset(counters, 0, 16 * 128));
        CK(hipEventRecord(a));
        if (local)
            hipLaunchKernelGGL(k_count<true>, dim3(2048), dim3(256), 0, 0, counters, 2000u);
        else
            hipLaunchKernelGGL(k_count<false>, dim3(2048), dim3(256), 0, 0, counters, 2000u);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        std::vector<uint32_t> c(16 * 32);
        CK(hipMemcpy(c.data(), counters, 16 * 128, hipMemcpyDeviceToHost));
        unsigned long long sum = 0;
        for (int i = 0; i < 16; ++i)
            sum += c[32 * i];
        printf("    %-28s sum %llu (expected %llu)  %.3f ms  = %.0f M atomics/s per line\n", local ? "L2-local (workgroup scope)" : "device scope (sc1)", sum,
               2048ull * 4 * 2000, ms, 2048.0 * 4 * 2000 / 8 / (ms * 1e-3) / 1e6);
    }
    printf("(3) plain stores -> vmcnt(0) -> L2-local count; the last workgroup of an XCD reads what its XCD wrote (3 passes over the same memory)\n");
    for (int sc0 = 0; sc0 < 2; ++sc0)
    {
        unsigned long long total_bad = 0, total_seen = 0;
        for (uint32_t pass = 0; pass < 3; ++pass)
        {
            CK(hipMemset(counters, 0, 16 * 128));
            CK(hipMemset(wrong, 0, 16));
            hipLaunchKernelGGL(k_visible, dim3(G), dim3(64), 0, 0, data, counters, owner, wrong, ROUNDS, 0x51u * (pass + 1), sc0);
            CK(hipDeviceSynchronize());
            unsigned long long hw[2];
            CK(hipMemcpy(hw, wrong, 16, hipMemcpyDeviceToHost));
            total_bad += hw[0], total_seen += hw[1];
        }
        printf("    %-12s wrong words %llu; workgroups verified by a last-on-XCD reader %llu (of %u x 3 at most)\n", sc0 ? "sc0 loads:" : "plain loads:", total_bad,
               total_seen, G);
    }
    return 0;
}
