// tools/ubench_pcie_store.hip — can a kernel write a frame's pixels to page-locked HOST memory as fast as a copy engine does?
// 1200 x 800 x 3 bytes (2.88 MB), written (a) by hipMemcpyAsync device -> pinned host, (b) by a kernel with contiguous 16-byte stores,
// (c) by a kernel in the pattern a per-tile resolve would produce: 96-byte runs (one 32-pixel tile row) W*3 bytes apart, dword stores,
// (d) the same with byte stores (what r1_resolve_kernel does today).  Question behind it: a zero-copy resolve for the synchronous frame.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static const int W = 1200, H = 800;
__global__ void k_contig(uint4 *out, size_t n16) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n16) out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u); }
// one workgroup = 256 pixels of a tile: 8 rows x 32 pixels; threads 0..191 store one dword each (24 per row)
__global__ void k_tiles_dword(uint8_t *out, int tiles_x)
{
    const int tile = blockIdx.y, part = blockIdx.x; // part: 4 groups of 8 rows
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int t = threadIdx.x;
    if (t < 192)
    {
        const int row = t / 24, dw = t % 24;
        const int y = ty * 32 + part * 8 + row, x0 = tx * 32;
        if (y < H && x0 + 32 <= W)
            *(uint32_t *)(out + ((size_t)y * W + x0) * 3 + dw * 4) = (uint32_t)t;
    }
}
__global__ void k_tiles_byte(uint8_t *out, int tiles_x)
{
    const int tile = blockIdx.y, part = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int t = threadIdx.x, row = t / 32, lx = t % 32;
    const int y = ty * 32 + part * 8 + row, x = tx * 32 + lx;
    if (y < H && x < W)
    {
        uint8_t *o = out + ((size_t)y * W + x) * 3;
        o[0] = (uint8_t)t, o[1] = 1, o[2] = 2;
    }
}
int main()
{
    const size_t bytes = (size_t)W * H * 3;
    uint8_t *host = nullptr, *dev = nullptr, *hdev = nullptr;
    CK(hipHostMalloc((void **)&host, bytes, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&hdev, host, 0));
    CK(hipMalloc((void **)&dev, bytes));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 31) / 32;
    auto run = [&](const char *name, auto fn) {
        float best = 1e9f; double wall_best = 1e9;
        for (int i = 0; i < 12; ++i)
        {
            auto t0 = std::chrono::steady_clock::now();
            hipEventRecord(a, st); fn(); hipEventRecord(b, st); hipStreamSynchronize(st);
            auto t1 = std::chrono::steady_clock::now();
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (i > 1 && ms < best) best = ms;
            double w = std::chrono::duration<double, std::milli>(t1 - t0).count();
            if (i > 1 && w < wall_best) wall_best = w;
        }
        printf("%-44s %7.1f us on the device (%5.1f GB/s)   %7.1f us wall incl. submit + sync\n", name, best * 1e3, bytes / (best * 1e-3) / 1e9, wall_best * 1e3);
    };
    run("hipMemcpyAsync device -> pinned host", [&] { hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st); });
    run("kernel, contiguous 16-byte stores to host", [&] { hipLaunchKernelGGL(k_contig, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, st, (uint4 *)hdev, bytes / 16); });
    run("kernel, 96-byte tile rows, dword stores", [&] { hipLaunchKernelGGL(k_tiles_dword, dim3(4, tiles_x * tiles_y), dim3(256), 0, st, hdev, tiles_x); });
    run("kernel, tile rows, byte stores (today's form)", [&] { hipLaunchKernelGGL(k_tiles_byte, dim3(4, tiles_x * tiles_y), dim3(256), 0, st, hdev, tiles_x); });
    run("kernel, contiguous 16-byte stores to DEVICE", [&] { hipLaunchKernelGGL(k_contig, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, st, (uint4 *)dev, bytes / 16); });
    return 0;
}
