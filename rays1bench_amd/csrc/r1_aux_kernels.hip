// r1_aux_kernels.hip — the kernels around the trace kernel (wavefront variant, resolve, batch counts, assemble) and the launch
// dispatch called from r1_capi.cpp.  The trace kernel template and its device functions: r1_trace.hpp.
#include "r1_trace.hpp"


// ============================================================================================
// Wavefront variant (R1_VARIANT_WAVEFRONT; SURVEY.md §8f-3 "the step either side of the
// megakernel"): generate -> [intersect -> shade] x (max_bounces + 1) with the path state and
// one queue of live path slots per color() level in HBM.  Same device functions as the
// megakernel (start_sample, sweep_bvh, shade_level), so the samples are bit-identical; what
// differs is where the state lives between steps.  Measured against the megakernel in
// DESIGN.md §4.5.
// ============================================================================================
namespace
{

// all lanes of the wave call this together; lanes with `want` get consecutive queue slots
__device__ __forceinline__ void wave_append(uint32_t *counter, uint32_t *queue, const bool want, const uint32_t value)
{
    const unsigned long long m = __ballot(want);
    if (m == 0ull)
        return;
    const int lane = (int)(threadIdx.x & 63u);
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader)
        base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    if (want)
        queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

} // namespace

// sample slot k -> primary ray (rayweek1.cpp:759-760) in path slot k; void slots are skipped
__global__ void __launch_bounds__(R1_BLOCK) r1_wf_generate(const R1WaveArgs W)
{
    const uint32_t stride = gridDim.x * R1_BLOCK;
    const uint32_t rounds = (W.n_paths + stride - 1) / stride; // every lane makes the same number of trips (wave_append)
    for (uint32_t r = 0; r < rounds; ++r)
    {
        const uint32_t k = r * stride + blockIdx.x * R1_BLOCK + threadIdx.x;
        Path p;
        bool valid = false;
        if (k < W.n_paths)
            valid = start_sample(W.t, p, k);
        if (valid)
            path_store(W.paths, W.n_paths, k, p);
        wave_append(&W.counts[0], W.queue[0], valid, k);
    }
}

// Hitable::hit for every path of the level's queue (rayweek1.cpp:519)
__global__ void __launch_bounds__(R1_BLOCK) r1_wf_intersect(const R1WaveArgs W)
{
    extern __shared__ uint32_t s_trav[];
    const uint32_t n = W.counts[W.level];
    const uint32_t *q = W.queue[W.level & 1];
    const uint32_t stride = gridDim.x * R1_BLOCK;
    for (uint32_t i = blockIdx.x * R1_BLOCK + threadIdx.x; i < n; i += stride)
    {
        const uint32_t slot = q[i];
        const float4 a = W.paths[slot], b = W.paths[(size_t)W.n_paths + slot];
        float t_hit = FLT_MAX;
        int hit = -1;
        sweep_bvh<false>(W.t.scene, true, mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), t_hit, hit, s_trav, (int)threadIdx.x, nullptr);
        W.hits[slot] = make_float2(t_hit, __int_as_float(hit));
    }
}

// the rest of color() for the level: scatter into the next level's queue, or finish the sample
__global__ void __launch_bounds__(R1_BLOCK) r1_wf_shade(const R1WaveArgs W)
{
    const uint32_t n = W.counts[W.level];
    const uint32_t *q = W.queue[W.level & 1];
    uint32_t *qn = W.queue[(W.level + 1) & 1];
    const uint32_t stride = gridDim.x * R1_BLOCK;
    const uint32_t rounds = (n + stride - 1) / stride;
    unsigned long long lane_rays = 0;
    for (uint32_t r = 0; r < rounds; ++r)
    {
        const uint32_t i = r * stride + blockIdx.x * R1_BLOCK + threadIdx.x;
        bool goes_on = false;
        uint32_t slot = 0;
        if (i < n)
        {
            slot = q[i];
            Path p;
            path_load(W.paths, W.n_paths, slot, p);
            const float2 h = W.hits[slot];
            V3 col;
            if (shade_level<true>(W.t, p, __float_as_int(h.y), h.x, nullptr, W.n_paths, slot, (int)threadIdx.x, col))
            {
                W.t.samples[p.k] = make_float4(col.x, col.y, col.z, __uint_as_float(path_rays(p)));
                lane_rays += path_rays(p);
            }
            else
            {
                path_store(W.paths, W.n_paths, slot, p);
                goes_on = true;
            }
        }
        wave_append(&W.counts[W.level + 1], qn, goes_on, slot);
    }
    for (int off = 32; off > 0; off >>= 1)
        lane_rays += __shfl_down(lane_rays, off, 64);
    if ((threadIdx.x & 63u) == 0 && lane_rays)
        atomicAdd(W.t.num_rays, lane_rays);
}

// ============================================================================================
// Resolve: one thread per pixel of this shard; sums the spp samples in sample order and
// quantises exactly as rayweek1.cpp:765-775.
// ============================================================================================
__global__ void __launch_bounds__(256) r1_resolve_kernel(const R1ResolveArgs A)
{
    if (blockIdx.x == 0 && blockIdx.y == 0 && A.rays_src)
    {
        static_assert(R1_COUNTER_BYTES == 256 * 16, "one 16-byte store per thread zeroes the counter block");
        if (threadIdx.x == 0)
            *A.rays_dst = *A.rays_src;
        __syncthreads();
        if (A.reset)
            ((uint4 *)A.reset)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t tiles_stride = gridDim.y;
    const uint32_t tiles_all = A.n_local_tiles * (A.n_frames ? A.n_frames : 1u);
    for (uint32_t lt_all = blockIdx.y; lt_all < tiles_all; lt_all += tiles_stride)
    {
        const uint32_t f = lt_all / A.n_local_tiles, lt = lt_all - f * A.n_local_tiles; // frame of the batch, its local tile
        uint8_t *const out = A.out + (size_t)f * A.out_stride;
        const uint32_t tile = (uint32_t)A.shard + lt * (uint32_t)A.num_shards;
        const int x0 = (int)(tile % (uint32_t)A.tiles_x) * A.tile_w;
        const int y0 = (int)(tile / (uint32_t)A.tiles_x) * A.tile_h;
        const int tw = min(A.tile_w, A.width - x0);
        const int th = min(A.tile_h, A.height - y0);
        const uint32_t base = lt_all * A.full;
        unsigned long long rays = 0;
        for (uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x; pix < (uint32_t)(A.tile_w * A.tile_h); pix += gridDim.x * blockDim.x)
        {
            const int ly = (int)(pix / (uint32_t)A.tile_w);
            const int lx = (int)(pix - (uint32_t)ly * (uint32_t)A.tile_w);
            if (lx >= tw || ly >= th)
                continue; // void slots of an edge tile
            const uint32_t tile_px = (uint32_t)(A.tile_w * A.tile_h);
            const float4 *s = A.samples + base + pix; // [tile][sample][pixel]
            float cr = 0, cg = 0, cb = 0;
            for (int i = 0; i < A.spp; ++i)
            {
                const float4 v = s[(size_t)i * tile_px];
                cr += v.x, cg += v.y, cb += v.z; // col += color(...) rayweek1.cpp:762
                rays += __float_as_uint(v.w);
            }
            cr *= A.inv_spp, cg *= A.inv_spp, cb *= A.inv_spp;
            cr = ieee_sqrt(cr), cg = ieee_sqrt(cg), cb = ieee_sqrt(cb);
            const uint8_t r = (uint8_t)(int)(cr * 255.99f);
            const uint8_t g = (uint8_t)(int)(cg * 255.99f);
            const uint8_t b = (uint8_t)(int)(cb * 255.99f);
            size_t o;
            if (A.block_layout)
                o = ((size_t)lt * A.tile_h * A.tile_w + (size_t)ly * A.tile_w + lx) * 3;
            else
                o = ((size_t)(y0 + ly) * A.width + (x0 + lx)) * 3;
            out[o + 0] = r;
            out[o + 1] = g;
            out[o + 2] = b;
        }
        if (A.frame_rays) // frame batches: the frame's ray count is the sum over its samples (rayweek1.cpp:809-813)
        {
            // no atomics: 15 000 waves adding to one word per frame cost more than the whole resolve (a returning or
            // non-returning atomic on ONE line sustains ~88 M/s on this chip, tools/ubench_atomic.hip).  Every workgroup
            // stores ONE partial sum per tile it touches; r1_batch_counts_kernel adds them up per frame.
            __shared__ unsigned long long s_part[4];
            for (int off = 32; off > 0; off >>= 1)
                rays += __shfl_down(rays, off, 64);
            if ((threadIdx.x & 63u) == 0)
                s_part[threadIdx.x >> 6] = rays;
            __syncthreads();
            if (threadIdx.x == 0)
                A.frame_rays[(size_t)lt_all * gridDim.x + blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
            __syncthreads();
        }
    }
}

// Frame batches: frame f's ray count = the sum of the partial sums the resolve launch left per (tile, workgroup column);
// one workgroup per frame; the count goes next to the frame's pixels (out + f * out_stride + rays_offset).
__global__ void __launch_bounds__(256)
    r1_batch_counts_kernel(const unsigned long long *__restrict__ partial, uint32_t per_frame, uint8_t *__restrict__ out, size_t out_stride,
                           size_t rays_offset)
{
    __shared__ unsigned long long s_part[4];
    const unsigned long long *src = partial + (size_t)blockIdx.x * per_frame;
    unsigned long long sum = 0;
    for (uint32_t i = threadIdx.x; i < per_frame; i += blockDim.x)
        sum += src[i];
    for (int off = 32; off > 0; off >>= 1)
        sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63u) == 0)
        s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0)
        *(unsigned long long *)(out + (size_t)blockIdx.x * out_stride + rays_offset) = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// Scatter gathered dense tile blocks (shard-major) into a row-major image.  Frame batches: blockIdx.y = frame; the
// frame's blocks start frame_in bytes after `blocks` (gathered layout [shard][frame][record]) and its image frame_out
// bytes after `rgb`.
__global__ void __launch_bounds__(256)
    r1_assemble_kernel(const uint8_t *__restrict__ blocks_all, uint8_t *__restrict__ rgb_all, int width, int height, int tile_w, int tile_h,
                       int tiles_x, int num_shards, size_t shard_stride, size_t frame_in, size_t frame_out, size_t total_offset, long long total_out,
                       int want_total)
{
    const uint8_t *__restrict__ blocks = blocks_all + (size_t)blockIdx.y * frame_in;
    uint8_t *__restrict__ rgb = rgb_all + (size_t)blockIdx.y * frame_out;
    // gathered RECORDS (block + uint64 count at the end of every record): the frame's ray count is the sum of the
    // shards' counts (rayweek1.cpp:809-813), written next to the image so that one copy brings both to the host
    if (want_total && blockIdx.x == 0 && threadIdx.x == 0)
    {
        unsigned long long sum = 0;
        for (int sh = 0; sh < num_shards; ++sh)
            sum += *(const unsigned long long *)(blocks + (size_t)sh * shard_stride + total_offset);
        *(unsigned long long *)(rgb + total_out) = sum;
    }
    const size_t n = (size_t)width * height;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        const int y = (int)(i / (size_t)width), x = (int)(i - (size_t)y * width);
        const int tile = (y / tile_h) * tiles_x + (x / tile_w);
        const int shard = tile % num_shards, lt = tile / num_shards;
        const size_t src = (size_t)shard * shard_stride + (((size_t)lt * tile_h + (size_t)(y % tile_h)) * tile_w + (x % tile_w)) * 3;
        rgb[3 * i + 0] = blocks[src + 0];
        rgb[3 * i + 1] = blocks[src + 1];
        rgb[3 * i + 2] = blocks[src + 2];
    }
}

// R1_LAND: the per-tile countdowns and per-frame accumulators of a context, set when its tiling changes (afterwards the resolvers
// re-arm what they consume): tile t of the launch lacks (valid pixels of its tile) x spp samples.
__global__ void __launch_bounds__(256) r1_land_arm_kernel(uint32_t *tile_cnt, unsigned long long *frame_rays, uint32_t *frame_left, uint32_t n_frames,
                                                          uint32_t n_local_tiles, int width, int height, int spp, int tile_w, int tile_h, int tiles_x,
                                                          int shard, int num_shards)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_frames)
        frame_rays[i] = 0ull, frame_left[i] = n_local_tiles;
    if (i >= n_frames * n_local_tiles)
        return;
    const uint32_t lt = i % n_local_tiles;
    const uint32_t tile = (uint32_t)shard + lt * (uint32_t)num_shards;
    const int x0 = (int)(tile % (uint32_t)tiles_x) * tile_w, y0 = (int)(tile / (uint32_t)tiles_x) * tile_h;
    const int tw = min(tile_w, width - x0), th = min(tile_h, height - y0);
    tile_cnt[(size_t)i * R1_LAND_CNT_STRIDE] = (uint32_t)(tw * th * spp);
}

// six words into device memory, the values travelling in the kernel arguments (copied when the launch is enqueued: no host buffer
// that must stay untouched until a copy has executed)
__global__ void r1_put6_kernel(uint32_t *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f)
{
    dst[0] = a, dst[1] = b, dst[2] = c, dst[3] = d, dst[4] = e, dst[5] = f;
}

// ---- launchers (called from r1_capi.cpp) -----------------------------------------------------

extern "C" hipError_t r1_launch_land_arm(uint32_t *tile_cnt, unsigned long long *frame_rays, uint32_t *frame_left, uint32_t n_frames, uint32_t n_local_tiles,
                                         int width, int height, int spp, int tile_w, int tile_h, int tiles_x, int shard, int num_shards, hipStream_t stream)
{
    const uint32_t n = n_frames * n_local_tiles > n_frames ? n_frames * n_local_tiles : n_frames;
    hipLaunchKernelGGL(r1_land_arm_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tile_cnt, frame_rays, frame_left, n_frames, n_local_tiles, width,
                       height, spp, tile_w, tile_h, tiles_x, shard, num_shards);
    return hipGetLastError();
}

extern "C" hipError_t r1_launch_put6(void *dst, const uint32_t *w, hipStream_t stream)
{
    hipLaunchKernelGGL(r1_put6_kernel, dim3(1), dim3(1), 0, stream, (uint32_t *)dst, w[0], w[1], w[2], w[3], w[4], w[5]);
    return hipGetLastError();
}

// The trace kernel's instantiations live in four translation units (tree / exhaustive sweep x small / big scenes); each exports one
// launch and one occupancy function for its family.
#define R1_TU_DECL(NAME)                                                                                                               \
    extern "C" hipError_t r1_tu_##NAME##_launch(const R1TraceArgs *args, int variant, int mode, int batch, int blocks, size_t dyn_lds, \
                                                hipStream_t stream);                                                                  \
    extern "C" hipError_t r1_tu_##NAME##_occupancy(int variant, int mode, size_t dyn_lds, int *blocks_per_cu);
R1_TU_DECL(tree_small)
R1_TU_DECL(tree_big)
R1_TU_DECL(sweep_small)
R1_TU_DECL(sweep_big)
#undef R1_TU_DECL

// The kernel mode that is built for (variant, big) given what the caller would like (0 samples + one guided queue,
// 1 latency, 2 pixel): the reference-form sweep only exists in mode 0, the diagnostic builds follow the latency
// mode, big scenes have no latency mode.
extern "C" int r1_trace_mode(int variant, int big, int wanted)
{
    if (variant == 1)
        return 0;
    if (variant == 3 || variant == 5)
        return big ? 0 : 1;
    if (wanted == 1)
        return big ? 0 : 1;
    return wanted == 2 ? 2 : 0;
}

extern "C" hipError_t r1_launch_trace(const R1TraceArgs *args, int variant, int big_in, int mode, int blocks, hipStream_t stream)
{
    // dynamic LDS of the tree kernels: the traversal stack, one entry per inner node on a path, and (small scenes) the node table
    const bool big = big_in != 0; // 32-bit hit indices, attenuation stack in the global workspace
    const bool tree = variant == 4 || variant == 5;
    const size_t trav = tree ? (size_t)args->bvh_depth * R1_BLOCK * (big ? sizeof(uint32_t) : sizeof(uint16_t)) + (size_t)args->bvh_lds_f4 * 16 + R1_ENTRY_LDS_BYTES(args->entry_lds) : 0;
    if (mode != r1_trace_mode(variant, big_in, mode))
        return hipErrorInvalidValue; // the caller sizes its arguments by the mode: it must be the one that is built
    const int batch = args->batch != nullptr; // frame batches: the MODE 3 build of the throughput kernels (variants 2 and 4 only)
    if (batch && (mode != 0 || (variant != 2 && variant != 4)))
        return hipErrorInvalidValue;
    if (variant == 3 && big)
        variant = 2; // (no diagnostic build of the LDS-tiled sweep)
    // the throughput builds of the product kernels sum their tiles themselves (R1_LAND): a launch through them says on how many XCDs
    const bool land_kernel = variant == 4 && R1_LAND_MODE(mode);
    if (land_kernel != (args->land_res > 0u))
        return hipErrorInvalidValue;
    // (the tree kernels look a primary ray's entry node up in args->bvh_entry whenever the tree has a root step: never launch them without)
    if (tree && R1_ENTRY_MODE(mode) && args->scene.bvh_root_leaf != 0u && args->bvh_entry == nullptr)
        return hipErrorInvalidValue;
    if (tree)
        return big ? r1_tu_tree_big_launch(args, variant, mode, batch, blocks, trav, stream) : r1_tu_tree_small_launch(args, variant, mode, batch, blocks, trav, stream);
    return big ? r1_tu_sweep_big_launch(args, variant, mode, batch, blocks, 0, stream) : r1_tu_sweep_small_launch(args, variant, mode, batch, blocks, 0, stream);
}

extern "C" hipError_t r1_trace_occupancy(int variant, int big, int mode, size_t dyn_lds, int *blocks_per_cu)
{
    const bool tree = variant == 4 || variant == 5;
    if (variant == 3 && big)
        variant = 2;
    if (tree)
        return big ? r1_tu_tree_big_occupancy(variant, mode, dyn_lds, blocks_per_cu) : r1_tu_tree_small_occupancy(variant, mode, dyn_lds, blocks_per_cu);
    return big ? r1_tu_sweep_big_occupancy(variant, mode, 0, blocks_per_cu) : r1_tu_sweep_small_occupancy(variant, mode, 0, blocks_per_cu);
}

// generate + (max_bounces + 1) x (intersect, shade); every launch reads its queue length on the device
extern "C" hipError_t r1_launch_wavefront(R1WaveArgs *w, int blocks, hipStream_t stream)
{
    const size_t trav = (size_t)w->t.bvh_depth * R1_BLOCK * sizeof(uint32_t);
    hipLaunchKernelGGL(r1_wf_generate, dim3(blocks), dim3(R1_BLOCK), 0, stream, *w);
    for (int level = 0; level <= w->t.max_bounces; ++level)
    {
        w->level = level;
        hipLaunchKernelGGL(r1_wf_intersect, dim3(blocks), dim3(R1_BLOCK), trav, stream, *w);
        hipLaunchKernelGGL(r1_wf_shade, dim3(blocks), dim3(R1_BLOCK), 0, stream, *w);
    }
    return hipGetLastError();
}

// max_rows: at most this many rows of workgroups (one row walks tiles row, row + rows, ...); 0 = one row per tile.  Frames in flight use
// FEW, long-lived workgroups: a resolve launch shares the chip with persistent trace workgroups and gets a slot only when one of them
// exits, so what it costs is the number of slot grants it needs, not its 26 us of work (profiles/r03/burst_timeline_20_frames.txt).
extern "C" hipError_t r1_launch_resolve(const R1ResolveArgs *args, int max_rows, hipStream_t stream)
{
    const int tile_pix = args->tile_w * args->tile_h;
    const int bx = (tile_pix + 255) / 256;
    const uint32_t tiles_all = args->n_local_tiles * (args->n_frames ? args->n_frames : 1u);
    int by = (int)(tiles_all < 65535u ? tiles_all : 65535u);
    if (max_rows > 0 && by > max_rows)
        by = max_rows;
    hipLaunchKernelGGL(r1_resolve_kernel, dim3(bx, by), dim3(256), 0, stream, *args);
    if (args->frame_rays) // frame batches: per-frame ray counts from the launch's partial sums ([tile of the batch][bx])
        hipLaunchKernelGGL(r1_batch_counts_kernel, dim3(args->n_frames), dim3(256), 0, stream, args->frame_rays, args->n_local_tiles * (uint32_t)bx,
                           args->out, args->out_stride, args->rays_offset);
    return hipGetLastError();
}

// blocks: gathered tile blocks or records; per frame f (0 .. n_frames - 1): shard sh's block at blocks + f * frame_in + sh * shard_stride,
// image at rgb + f * frame_out.  want_total: the shards' uint64 counts at (block start + total_offset) are summed into the
// uint64 at (the frame's image + total_out).
extern "C" hipError_t r1_launch_assemble(const void *blocks, void *rgb, int width, int height, int tile_w, int tile_h, int tiles_x, int num_shards,
                                         size_t shard_stride, int n_frames, size_t frame_in, size_t frame_out, size_t total_offset, long long total_out,
                                         int want_total, hipStream_t stream)
{
    const size_t n = (size_t)width * height;
    int grid = (int)((n + 255) / 256);
    if (grid > 8192)
        grid = 8192;
    hipLaunchKernelGGL(r1_assemble_kernel, dim3(grid, n_frames > 0 ? n_frames : 1), dim3(256), 0, stream, (const uint8_t *)blocks, (uint8_t *)rgb, width,
                       height, tile_w, tile_h, tiles_x, num_shards, shard_stride, frame_in, frame_out, total_offset, total_out, want_total);
    return hipGetLastError();
}
