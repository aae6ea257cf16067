// r1_device.h — device-side data layout shared by r1_kernels.hip and r1_capi.cpp.
#ifndef R1_DEVICE_H
#define R1_DEVICE_H

#include <stdint.h>

// Launch geometry of the trace kernel.
#define R1_BLOCK 256        // threads per workgroup = 4 wave64
#define R1_CAND_CAP 16      // per-lane candidate slots in LDS (flushed when a lane passes CAP-8)
#define R1_PAIR_CAP 1024    // (lane, sphere) pairs of one wave: 64 lanes x R1_CAND_CAP
#define R1_STACK_WORDS 17   // ceil(51 / 3) packed 10-bit hit indices per lane (max_bounces <= 51)
#define R1_CHUNK 256        // samples a wave takes from the global queue per atomic
#define R1_MAX_ACTIVE_10BIT 1023

// Everything the trace kernel needs; passed by value (kernarg segment => SGPRs).
struct R1DeviceScene
{
    // Prefilter table over the ACTIVE spheres (inv_radius != 0), 8 floats per PAIR of spheres
    // {cx0 cx1 cy0 cy1 cz0 cz1 Kp0 Kp1}, Kp = |c|^2 - r^2 - slack; padded with never-candidate
    // entries (Kp = +inf) to a multiple of 16 spheres PLUS one extra chunk of 8 (prefetch target).
    const float4 *sweep;
    // Exact table, same indexing: {cx, cy, cz, radius_sq} and {inv_radius, albedo rgb},
    // {type, param}.
    const float4 *exact;
    const float4 *shade;   // {inv_radius, albedo_r, albedo_g, albedo_b}
    const float2 *mat;     // {bit_cast<float>(type), param}
    uint32_t n_active;     // real entries
    uint32_t n_sweep;      // padded to a multiple of 16 (the table holds 8 more for the prefetch)
};

struct R1DeviceCamera
{
    float origin[3], lower_left[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius;
};

struct R1TraceArgs
{
    R1DeviceScene scene;
    R1DeviceCamera cam;
    int32_t width, height, spp, max_bounces;
    uint32_t seed;
    float inv_w, inv_h;          // 1.0f / width, 1.0f / height (IEEE divisions done on the host)
    int32_t tile_w, tile_h, tiles_x;
    int32_t shard, num_shards;
    uint32_t n_local_tiles;      // tiles this shard owns
    const uint32_t *tile_sample_base; // [n_local_tiles + 1] prefix sums of samples per local tile
    uint32_t total_samples;      // tile_sample_base[n_local_tiles]
    uint32_t *queue;             // global sample counter (zeroed before the launch)
    float4 *samples;             // [total_samples] {r, g, b, bit_cast<float>(rays)}
    unsigned long long *num_rays; // accumulated color() invocations
    unsigned long long *stats;    // diagnostic counters (R1_VARIANT_STATS builds only), else null
};

struct R1ResolveArgs
{
    const float4 *samples;
    const uint32_t *tile_sample_base;
    int32_t width, height, spp;
    int32_t tile_w, tile_h, tiles_x;
    int32_t shard, num_shards;
    uint32_t n_local_tiles;
    float inv_spp;               // (float)(1.0f / spp)
    uint8_t *out;                // row-major image or dense tile block
    int32_t block_layout;        // 0: row-major width*height*3, 1: dense tile block
};

#endif
