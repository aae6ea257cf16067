/* rays1.h — C-ABI of librays1.so, the MI355X-native replacement for the per-pixel /
 * per-sample path-tracing hot path of montib/rays1bench `src/step13`.
 *
 * The reference has no FFI or plugin interface: its boundary for this path is three
 * scene builders and one function in a single translation unit
 *     Scene *create_small_scene()/create_medium_scene()/create_large_scene()
 *                                      src/step13/rayweek1.cpp:552 / :582 / :654
 *     RESULT benchmark(Scene*, Pix*, bool write_tga, const char *scene_name)
 *                                      src/step13/rayweek1.cpp:845
 * This header is what a maintainer would bind from that function (INTEGRATION.md shows
 * the patch): plain C, POD structs, caller-owned memory, integer return codes, no
 * exceptions, no C++/torch types.  `rays1bench_amd/csrc/rayweek1_hip.cpp` is the
 * drop-in host program built on top of it (same entry points, CLI and output files).
 *
 * Threading: one call at a time per context (the reference calls benchmark() serially
 * from main, rayweek1.cpp:969-984).  All functions return R1_OK (0) or a negative
 * R1_E* code; r1_last_error() describes the last failure on the calling thread.
 * There is NO CPU fallback: without a usable HIP device every compute entry point
 * fails with R1_ENODEVICE.
 *
 * Environment variables.  None.  The shipped library never reads the environment: what it computes and how it
 * launches depends on its arguments only.  (A separate build with -DR1_TUNING, `make -C rays1bench_amd/csrc tuning`
 * -> lib/librays1_tuning.so, reads R1_* launch-shape and index-layout knobs for the measurements under tools/ and
 * profiles/; it is loaded explicitly by those scripts, never by the product or its tests, and every knob only chooses
 * among forms that produce the same pixels and ray counts.)
 */
#ifndef RAYS1_H
#define RAYS1_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R1_ABI_VERSION 4

enum
{
    R1_OK = 0,
    R1_EINVAL = -1,    /* bad argument (null pointer, non-positive size, bad sharding) */
    R1_ENODEVICE = -2, /* no HIP device / HIP runtime unusable                        */
    R1_EHIP = -3,      /* a HIP call failed; see r1_last_error()                      */
    R1_ENOMEM = -4,    /* device or host allocation failed                            */
    R1_ELIMIT = -5     /* scene / image exceeds a documented limit                    */
};

/* Material classes of the reference (rayweek1.cpp:396-511), flattened. */
enum
{
    R1_MAT_LAMBERTIAN = 0, /* albedo                              rayweek1.cpp:396-412 */
    R1_MAT_METAL = 1,      /* albedo, param = fuzz (<= 1)         rayweek1.cpp:419-436 */
    R1_MAT_DIELECTRIC = 2, /* param = refraction index            rayweek1.cpp:461-511 */
    R1_MAT_NONE = 255      /* placeholder sphere (material == nullptr, rayweek1.cpp:574-576) */
};

/* Mirror of SphereSOA::InstanceData (src/step13/soa_sphere.h:38-53) with the
 * polymorphic `Material*` column replaced by a flat material table.  `count` includes
 * the reference's padding placeholders (centre 1e9, inv_radius 0); spheres with
 * inv_radius == 0 (placeholders AND non-positive radii, soa_sphere.cpp:81) are never
 * hit, exactly as rayweek1.cpp:291 (and so are spheres with a non-finite centre or
 * radius_sq, which the reference's arithmetic can never hit).  `radius_sq[i]` and
 * `inv_radius[i]` must belong to the same radius, as SphereSOA::add stores them
 * (soa_sphere.cpp:70-85).  All arrays are host memory, length `count`, owned by the
 * caller and only read during the call they are passed to. */
typedef struct r1_scene
{
    uint32_t count;
    const float *center_x;
    const float *center_y;
    const float *center_z;
    const float *radius_sq;
    const float *inv_radius;
    const uint8_t *mat_type; /* R1_MAT_* */
    const float *albedo_r;
    const float *albedo_g;
    const float *albedo_b;
    const float *mat_param; /* fuzz (metal) or refraction index (dielectric) */
} r1_scene;

/* Mirror of struct Camera after Camera::init (rayweek1.cpp:364-395). */
typedef struct r1_camera
{
    float origin[3];
    float lower_left[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float w[3];
    float lens_radius;
} r1_camera;

/* Runtime replacements for the reference's compile-time configuration
 * (src/common/common.h:19-28: SCREEN_W, SCREEN_H, NUM_SAMPLES_PER_PIXEL, MAX_BOUNCES)
 * plus the build-defined seeding contract (include/rays1_seed.h) and sharding. */
typedef struct r1_params
{
    int32_t width;       /* SCREEN_W                                              */
    int32_t height;      /* SCREEN_H                                              */
    int32_t spp;         /* NUM_SAMPLES_PER_PIXEL                                 */
    int32_t max_bounces; /* MAX_BOUNCES (reference: 50); 1..R1_MAX_BOUNCES_LIMIT  */
    uint32_t seed;       /* frame seed of the per-sample seeding contract         */
    int32_t tile_w;      /* tile size for multi-device sharding (reference: 32)   */
    int32_t tile_h;
    int32_t shard;       /* this device renders tiles t with t % num_shards == shard */
    int32_t num_shards;  /* 1 = whole frame                                       */
    int32_t variant;     /* R1_VARIANT_*; 0 = default kernel                      */
} r1_params;

#define R1_MAX_BOUNCES_LIMIT 51

enum
{
    R1_VARIANT_DEFAULT = 0,   /* BVH for every scene (a property of the build: what a context launches depends on its
                                 arguments only; same pixels as every other variant; r1_launch_info.kernel says which ran) */
    R1_VARIANT_REFERENCE = 1, /* pass 1 in the reference's exact arithmetic (rayweek1.cpp:190-202),
                                 no prefilter; slower, used to cross-check the default       */
    R1_VARIANT_PREFILTER = 2, /* the exhaustive sweep (every ray against every sphere, as Hitable::hit does,
                                 rayweek1.cpp:182-322): conservative grouped prefilter + exact re-test */
    R1_VARIANT_STATS = 3,     /* PREFILTER plus in-kernel phase/utilisation counters (diagnostic; r1_last_stats) */
    R1_VARIANT_BVH = 4,       /* optional spatial index (the reference has none, README.md:163): a conservative
                                 box tree chooses the spheres given to the reference's per-sphere test; results
                                 are bit-identical to the exhaustive sweeps   */
    R1_VARIANT_BVH_STATS = 5, /* BVH plus traversal counters (diagnostic; r1_last_stats slots [2] node-loop trips,
                                 [3] leaf-loop trips, [5] sphere-pair tests, [9] node visits, [14] leaf trips x lanes, [15] root steps: the root's leaf and
                                 the box of its other child tested outside the walk's loops) */
    R1_VARIANT_WAVEFRONT = 6  /* the same tracer as separate generate / intersect / shade kernels with the paths
                                 and per-level queues in HBM; a comparison build: same pixels,
                                 slower than the megakernel; frames of <= 2^24 sample slots       */
};

typedef struct r1_context r1_context; /* opaque: device, stream, events, workspace */

/* ---- lifetime ------------------------------------------------------------------- */

/* Library ABI version (R1_ABI_VERSION of the build). */
int r1_abi_version(void);

/* Creates a context on HIP device `device` (stream, timing events, scene + workspace
 * buffers are cached in it so that the `-n` runs of rayweek1.cpp:969-984 do not pay
 * context creation inside the timed region).  *out is NULL on failure. */
int r1_create(int device, r1_context **out);
void r1_destroy(r1_context *ctx);

/* Text of the last error on this thread ("" if none). Never NULL. */
const char *r1_last_error(void);

/* Number of visible HIP devices, or a negative R1_E* code. */
int r1_device_count(void);

/* ---- the hot path ---------------------------------------------------------------- */

/* Uploads (and caches) the scene + camera in the context: builds the device-side
 * sphere table and material table.  Replaces what benchmark() receives as `Scene*`
 * (rayweek1.cpp:845, :851). */
int r1_set_scene(r1_context *ctx, const r1_scene *scene, const r1_camera *camera);

/* Renders the frame (or this shard's tiles of it) and returns everything on the host.
 * Replaces TileRenderScheduler::run + render_tile (rayweek1.cpp:785-842, :722-782).
 *   rgb_out      width*height*3 bytes, row-major, row 0 = bottom row, Pix{r,g,b}
 *                (common.h:80-83, rayweek1.cpp:750); with num_shards > 1 only this
 *                shard's tiles are written, other bytes are left untouched.
 *   num_rays_out number of color() invocations (rayweek1.cpp:517) of this shard.
 *   device_seconds_out  (optional) GPU time of the kernels, from HIP events.
 * The frame is ONE launch: the trace kernel resolves each 32 x 32 tile as soon as its samples are complete
 * (rayweek1.cpp:762-775 resolves a pixel where it traced it).  If rgb_out is page-locked memory (r1_host_alloc) the
 * tiles are stored straight into it and nothing is copied; any other memory receives one copy of the finished image. */
int r1_render(r1_context *ctx, const r1_params *params, uint8_t *rgb_out, uint64_t *num_rays_out,
              double *device_seconds_out);

/* Same, plus the per-sample results the resolve pass sums: samples_out receives
 * width*height*spp records of 4 floats {r, g, b, bit_cast<float>(uint32 rays)} in the
 * order ((y*width + x)*spp + s).  For parity tests; whole frame only (num_shards 1). */
int r1_render_samples(r1_context *ctx, const r1_params *params, uint8_t *rgb_out, uint64_t *num_rays_out,
                      float *samples_out);

/* Pipelined form of r1_render — frames in flight whose results land on the HOST.  The reference times
 * dispatch -> pixels + ray count on the host (rayweek1.cpp:848 -> :891) for ONE frame and waits; a caller that
 * renders frame after frame (main's `-n` runs, rayweek1.cpp:969-984) can keep several in flight instead: the call
 * enqueues the frame (throughput kernels: few long-lived waves per frame) on `hip_stream` (a hipStream_t; NULL =
 * the context's stream) and returns without waiting.  Buffers from r1_host_alloc (page-locked, num_rays_out 8-byte
 * aligned) are written by the launch itself: every tile lands in rgb_out (row-major, width*height*3 bytes, as
 * r1_render) when its samples are complete, the ray count when the frame's last tile has — no resolve launch, no copy.
 * Pageable memory works too: the frame is then resolved into the context's device image and two copies follow the
 * launch, each waiting for its frame.  Both buffers are valid once the stream is idle (r1_sync for the context's
 * stream).  One frame per context at a time (the calls on a context are ordered by one stream, or by the caller): K
 * frames in flight = K contexts.  Whole frames only (num_shards == 1).  rgb_out and num_rays_out both NULL: the frame is
 * rendered and left in the context's device buffers (a measurement aid). */
int r1_render_async(r1_context *ctx, const r1_params *params, uint8_t *rgb_out, uint64_t *num_rays_out, void *hip_stream);

/* Frame BATCHES: n_frames frames of the same scene, camera and size in ONE launch; frame f is seeded
 * params->seed + f * seed_stride (0: identical frames; 1: the passes of a progressive render).  The trace kernel is
 * persistent — a wave keeps refilling its lanes from a queue of samples — and what a launch costs beyond its samples is
 * its ramp and, above all, its drain: the last ~40 iterations of every wave run with few live lanes.  In a batch the
 * queue is frame-major and the waves flow from one frame into the next, so that cost is paid once per batch instead of
 * once per frame (measured: DESIGN.md §4.9).  The price is latency: all frames of a batch are delivered together.
 * host_frames receives n_frames frame records of r1_frame_record_bytes() each — the row-major image (as r1_render),
 * padded to a multiple of 8 bytes, then the frame's uint64 ray count — written by the launch itself when the memory is
 * page-locked (r1_host_alloc), else with ONE copy enqueued behind the launch; nothing
 * is waited for (as r1_render_async; NULL leaves the frames on the device).  Whole
 * frames only.  Each frame's pixels and count equal what r1_render returns for its seed. */
size_t r1_frame_record_bytes(const r1_params *params);
int r1_render_batch_async(r1_context *ctx, const r1_params *params, int32_t n_frames, uint32_t seed_stride, void *host_frames, void *hip_stream);

/* Page-locked, device-visible host memory for the render entry points' outputs (hipHostMalloc / hipHostFree). */
int r1_host_alloc(size_t bytes, void **out);
void r1_host_free(void *p);

/* ---- device-resident variants (multi-GPU gather, benchmarks) ----------------------- */

/* Number of tiles / bytes of the dense tile block one shard produces for `params`
 * (every shard's block is padded to the same size so it can be all-gathered). */
int r1_tile_count(const r1_params *params, int32_t *tiles_total, int32_t *tiles_per_shard);
size_t r1_shard_block_bytes(const r1_params *params);
/* Bytes of one shard's gather RECORD: its tile block padded to a multiple of 8, followed by its uint64 ray count
 * (at r1_shard_record_bytes() - 8, 8-byte aligned for any tile size), so that one all-gather moves pixels and
 * counts together (the reference sums `out_num_rays` after the join, rayweek1.cpp:809-813). */
size_t r1_shard_record_bytes(const r1_params *params);

/* Enqueues the render of this shard on the context's stream (or on `hip_stream` if
 * non-NULL, a hipStream_t) and writes DEVICE memory only:
 *   d_block      r1_shard_block_bytes() bytes: tiles_per_shard tiles of tile_h*tile_w*3
 *                bytes each, local tile j = global tile shard + j*num_shards
 *   d_num_rays   one uint64 (overwritten); must be 8-byte aligned (R1_EINVAL otherwise)
 * Does not synchronise.  Sized for throughput: meant to be called for several frames in
 * flight (one context + stream per frame in flight). */
int r1_render_shard_device(r1_context *ctx, const r1_params *params, void *d_block, void *d_num_rays, void *hip_stream);

/* A batch of n_frames frames of this shard in one launch (see r1_render_batch_async): d_records receives n_frames
 * records of r1_shard_record_bytes() each (dense tile block, padded to 8 bytes, + the shard's uint64 ray count of that
 * frame), 8-byte aligned device memory — what a rank hands to ONE all-gather per batch. */
int r1_render_shard_device_batch(r1_context *ctx, const r1_params *params, int32_t n_frames, uint32_t seed_stride, void *d_records, void *hip_stream);

/* PIXEL mode for r1_render_shard_device (off by default).  On: a lane of the trace kernel owns a pixel and runs
 * its spp samples one after the other, so the `col += color()` of rayweek1.cpp:762 happens in a register in the
 * reference's order and the resolved pixel (rayweek1.cpp:765-775) is all the frame writes — 3 bytes per pixel
 * instead of 16 bytes per sample, no resolve launch, no per-sample workspace (3.9 GB per frame in flight at
 * 1200x800x250).  Same pixels and ray counts; measured ~10 % slower on the reference's scenes (a frame's last
 * pixels are ten dependent samples long), which is why it is a choice.  The host-returning entry points and
 * r1_render_samples always keep per-sample records. */
int r1_set_pixel_mode(r1_context *ctx, int32_t on);

/* Same outputs, sized for latency instead: ONE frame whose result the caller waits for (the full
 * persistent grid and the latency-mode kernels of r1_render).  Used by r1_multi_render. */
int r1_render_shard_device_once(r1_context *ctx, const r1_params *params, void *d_block, void *d_num_rays, void *hip_stream);

/* Scatters `num_shards` gathered blocks (concatenated in shard order, device memory)
 * into the row-major image d_rgb (width*height*3 bytes, device memory). */
int r1_assemble_device(r1_context *ctx, const r1_params *params, const void *d_blocks, void *d_rgb, void *hip_stream);

/* Same, for blocks that are `shard_stride_bytes` apart (>= r1_shard_block_bytes; 0 = tight):
 * lets a caller append per-shard trailers (e.g. the 8-byte ray count) to the gathered records so
 * that one all-gather moves pixels and counts together. */
int r1_assemble_device_strided(r1_context *ctx, const r1_params *params, const void *d_blocks, size_t shard_stride_bytes,
                               void *d_rgb, void *hip_stream);

/* Same for gathered RECORDS (num_shards x r1_shard_record_bytes(), the layout one all-gather returns): scatters the
 * tile blocks into d_rgb and writes the sum of the shards' ray counts — the join of rayweek1.cpp:809-813 — to
 * *d_total_rays (device memory, 8-byte aligned; e.g. right behind the image, so that one copy brings both home). */
int r1_assemble_device_records(r1_context *ctx, const r1_params *params, const void *d_records, void *d_rgb, void *d_total_rays,
                               void *hip_stream);

/* Batches: d_gathered = the result of ONE all-gather of every shard's n_frames records, [shard][frame][record];
 * d_frames receives n_frames frame records (r1_frame_record_bytes each: image + summed ray count). */
int r1_assemble_device_records_batch(r1_context *ctx, const r1_params *params, int32_t n_frames, const void *d_gathered, void *d_frames,
                                     void *hip_stream);

/* Blocks until the context's stream is idle. */
int r1_sync(r1_context *ctx);

/* HIP-event duration (ms) of the trace kernel / of all kernels of the last render
 * enqueued through this context (valid after r1_sync or a host-returning call). */
int r1_last_timing(r1_context *ctx, double *trace_kernel_ms, double *total_ms);

/* Per-frame kernel timing over a run of frames: between r1_timing_begin and r1_timing_end
 * every render enqueued through the context records its own HIP events (on the stream it
 * is launched on) instead of the single "last frame" set; r1_timing_end waits for them and
 * returns the summed trace-kernel and trace+resolve durations (ms) and the frame count.
 * Frames beyond max_frames reuse the last slot. */
int r1_timing_begin(r1_context *ctx, int32_t max_frames);
int r1_timing_end(r1_context *ctx, double *trace_ms_sum, double *total_ms_sum, int32_t *frames);

/* Diagnostic counters of the last R1_VARIANT_STATS render through the context's own stream:
 * 16 uint64: [0] wave iterations, [1] alive lanes summed over iterations, [2] candidate-loop
 * trips, [3] lanes that overflowed the candidate list, [4..7] cycles in refill / pass 1 /
 * candidate re-test / shade, [8] wave cycles, [9] candidates, [14] (tree) leaf trips summed over lanes, [15] (tree) root steps. */
int r1_last_stats(r1_context *ctx, uint64_t *out16);

/* Per-wave log of the last R1_VARIANT_*_STATS render: 4 uint64 per wave {start, sample queue found
 * empty, end (100 MHz device clock), outer iterations}.  out == NULL only returns the wave count.  Diagnostic. */
int r1_last_wave_log(r1_context *ctx, uint64_t *out, size_t cap_waves, uint32_t *waves);

/* Launch geometry and occupancy facts of the last render (for reports). */
typedef struct r1_launch_info
{
    int32_t compute_units;
    int32_t blocks;
    int32_t threads_per_block;
    int32_t spheres_active; /* spheres with inv_radius != 0 that the sweep visits */
    int32_t spheres_padded; /* `count` of the scene (N_pad of the reference)     */
    int32_t groups;         /* sphere groups the first sweep level tests          */
    uint64_t samples;       /* pixel-samples traced                              */
    int32_t kernel;         /* R1_VARIANT_* the launch actually ran (DEFAULT resolved) */
    int32_t bvh_nodes;      /* inner nodes of the spatial index                   */
    int32_t bvh_leaves;
    int32_t bvh_depth;      /* inner nodes on the longest root-to-leaf path       */
    int32_t tiles_in_kernel; /* 1: the trace launch summed its tiles itself (frames in flight); 0: a resolve launch followed */
} r1_launch_info;
int r1_last_launch_info(r1_context *ctx, r1_launch_info *out);

/* ---- one process, N GPUs: tile split + one RCCL all-gather per frame --------------------- */

/* The multi-GPU form of benchmark()'s join (rayweek1.cpp:804-813, :773-775) behind ONE call, so that
 * the reference's four-argument benchmark() stays a single function: tile t is
 * rendered by device t % N (r1_params.shard / num_shards are set by the library), every device's
 * record (dense tile block + its uint64 ray count) goes through one ncclAllGather over xGMI, device 0
 * assembles the row-major image and copies it to the host once.  RCCL is loaded at run time
 * (librccl.so.1); r1_multi_create fails with R1_ENODEVICE where it is missing — no fallback.
 * `devices` = N distinct HIP device ordinals (NULL: 0..N-1).  One call at a time per r1_multi. */
typedef struct r1_multi r1_multi;
int r1_multi_create(int32_t n_devices, const int32_t *devices, r1_multi **out);
void r1_multi_destroy(r1_multi *m);
int r1_multi_set_scene(r1_multi *m, const r1_scene *scene, const r1_camera *camera);
/* rgb_out / num_rays_out as r1_render (whole frame); device_seconds_out (optional): render + gather
 * on the slowest device, from HIP events. */
int r1_multi_render(r1_multi *m, const r1_params *params, uint8_t *rgb_out, uint64_t *num_rays_out, double *device_seconds_out);
/* Frames in flight across the N GPUs from one process: the enqueue-only form of r1_multi_render (throughput kernels, this
 * object's own all-gather, device 0 assembles image + summed count into ONE frame record — r1_frame_record_bytes() — and copies
 * it into `host_frame`, page-locked memory).  Nothing is waited for; r1_multi_sync waits for this object's frame.  A caller
 * keeps K frames in flight with K r1_multi objects (each owns a communicator), as K r1_contexts do on one GPU. */
int r1_multi_render_async(r1_multi *m, const r1_params *params, void *host_frame);
/* The same for a batch of n_frames frames (seeds params->seed + f * seed_stride) per launch, all-gather and copy (see
 * r1_render_batch_async): host_frames receives n_frames frame records. */
int r1_multi_render_batch_async(r1_multi *m, const r1_params *params, int32_t n_frames, uint32_t seed_stride, void *host_frames);
int r1_multi_sync(r1_multi *m);
/* Where everything lies in the buffers of an n_devices-device frame or batch of n_frames frames: pure arithmetic (no device, no
 * r1_multi object), the very function the r1_multi_* entry points size and address their buffers with; so the layout of an N-GPU
 * run can be checked on a machine with one GPU or none (the join it describes: rayweek1.cpp:869-877, :809-813). */
typedef struct r1_multi_layout_info
{
    size_t block_bytes;        /* a device's dense tile block of ONE frame (r1_shard_block_bytes) */
    size_t record_bytes;       /* the block padded to 8 bytes + the device's uint64 ray count (r1_shard_record_bytes) */
    size_t count_offset;       /* the count's offset in a record */
    size_t send_bytes;         /* what a device hands to the all-gather: its n_frames records */
    size_t gathered_bytes;     /* what it receives: [device][frame][record] */
    size_t frame_record_bytes; /* an assembled frame: row-major image padded to 8 bytes + the frame's uint64 count (r1_frame_record_bytes) */
    size_t frame_count_offset;
    size_t host_bytes;         /* the one copy to the host: n_frames frame records */
    size_t counts_pitch;       /* distance between two devices' records in the gathered buffer */
} r1_multi_layout_info;
int r1_multi_layout(const r1_params *params, int32_t n_devices, int32_t n_frames, r1_multi_layout_info *out);
/* Facts for reports: device count, RCCL version code (ncclGetVersion), launch info of the first device. */
int r1_multi_info(r1_multi *m, int32_t *n_devices, int32_t *rccl_version, r1_launch_info *first_device);

/* ---- host-side helpers of the drop-in (no GPU needed) ------------------------------ */

/* Shape of the spatial index R1_VARIANT_BVH uses (r1_bvh.cpp; the reference has no such
 * structure, README.md:163).  Builds it on the host exactly as r1_set_scene does.  Optional
 * outputs: `nodes_out` receives 16 floats per inner node {m0x m1x m0y m1y | m0z m1z e0x e1x |
 * e0y e1y e0z e1z | A K child0 child1} (child i: box centre m_i, half extent e_i, both inflated
 * per ray by A |o - info.centre|^2 + K, or A |m_0 + m_1 - 2 o|^2 + K if info.pad_local; reference:
 * bit 31 = leaf, then bits 28..30 = number of sphere PAIRS and bits 0..27 the first pair; else
 * inner node index), `ids_out[2 * pair + {0, 1}]` the scene index of each leaf sphere
 * (0xFFFFFFFF = the empty partner of an odd sphere); needs 2 * info.pairs entries.
 * leaf_max <= 0 selects the build's default spheres per leaf. */
typedef struct r1_bvh_info
{
    int32_t nodes, leaves, depth, stack_entries, spheres, pairs;
    float centre[3];   /* C of the kernel's box inflation pad = A |o - C|^2 + K (r1_bvh.cpp) */
    int32_t pad_local; /* 1: the tree uses pad = A |m0 + m1 - 2 o|^2 + K instead (scenes of small spheres) */
    int32_t root_leaf; /* 1 / 2: child 0 / 1 of the root is a leaf of <= 2 sphere pairs and the other child an inner node — the kernels test
                          that leaf and the other child's box once per ray outside the walk's loops; 0: the root has no such shape */
} r1_bvh_info;
int r1_bvh_describe(const r1_scene *scene, int32_t leaf_max, r1_bvh_info *info, float *nodes_out, size_t nodes_cap, uint32_t *ids_out,
                    size_t ids_cap);

enum
{
    R1_SCENE_SMALL = 0,  /* create_small_scene   rayweek1.cpp:552 */
    R1_SCENE_MEDIUM = 1, /* create_medium_scene  rayweek1.cpp:582 */
    R1_SCENE_LARGE = 2,  /* create_large_scene   rayweek1.cpp:654 */
    R1_SCENE_GRID = 3    /* build-defined: the large generator scaled to grid_w x grid_h spheres */
};

typedef struct r1_host_scene r1_host_scene; /* owns the arrays an r1_scene points to */

/* Builds one of the reference scenes for an image of width x height (the aspect the
 * reference takes from SCREEN_W/SCREEN_H, rayweek1.cpp:564).  grid_w/grid_h are only
 * used by R1_SCENE_GRID (0 = the reference's 30 x 16). */
int r1_host_scene_create(int kind, int32_t width, int32_t height, int32_t grid_w, int32_t grid_h, r1_host_scene **out);
void r1_host_scene_destroy(r1_host_scene *hs);
const r1_scene *r1_host_scene_spheres(const r1_host_scene *hs);
const r1_camera *r1_host_scene_camera(const r1_host_scene *hs);

/* tga_write_rgb24 (common.h:86-122): writes a 24-bit TGA and, like the reference,
 * leaves `pixels` with R and B swapped.  Returns R1_OK or R1_EINVAL if the file
 * cannot be opened. */
int r1_tga_write_rgb24(const char *filename, int32_t width, int32_t height, uint8_t *pixels);

/* log_results (common.h:47-77): writes out_<scene>.txt as
 * `version|%.3fs|%llu|%0.3f mrays/s|` averaged over the runs. */
int r1_log_results(const char *version, const char *scene, const double *elapsed_seconds, const uint64_t *num_rays,
                   int32_t num_runs);

#ifdef __cplusplus
}
#endif

#endif /* RAYS1_H */
