// r1_device.h — device-side data layout shared by r1_trace.hpp and r1_capi.cpp.
#ifndef R1_DEVICE_H
#define R1_DEVICE_H

#include <stdint.h>

// Launch geometry of the trace kernel.
#define R1_BLOCK 256        // threads per workgroup = 4 wave64
#define R1_BIT_WORDS 8      // small scenes: per-lane flag words in LDS, 32 groups each, worked off 256 groups at a time
#define R1_CAND_CAP 16      // big scenes: per-lane flagged-group slots in LDS (flushed when a lane passes CAP-8)
#define R1_PAIR_CAP 1024    // (lane, sphere) pairs of one wave, big scenes: 64 lanes x R1_CAND_CAP
#define R1_PAIR_CAP_SMALL 512 // same, small scenes (bit path): 4 KB less LDS per workgroup = a fifth workgroup per CU
#define R1_STACK_WORDS 17   // ceil(51 / 3) packed 10-bit hit indices per lane (max_bounces <= 51)
#ifndef R1_STACK_LDS_WORDS
#define R1_STACK_LDS_WORDS 10 // tree kernel, small scenes: words of the packed stack kept in LDS (30 entries); deeper entries (rare) go to
#endif                        // the global workspace.  10 KB instead of 17: with the 128-node table (8 KB) and the 16-bit traversal stack (4 KB,
                              // round 3) a workgroup holds 22 KB = SEVEN workgroups per CU (23.4 KB each at most; six with round 2's 32-bit
                              // traversal stack); 4 / 6 / 8 / 10 / 12 words at six workgroups: 28.4 / 29.1 / 29.3 (32.3) / (32.5) / 28.2 Grays/s
#ifndef R1_STACK_LDS_WORDS_TP
#define R1_STACK_LDS_WORDS_TP 10 // the same for the THROUGHPUT builds of that kernel (MODE 0 / 3), kept apart for experiments: 6 words with the 32-bit
#endif                           // traversal stack was the first way to seven workgroups (+3 % from the wave, -1.7 % from the deeper global overflow)
#define R1_RESOLVE_ROWS_TP 256 // frames in flight: rows of workgroups of the resolve launch (0 = one per tile; 16 / 64 / 256 / one per tile: 29.4 / 30.6 / 31.3 / 30.7 Grays/s over 20 steps, 34.9 / 34.85 / 34.87 / 34.7 over 300), see r1_launch_resolve
#define R1_NODES_LDS_MAX 256  // tree kernel: scenes whose tree has at most this many nodes (16 KB) run the small-scene kernels, which keep
                              // the node table in LDS; bigger trees run the big-scene kernels (node table through the vector L1)
#define R1_CHUNK 256        // most samples a wave takes from the global queue per atomic (small frames) ...
#define R1_CHUNK_BIG 1024   // ... growing with a wave's share of the frame up to this (enqueue_frame)
#define R1_CHUNK_MIN 32      // fewest (end of the queue: guided self-scheduling)
#define R1_GROUP_MAX 4         // spheres per group (level 1 of the sweep tests group bounds)
#define R1_GROUP_MIN_SPHERES 128 // scenes with fewer active spheres are swept ungrouped
#define R1_GROUP_RATIO 3.5     // a group's bounding radius stays within this factor of its smallest member radius
#define R1_SAMPLES_PER_LANE 125    // throughput mode grid sizing: samples each lane should get (see enqueue_frame): 1200x800x10 -> 300
                                   // workgroups per frame.  Re-tuned after DESIGN §4.13 (a cheaper refill shifts the balance towards fewer,
                                   // longer-lived workgroups; tools/spl_sweep2.sh, three alternating rounds): 100 / 115 / 125 / 135 / 150 samples
                                   // -> 36.46 / 36.75 / 36.90 / 37.00 / 37.3 Grays/s over 300 steps and 32.7 / 33.3 / 33.4 / 32.7 / 32.0 over the
                                   // driver's 20 (a burst drains better with more, shorter-lived workgroups; a long run loses to their drains) ...
#define R1_SAMPLES_PER_LANE_SHARD 100 // ... a rank's share of a frame (num_shards > 1) keeps 100: 0.1054 against 0.1069 ms per frame for an
                                   // emulated rank of eight ...
#define R1_MIN_BLOCKS 128          // ... but at least this many workgroups per frame (the per-rank frames of a 4- or 8-GPU run:
                                   // 0.301 ms per frame against 0.329 with 256) ...
#define R1_SAMPLES_PER_LANE_MIN 32 // ... as long as a lane still gets this many samples
#define R1_TILE_SPHERES 512    // big-scene sweep: spheres per LDS tile (8 KB in pair layout), two tiles in LDS
#define R1_TILE_F4 (R1_TILE_SPHERES / 2 * 2) // float4 per tile: 2 per pair of spheres
#define R1_MAX_ACTIVE_10BIT 1023
#define R1_MAX_ACTIVE (1u << 21) // big-scene kernels: 26-bit pair indices, limit kept at 2 M spheres
#define R1_STACK_ENTRIES 51
#define R1_BVH_STACK 32        // R1_VARIANT_BVH: per-lane traversal stack entries in LDS = most inner nodes on a path
#define R1_BVH_TOP_NODES 63    // box tree: the first nodes in breadth-first order (6 levels); the big-scene kernels keep them in LDS (4 KB: 0 / 63 / 127 / 255 nodes -> 12.9 / 13.3 / 13.0 / 12.6 Grays/s on 100 004 spheres, the larger copies cost workgroups per CU)
#define R1_BVH_LEAF 4          // spheres per leaf (<= 14; stored as pairs)
#define R1_SUBQUEUES 16        // latency mode: sub-queues of the sample queue (R1TraceArgs::nq)
#define R1_COUNTER_BYTES 4096  // per-context counter block: queue heads, ray count (+32), drain counts, stats (+128), sub-queues at +1024;
                               // the allocation carries 64 more bytes: the published ray count of the synchronous entry points
#define R1_COOP_LANES 2        // R1TraceArgs::coop_lanes (one synchronous frame: 2 -> 1.144 ms, 4 -> 1.160, 8 -> 1.193, 16 -> 1.263)
// Tiles resolved INSIDE the trace kernel (round 4; r1_trace.hpp has the protocol): the throughput kernels sum every
// 32 x 32 tile themselves, on the XCD that traced it, and store the pixels where the frame is wanted — device memory or the caller's
// page-locked host buffer.  No resolve launch, no copy.  0 builds the round-3 form (r1_resolve_kernel after the trace kernel) for A/B.
#ifndef R1_LAND
#define R1_LAND 1
#endif
// per-tile entry nodes: a PRIMARY ray starts its walk below the root's inner child at the deepest node all primary rays of
// its 32 x 32 tile stay under (computed on the host per camera and tiling, r1_capi.cpp compute_entries); 0: every walk starts at that child.
// Built, bit-identical, measured and NOT adopted (round 4): 9 % fewer node visits per ray (5.41 -> 4.91 on large 1200x800x10; 42 % of the
// tiles' primary rays are done after the root step) but 5 % more wave iterations — the primary rays' shorter walks do not shorten the LONGEST
// walk of a wave, which sets the trips of the divergent loops — and 8 % (table in global memory: a vector-memory wait in the root step) to 12 %
// (table in LDS: six workgroups per CU instead of seven) FEWER Grays/s (profiles/r04/entry_nodes_ab.txt).
// `make tuning EXTRA=-DR1_ENTRY=1` builds it; tools/entry_ab.sh runs the parity tests, tools/walk_ab*.sh the A/B.
#ifndef R1_ENTRY
#define R1_ENTRY 0
#endif
#define R1_ENTRY_LDS_MAX 2048u // tiles per launch up to which the small-scene kernels keep the entry table in LDS (4 KB)
#define R1_ENTRY_LDS_BYTES(n) ((((size_t)(n) * 2u) + 15u) & ~(size_t)15u)
#define R1_ENTRY_MODE(mode) (R1_ENTRY && ((mode) == 0 || (mode) == 3 || (mode) == 1))
// 4-wide nodes for the small-scene tree kernels: the binary tree collapsed on the host (a node takes
// its grandchildren until it holds four children), seven float4 per node in the workgroup's LDS table, four box tests per trip, the hits
// pushed far-to-near.  An A/B build: make tuning EXTRA=-DR1_BVH4=1.
// exhaustive sweep, exact phase (exact_trips): member spheres from a table in group order (one global fetch in the dependent chain instead
// of two), and the wave's rays as two 16-byte LDS reads per slot instead of six 4-byte ones
#ifndef R1_EXACT_G
#define R1_EXACT_G 1
#endif
#ifndef R1_RAYS_AOS
#define R1_RAYS_AOS 0
#endif
#ifndef R1_SWEEP_PAIRS2
#define R1_SWEEP_PAIRS2 1 // exhaustive sweep, group test: two pairs of groups interleaved (sweep_prefilter)
#endif
#ifndef R1_BVH4
#define R1_BVH4 0
#endif
#if R1_BVH4 && R1_ENTRY
#error "R1_ENTRY's table holds references into the binary tree: not with R1_BVH4"
#endif
#ifndef R1_LAND_SYNC
#define R1_LAND_SYNC 0 // 1: the synchronous frame's kernels (MODE 1) too — measured and not adopted: 1.276 against 1.077 ms on the device (the tiles a wave owes are summed at ITS exit, i.e. at the end of the frame's critical path; the resolve launch sums all 950 in 26 us with the whole chip), profiles/r04/land_sync_frame.txt
#endif
#define R1_LAND_MODE(mode) (R1_LAND && ((mode) == 0 || (mode) == 3 || (R1_LAND_SYNC && (mode) == 1)))
#define R1_LAND_CNT_STRIDE 32u  // words between two tiles' countdowns: every countdown on its own 128-byte line (an atomic on ONE line sustains ~88 M/s on this chip,
                               // tools/ubench_atomic.hip; the ~40 tiles a synchronous frame's waves work on at a time shared two lines at first: 3.8 ms per frame instead of 1.1)
#define R1_LAND_MAX_WAIT (1u << 16) // passes over a claimed tile that still find a record of an earlier launch before the wave gives up and flags the launch
                                    // (a record's store is on its way for microseconds; a bug would otherwise hang the device instead of failing the call)
#define R1_COUNTER_TAIL 4096   // behind the R1_COUNTER_BYTES block: +0 published ray count, +64 batch-argument slots (8 x 32 B), +1024 the second
                               // set of queue heads (frames alternate between the sets; the resolvers of a launch zero the set the
                               // launch before it used), then per launch: frame accumulators and the per-tile countdowns

// Tuning knobs.  The SHIPPED library never reads the environment: what it does depends on its arguments only.  A build made
// with -DR1_TUNING (`make tuning` -> lib/librays1_tuning.so; the scripts under tools/ load it explicitly) reads the R1_*
// variables named at the call sites, once per process — launch shapes and conservative index layouts that produce the
// same pixels and ray counts; the defaults are what every number in DESIGN.md refers to.
#ifdef R1_TUNING
#include <stdlib.h>
static inline long long r1_knob(const char *name, long long dflt)
{
    const char *v = getenv(name);
    return v ? atoll(v) : dflt;
}
static inline double r1_knob_f(const char *name, double dflt)
{
    const char *v = getenv(name);
    return v ? atof(v) : dflt;
}
#else
static inline long long r1_knob(const char *, long long dflt) { return dflt; }
static inline double r1_knob_f(const char *, double dflt) { return dflt; }
#endif

// Division of n < 2^31 by a launch constant: pow2 ? n >> shift : mulhi(n, mul) >> shift, with
// mul = ceil(2^(32+shift) / d), shift = floor(log2 d) (exact for every n < 2^31; r1_capi.cpp).
struct R1FastDiv
{
    uint32_t mul, shift, pow2;
};

// Frame batches (R1TraceArgs::batch), device memory
struct R1BatchArgs
{
    uint32_t n_frames;   // >= 2
    uint32_t seed_stride;
    R1FastDiv div_tiles; // by n_local_tiles
    uint32_t n_local_tiles;
};

// What the waves that sum finished tiles need (R1_LAND); by value in the kernel arguments, read when a wave has run out of samples.
struct R1LandArgs
{
    uint8_t *out;                   // frame 0's pixels: row-major image (block_layout 0) or dense tile block (1); device memory or page-locked host memory
    unsigned long long *rays_dst;   // !rays_in_out: the frame's ray count (device memory or a page-locked host word)
    unsigned long long out_stride;  // frame f of a batch: out + f * out_stride
    unsigned long long rays_offset; // rays_in_out: frame f's uint64 count at out + f * out_stride + rays_offset
    unsigned long long *frame_rays; // [n_frames] ray-count accumulators (zero between launches)
    uint32_t *frame_left;           // [n_frames] tiles not yet resolved (n_local_tiles between launches)
    uint32_t *clear_heads;          // queue heads and wave counts of the set the PREVIOUS launch through this context used: zeroed by workgroup 0
    uint32_t clear_count;           // ... that many words, 32 words apart
    uint32_t n_frames;              // frames of the launch (1: a single frame)
    uint32_t rays_in_out, block_layout;
    float inv_spp;                  // (float)(1.0f / spp), rayweek1.cpp:765
    uint32_t *owed_spill;           // [waves of the grid][spill_stride]: a wave's list of the tiles it took chunks from, beyond the 24 entries it keeps in LDS
    uint32_t spill_stride;          // = tiles of the launch (a wave meets a tile at most once)
    uint32_t *error;                // page-locked host word (or null): set if a wave gave up on a tile (R1_LAND_MAX_WAIT) — never in a correct run
};

// Everything the trace kernel needs; passed by value (kernarg segment => SGPRs).
struct R1DeviceScene
{
    // Prefilter table over the GROUPS of active spheres (inv_radius != 0; <= R1_GROUP_MAX nearby
    // spheres per group, bounding sphere (g, R)), 8 floats per PAIR of groups
    // {gx0 gx1 gy0 gy1 gz0 gz1 Kp0 Kp1}, Kp = |g|^2 - R^2 - slack; padded with never-candidate
    // entries (Kp = +inf) to a multiple of 8 spheres PLUS one extra chunk of 8 (prefetch target).
    const float4 *sweep;
    // Exact table, same indexing: {cx, cy, cz, radius_sq} and {inv_radius, albedo rgb},
    // {type, param}.
    const float4 *exact;
    const float4 *shade;   // {inv_radius, albedo_r, albedo_g, albedo_b}
    const float4 *exact_g;   // R1_EXACT_G: the same {cx, cy, cz, radius_sq} in GROUP order, [n_sweep (+pad)][R1_GROUP_MAX] ({0, 0, 0, -inf} = none: never an offer):
                             // the exact phase fetches a member's sphere and its index side by side instead of one after the other
    const uint32_t *members; // [n_sweep (+pad)][R1_GROUP_MAX] active indices of a group's spheres, 0xFFFFFFFF = none
    const float4 *mat;     // {bit_cast<float>(type), param, 1/ref_idx, ((1-ref)/(1+ref))^2} (last two: dielectrics)
    uint32_t n_multi;      // groups [0, n_multi) have 2..R1_GROUP_MAX members, groups >= n_multi exactly one
    uint32_t n_active;     // real entries
    uint32_t n_sweep;      // GROUPS, padded to a multiple of 8 (+8 prefetch); big scenes: of R1_TILE_SPHERES (+ one tile)
    // R1_VARIANT_BVH (r1_bvh.cpp): binary tree of boxes over the active spheres.  Node = 4 float4:
    // {m0x m1x m0y m1y} {m0z m1z e0x e1x} {e0y e1y e0z e1z} {A K child0 child1}; child i has
    // centre m_i and half extent e_i, inflated per ray by pad = A |o - bvh_centre|^2 + K.  Child
    // reference: bit 31 clear = inner node index; set = leaf, bits 28..30 number of sphere PAIRS,
    // bits 0..27 first pair of bvh_prims (2 float4 per pair {cx_a cx_b cy_a cy_b} {cz_a cz_b rsq_a
    // rsq_b}) / bvh_ids (2 active indices per pair).  Node 0 is the root.
    const float4 *bvh_nodes;
    const float4 *bvh_prims;
    const uint32_t *bvh_ids;
    float bvh_centre[3];
    uint32_t bvh_pad_local; // 1: pad = A |m0 + m1 - 2 o|^2 + K (scenes of small spheres, r1_bvh.cpp)
    uint32_t bvh_root_leaf; // 1 / 2: child 0 / 1 of the root is a leaf of <= 2 pairs that every ray tests: the root step of bvh_advance; 0: none
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
    uint32_t n_local_tiles;      // tiles this shard owns (per frame)
    // Frame batches (throughput entry points): ONE launch carries n_frames frames of the same scene, camera and size.  The
    // queue is frame-major — padded tile index j = f * n_local_tiles + (local tile of frame f) — so the persistent waves flow
    // from one frame into the next without draining (a wave's last ~40 iterations run with few live lanes; per frame of
    // 1200x800x10 that is ~8 % of its iterations, per 1/8-frame of an 8-GPU rank ~25 %).  Frame f is seeded seed + f * seed_stride.
    // The batch's numbers sit behind a pointer (null: a single frame) and are fetched with scalar loads where a sample starts:
    // five more kernel arguments cost the tree kernel nine more spilled SGPRs and 2 % of its rate.
    const struct R1BatchArgs *batch;
    // Sample slots are enumerated over PADDED tiles: slot k = (local tile j, pixel in the
    // tile_w x tile_h tile, sample s) = ((j * tile_h + ly) * tile_w + lx) * spp + s.  Slots of
    // pixels outside the image (right/top edge tiles) are void: skipped by the tracer, never
    // written, ignored by the resolve pass.
    uint32_t full;               // tile_w * tile_h * spp slots per local tile
    R1FastDiv div_full, div_spp, div_tw, div_tx; // by full, spp, tile_w, tiles_x
    uint32_t total_samples;      // n_local_tiles * full (queue length)
    uint32_t chunk_min, chunk_max; // samples a wave takes from the queue per atomic (guided: remaining / (2 waves), clamped)
    uint32_t *queue;             // global sample counter(s) (zeroed before the launch); sub-queue q at queue + 32 q (its own 128-byte line)
    uint32_t nq;                 // 1: one guided queue (chunk_min..chunk_max).  > 1 (latency mode): nq sub-queues of fixed chunks of
                                 // chunk_max slots, chunk c belongs to sub-queue c % nq and a wave only pulls from sub-queue wave % nq:
                                 // a returning atomic on ONE line sustains 88 M/s on this chip (tools/ubench_atomic.hip), too few for
                                 // 6144 waves taking a wave-full at a time
    float4 *samples;             // [total_samples] {r, g, b, bit_cast<float>(rays)}.  PIXEL mode (r1_trace.hpp struct Pixel): the
                                 // queue holds PIXELS (full = tile_w * tile_h, total_samples = padded pixels of the shard) and this
                                 // is the uint8 RGB output the kernel resolves into
    unsigned long long *num_rays; // accumulated color() invocations
    uint32_t *gstack;             // big scenes: attenuation stack [R1_STACK_ENTRIES][grid threads], else null
    unsigned long long *stats;    // diagnostic counters (R1_VARIANT_STATS builds only), else null
    uint32_t bvh_lds_f4;          // tree kernels: float4 of the node table each workgroup copies into LDS behind the traversal stack (0: none)
    int32_t bvh_depth;            // tree kernels: traversal stack entries per thread (dynamic LDS = depth * R1_BLOCK * 4)
    int32_t block_layout;         // PIXEL mode: 1 = `samples` is a dense tile block (pixel index = queue slot), 0 = a row-major image
    float inv_spp;                // PIXEL mode: (float)(1.0f / spp), rayweek1.cpp:765
    // R1_LAND (tiles resolved inside the kernel; the throughput builds of the product kernels)
    uint32_t land_res;            // 1: a landing launch (the throughput builds take no other)
    uint32_t land_tag;            // launch generation << 8, or-ed into every sample record's ray-count word: a record is the launch's own iff its tag matches
    uint32_t *land_cnt;           // [n_frames * n_local_tiles] x R1_LAND_CNT_STRIDE words: samples each tile still lacks; the tracing waves subtract, the wave that
                                  // owes the tile re-arms
    R1LandArgs land;
    const float4 *bvh_wide;       // R1_BVH4: the 4-wide table the small-scene kernels copy into LDS instead of scene.bvh_nodes (bvh_lds_f4 float4)
    uint32_t entry_lds;           // R1_ENTRY, small-scene kernels, single frames: the workgroups keep the table as 16-bit words in LDS behind their node table (0: read from bvh_entry)
    const uint32_t *bvh_entry;    // R1_ENTRY: [n_frames * n_local_tiles] child reference (the kernel's form) a primary ray of that tile starts at after the root step
    uint32_t coop_lanes;          // small scenes: once the queue is empty, a wave with <= coop_lanes live paths tests each of them
                                  // against ALL spheres, 64 at a time across the wave (cooperative_sweep), instead of walking the tree
                                  // with 60 lanes masked off: the frame's tail is a few 51-bounce chains, and this shortens a step
                                  // of such a chain from ~16 k cycles of dependent node fetches to ~4 k
};

// Wavefront variant (R1_VARIANT_WAVEFRONT): the same path tracer split into
// generate / intersect / shade kernels with the paths and per-level queues in HBM.
struct R1WaveArgs
{
    R1TraceArgs t;        // scene, camera, frame, samples, num_rays; gstack = attenuation stack [entry][path]
    float4 *paths;        // [3][n_paths]: {ox oy oz dx} {dy dz s_scalar s0} {s1 s2 k rays|depth<<8|sp<<16}
    float2 *hits;         // [n_paths] {t, bit_cast<float>(hit index)}
    uint32_t *queue[2];   // path slots alive at the current / next level
    uint32_t *counts;     // [R1_STACK_ENTRIES + 2] queue length per level (zeroed before the frame)
    uint32_t n_paths;     // = total_samples: path slot = sample slot
    int32_t level;        // color() depth this launch works on
};

struct R1ResolveArgs
{
    const float4 *samples;
    uint32_t full;
    int32_t width, height, spp;
    int32_t tile_w, tile_h, tiles_x;
    int32_t shard, num_shards;
    uint32_t n_local_tiles;      // per frame
    float inv_spp;               // (float)(1.0f / spp)
    uint8_t *out;                // row-major image or dense tile block (of frame 0)
    int32_t block_layout;        // 0: row-major width*height*3, 1: dense tile block
    // frame batches: frame f resolves tiles [f * n_local_tiles, (f + 1) * n_local_tiles) of the sample records into
    // out + f * out_stride; its ray count — the sum of its samples' counts, rayweek1.cpp:809-813 — is left as partial sums
    // in frame_rays[(tile of the batch) * gridDim.x + blockIdx.x] (plain stores, no atomics) and added up per frame by
    // r1_batch_counts_kernel into out + f * out_stride + rays_offset (uint64)
    uint32_t n_frames;
    size_t out_stride, rays_offset;
    unsigned long long *frame_rays; // null: single frame, the trace kernel counted
    // end of the frame (block (0,0), after the trace kernel): publish the ray count the trace kernel accumulated in the
    // context's counter block, then zero the block for the next frame — two memset launches per frame less
    const unsigned long long *rays_src; // null: the trace kernel counted into the caller's word itself
    unsigned long long *rays_dst;
    uint32_t *reset;             // R1_COUNTER_BYTES to zero, or null
};

#endif
