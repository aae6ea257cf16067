// r1_capi.cpp — the C-ABI of include/rays1.h on top of the HIP runtime: context, scene
// upload, launches, timing.  Host code only (the kernels live in r1_trace.hpp).
//
// There is no CPU fallback anywhere in this file: without a HIP device every compute
// entry point returns R1_ENODEVICE / R1_EHIP.

#include <hip/hip_runtime.h>
#include <cmath>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <utility>
#include <functional>
#include <vector>

#include "../../include/rays1.h"
#include "r1_device.h"

extern "C" hipError_t r1_launch_trace(const R1TraceArgs *args, int variant, int big, int mode, int blocks, hipStream_t stream);
extern "C" int r1_trace_mode(int variant, int big, int wanted); // 0 samples + one queue, 1 latency, 2 pixel: what is built for (variant, big)
extern "C" hipError_t r1_launch_resolve(const R1ResolveArgs *args, int max_rows, hipStream_t stream);
extern "C" hipError_t r1_launch_wavefront(R1WaveArgs *w, int blocks, hipStream_t stream);
extern "C" hipError_t r1_launch_land_arm(uint32_t *tile_cnt, unsigned long long *frame_rays, uint32_t *frame_left, uint32_t n_frames, uint32_t n_local_tiles,
                                         int width, int height, int spp, int tile_w, int tile_h, int tiles_x, int shard, int num_shards, hipStream_t stream);
extern "C" hipError_t r1_launch_put6(void *dst, const uint32_t *words, hipStream_t stream);
extern "C" hipError_t r1_launch_assemble(const void *blocks, void *rgb, int width, int height, int tile_w, int tile_h, int tiles_x, int num_shards,
                                         size_t shard_stride, int n_frames, size_t frame_in, size_t frame_out, size_t total_offset, long long total_out,
                                         int want_total, hipStream_t stream);
extern "C" size_t r1_frame_record_bytes(const r1_params *p); // r1_host.cpp
extern "C" hipError_t r1_trace_occupancy(int variant, int big, int mode, size_t dyn_lds, int *blocks_per_cu);
extern "C" int r1_params_check(const r1_params *p); // r1_host.cpp

// r1_bvh.cpp
struct R1Bvh
{
    std::vector<float> nodes;
    std::vector<float> prims;
    std::vector<uint32_t> ids;
    int max_depth = 0;
    uint32_t n_leaves = 0;
    float centre[3] = {0, 0, 0};
    int pad_local = 0;
    int root_leaf = 0;
};
void r1_build_bvh(uint32_t na, const float *cx, const float *cy, const float *cz, const float *rsq, const double *rbound, int leaf_max,
                  R1Bvh &out);
int r1_active_spheres(const r1_scene *s, std::vector<uint32_t> &active_to_scene); // inv_radius != 0, finite (r1_bvh.cpp)
double r1_bound_radius(float radius_sq, float inv_radius);

// ---- errors ---------------------------------------------------------------------------------

static thread_local char g_error[512] = "";

extern "C" void r1_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *r1_last_error(void) { return g_error; }
extern "C" int r1_abi_version(void) { return R1_ABI_VERSION; }

#define R1_HIP(call)                                                                                                   \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (call);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
        {                                                                                                              \
            r1_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);                   \
            return e_ == hipErrorOutOfMemory ? R1_ENOMEM : R1_EHIP;                                                    \
        }                                                                                                              \
    } while (0)

// ---- context ----------------------------------------------------------------------------------

struct DevBuf
{
    void *p = nullptr;
    size_t cap = 0;
};

struct r1_context
{
    int device = 0;
    int cus = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    hipEvent_t last0 = nullptr, last1 = nullptr, last2 = nullptr; // the events the last enqueued frame recorded (own set or ring slot)
    bool timing_valid = false;
    // optional per-frame event ring (r1_timing_begin/_end): 3 events per frame
    std::vector<hipEvent_t> ring;
    int ring_frames = 0, ring_used = 0;
    bool ring_on = false;

    // scene
    DevBuf sweep, exact, exact_g, shade, mat, members;
    DevBuf bvh_nodes, bvh_prims, bvh_ids; // R1_VARIANT_BVH (r1_bvh.cpp)
    DevBuf wf_paths, wf_hits, wf_queue, wf_counts; // R1_VARIANT_WAVEFRONT workspace
    DevBuf wave_log;                               // STATS builds: per-wave {start, queue empty, end, iterations}
    unsigned long long wave_log_ptr = 0;
    uint32_t wave_log_waves = 0;
    uint32_t n_bvh_nodes = 0, n_bvh_leaves = 0;
    int bvh_depth = 0;
    float bvh_centre[3] = {0, 0, 0};
    int bvh_pad_local = 0;
    int bvh_root_leaf = 0;
    uint32_t n_active = 0, n_sweep = 0, n_padded_scene = 0, n_groups = 0, n_multi = 0;
    std::vector<uint32_t> active_to_scene;
    R1DeviceCamera cam;
    bool have_scene = false;
    // what the device tables were built from: r1_set_scene with the same arrays again (the drop-in's benchmark()
    // uploads its scene on every run, rayweek1.cpp:969-984) keeps the tables and the tree
    std::vector<float> src_f32[9];
    std::vector<uint8_t> src_mat;
    r1_camera src_cam;

    // per-frame workspace
    DevBuf counters, samples, image;
    // frame batches: the batch's numbers live in one of eight device slots (counter tail + 64 + 32 i), written by a one-thread launch whose
    // values travel in its kernel arguments (an asynchronous copy from a host member could be overtaken by the next call)
    R1BatchArgs batch_args_last = {0, 0, {0, 0, 0}, 0};
    hipStream_t batch_args_stream = nullptr;
    int batch_args_slot = -1;
    // R1_LAND (tiles resolved inside the trace kernel): per-context state
    bool land_prev = false;      // the last launch through this context was a LAND launch
    int land_parity = 0;         // the set of queue heads that launch used
    uint32_t land_gen = 0;       // launch generation: the tag of its sample records (1 .. 2^24 - 1)
    bool land_armed = false;     // the countdowns / accumulators behind the counter block hold the values of land_key
    r1_params land_key;
    int land_frames = 0;
    DevBuf batch_rays;  // frame batches: per-frame ray-count accumulators of the resolve launch + its finished-workgroup counter
    int tile_frames = 0; // frames per launch the tile arithmetic below was made for
    bool counters_clean = false; // the last frame's resolve launch zeroed the counter block: the next frame needs no memset
    r1_params tile_key;
    bool tile_key_valid = false;
    uint32_t n_local_tiles = 0, total_samples = 0, full = 0;

    // one page-locked, device-visible word: the synchronous entry points let the frame's last launch store the ray count
    // straight into host memory (8 bytes over PCIe at the end of r1_resolve_kernel) instead of enqueueing a second copy
    unsigned long long *host_word = nullptr, *host_word_dev = nullptr;
    int default_variant = 4;    // what R1_VARIANT_DEFAULT resolves to for the scene in the context (r1_set_scene): synchronous frames
    int default_variant_tp = 4; // ... and frames in flight (the throughput kernels: measured apart, the two kernel families do not rank alike)
    int occupancy[48] = {0}; // [variant + 8 * big + 16 * mode]
    bool pixel_mode = false; // r1_set_pixel_mode
    DevBuf gstack; // blocks per CU of the trace kernel, by variant
    // per-tile entry nodes for primary rays (R1_ENTRY): the tree's nodes on the host, the device table, what it was computed for
    std::vector<float> bvh_nodes_host;
    DevBuf bvh_entry;
    DevBuf bvh_wide; // R1_BVH4: the collapsed table (small scenes)
    uint32_t bvh_wide_f4 = 0;
    int bvh_wide_stack = 0;
    r1_params entry_key;
    int entry_frames = 0;
    bool entry_valid = false;
    DevBuf land_spill; // R1_LAND: [waves of the grid][tiles of the launch] every wave's list of the tiles it took chunks from

    r1_launch_info info;
};

static int ensure(DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p)
        return R1_OK;
    if (b.p)
    {
        R1_HIP(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    R1_HIP(hipMalloc(&b.p, want));
    b.cap = want;
    return R1_OK;
}

static void release(DevBuf &b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" int r1_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
    {
        r1_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return R1_ENODEVICE;
    }
    return n;
}

extern "C" int r1_create(int device, r1_context **out)
{
    if (!out)
        return R1_EINVAL;
    *out = nullptr;
    int n = r1_device_count();
    if (n <= 0)
    {
        if (n == 0)
            r1_set_error("no HIP device visible (librays1 has no CPU fallback)");
        return R1_ENODEVICE;
    }
    if (device < 0 || device >= n)
    {
        r1_set_error("device %d out of range (0..%d)", device, n - 1);
        return R1_EINVAL;
    }
    r1_context *c = new (std::nothrow) r1_context();
    if (!c)
        return R1_ENOMEM;
    c->device = device;
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess)
        e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreate(&c->ev0);
    if (e == hipSuccess)
        e = hipEventCreate(&c->ev1);
    if (e == hipSuccess)
        e = hipEventCreate(&c->ev2);
    if (e != hipSuccess)
    {
        r1_set_error("r1_create: %s", hipGetErrorString(e));
        delete c;
        return R1_EHIP;
    }
    if (hipHostMalloc((void **)&c->host_word, 64, hipHostMallocMapped) == hipSuccess)
    {
        if (hipHostGetDevicePointer((void **)&c->host_word_dev, c->host_word, 0) != hipSuccess)
        {
            (void)hipHostFree(c->host_word);
            c->host_word = c->host_word_dev = nullptr;
        }
    }
    else
        c->host_word = nullptr; // (not fatal: the count is copied instead)
    (void)hipGetLastError();
    c->cus = prop.multiProcessorCount;
    memset(&c->info, 0, sizeof(c->info));
    c->info.compute_units = c->cus;
    *out = c;
    return R1_OK;
}

// internal (r1_multi.cpp): the context's own stream, so that the in-process multi-GPU path runs its collectives on the stream the
// context's uploads and frames already use instead of a second stream per device (every stream wants a hardware queue)
extern "C" void *r1_context_stream(r1_context *c) { return c ? (void *)c->stream : nullptr; }

extern "C" void r1_destroy(r1_context *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    release(c->sweep), release(c->exact), release(c->exact_g), release(c->shade), release(c->mat), release(c->members);
    release(c->bvh_nodes), release(c->bvh_prims), release(c->bvh_ids);
    release(c->wf_paths), release(c->wf_hits), release(c->wf_queue), release(c->wf_counts);
    release(c->bvh_wide), release(c->bvh_entry), release(c->land_spill), release(c->gstack), release(c->counters), release(c->samples), release(c->image), release(c->batch_rays);
    release(c->wave_log);
    if (c->host_word)
        (void)hipHostFree(c->host_word);
    for (hipEvent_t e : c->ring)
        (void)hipEventDestroy(e);
    if (c->ev0)
        (void)hipEventDestroy(c->ev0);
    if (c->ev1)
        (void)hipEventDestroy(c->ev1);
    if (c->ev2)
        (void)hipEventDestroy(c->ev2);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---- scene upload -------------------------------------------------------------------------------

// largest float <= v, then one more step down (guards the double->float conversion)
static float round_down(double v)
{
    float f = (float)v;
    if ((double)f > v)
        f = nextafterf(f, -INFINITY);
    return nextafterf(f, -INFINITY);
}


// ---- sphere groups (level 1 of the sweep) ------------------------------------------------------
// The sweep tests GROUPS of up to R1_GROUP_MAX nearby spheres against a bounding sphere first and
// re-tests the members of flagged groups exactly (r1_trace.hpp).  Grouping is a pure work
// reduction: every active sphere belongs to exactly one group, the group test is conservative,
// and hits are still resolved per sphere in the reference's arithmetic and index order.
struct R1Group
{
    double gx, gy, gz, radius; // bounding sphere: |c_i - g| + r_i <= radius for every member
    double c_max2;             // max(|g|^2, max_i |c_i|^2): magnitude that scales the fp32 error terms
    uint32_t member[R1_GROUP_MAX];
    int n;
};

static uint64_t spread21(uint64_t v) // 21 bits -> every third bit
{
    v &= 0x1FFFFF;
    v = (v | v << 32) & 0x1F00000000FFFFull;
    v = (v | v << 16) & 0x1F0000FF0000FFull;
    v = (v | v << 8) & 0x100F00F00F00F00Full;
    v = (v | v << 4) & 0x10C30C30C30C30C3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}

static void bound_of(const std::vector<uint32_t> &m, const std::vector<double> &x, const std::vector<double> &y,
                     const std::vector<double> &z, const std::vector<double> &r, R1Group &g)
{
    // centre: middle of the members' axis-aligned extent (tight for the 2x2 blocks of a lattice)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t a : m)
    {
        const double c[3] = {x[a], y[a], z[a]};
        for (int k = 0; k < 3; ++k)
            lo[k] = fmin(lo[k], c[k] - r[a]), hi[k] = fmax(hi[k], c[k] + r[a]);
    }
    g.gx = 0.5 * (lo[0] + hi[0]), g.gy = 0.5 * (lo[1] + hi[1]), g.gz = 0.5 * (lo[2] + hi[2]);
    g.radius = 0;
    g.c_max2 = g.gx * g.gx + g.gy * g.gy + g.gz * g.gz;
    for (uint32_t a : m)
    {
        const double dx = x[a] - g.gx, dy = y[a] - g.gy, dz = z[a] - g.gz;
        g.radius = fmax(g.radius, sqrt(dx * dx + dy * dy + dz * dz) + r[a]);
        g.c_max2 = fmax(g.c_max2, x[a] * x[a] + y[a] * y[a] + z[a] * z[a]);
    }
}

// lone[a]: sphere a must stay a group of its own (its radius_sq and inv_radius disagree, so the
// R <= R1_GROUP_RATIO x r bound of the slack analysis cannot be relied on; a single-sphere group
// needs no such bound: flagged by the reference means dist^2 <= r^2 + E1 <= R^2 + E1)
static std::vector<R1Group> build_groups(uint32_t na, const std::vector<double> &x, const std::vector<double> &y,
                                         const std::vector<double> &z, const std::vector<double> &r, const std::vector<char> &lone)
{
    // Grouping trades level-1 tests for extra member slots in the exact phase: it pays once the
    // sweep is long (large scene: 484 spheres, 1.5x), not for a few dozen spheres (medium scene:
    // 46 spheres, 27.2 vs 24.5 Grays/s ungrouped vs grouped) — R1_GROUP_MAX overrides for tuning.
    static const long gmax_env = (long)r1_knob("R1_GROUP_MAX", 0); // 0: automatic
    const long gmax_want = gmax_env > 0 ? gmax_env : (na > R1_GROUP_MIN_SPHERES ? R1_GROUP_MAX : 1);
    const int gmax = gmax_want > R1_GROUP_MAX ? R1_GROUP_MAX : (int)gmax_want;
    std::vector<R1Group> groups;
    auto close = [&](const std::vector<uint32_t> &m) {
        R1Group g;
        memset(&g, 0, sizeof(g));
        bound_of(m, x, y, z, r, g);
        g.n = (int)m.size();
        for (int k = 0; k < R1_GROUP_MAX; ++k)
            g.member[k] = k < g.n ? m[k] : 0xFFFFFFFFu;
        groups.push_back(g);
    };
    if (na == 0)
        return groups;
    // spheres much larger than the typical one (the ground, the r = 2 balls) stay alone
    std::vector<double> rs(r.begin(), r.begin() + na);
    std::nth_element(rs.begin(), rs.begin() + na / 2, rs.end());
    const double r_med = rs[na / 2];
    std::vector<uint32_t> small;
    for (uint32_t a = 0; a < na; ++a)
        if (gmax > 1 && r[a] <= 2.5 * r_med && !lone[a])
            small.push_back(a);
        else
            close(std::vector<uint32_t>(1, a));
    if (small.empty())
        return groups;
    // Morton order of the centres, then greedy runs of <= gmax spheres whose bounding sphere
    // stays within R1_GROUP_RATIO x the smallest member radius (keeps the bound selective and the
    // slack analysis of DESIGN.md §4.1 valid)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t a : small)
    {
        lo[0] = fmin(lo[0], x[a]), hi[0] = fmax(hi[0], x[a]);
        lo[1] = fmin(lo[1], y[a]), hi[1] = fmax(hi[1], y[a]);
        lo[2] = fmin(lo[2], z[a]), hi[2] = fmax(hi[2], z[a]);
    }
    const double ext = fmax(fmax(hi[0] - lo[0], hi[1] - lo[1]), fmax(hi[2] - lo[2], 1e-30));
    std::vector<std::pair<uint64_t, uint32_t>> keyed;
    for (uint32_t a : small)
    {
        const uint64_t qx = (uint64_t)((x[a] - lo[0]) / ext * 2097151.0), qy = (uint64_t)((y[a] - lo[1]) / ext * 2097151.0),
                       qz = (uint64_t)((z[a] - lo[2]) / ext * 2097151.0);
        keyed.push_back({spread21(qx) | spread21(qy) << 1 | spread21(qz) << 2, a});
    }
    std::sort(keyed.begin(), keyed.end());
    std::vector<uint32_t> cur;
    for (auto &ka : keyed)
    {
        std::vector<uint32_t> tryg = cur;
        tryg.push_back(ka.second);
        bool ok = (int)tryg.size() <= gmax;
        if (ok && tryg.size() > 1)
        {
            R1Group g;
            bound_of(tryg, x, y, z, r, g);
            double rmin = 1e300;
            for (uint32_t a : tryg)
                rmin = fmin(rmin, r[a]);
            static const double ratio = r1_knob_f("R1_GROUP_RATIO", R1_GROUP_RATIO);
            ok = g.radius <= (ratio < R1_GROUP_RATIO ? ratio : R1_GROUP_RATIO) * rmin; // the slack analysis needs <= R1_GROUP_RATIO
        }
        if (ok)
            cur = tryg;
        else
        {
            close(cur);
            cur.assign(1, ka.second);
        }
    }
    if (!cur.empty())
        close(cur);
    return groups;
}

// ---- R1_VARIANT_DEFAULT -------------------------------------------------------------------------------------------------------------
// DEFAULT is the box tree for every scene: a property of the build, so what a context launches depends on its arguments only and
// the ranks of a multi-GPU job always run the same kernel.  (Round 3 timed scenes of 9..127 spheres through both kernels when
// they were set, because the reference's dense medium scene was 4-5 % faster through the ungrouped sweep in round 2.  Since the
// root step of the walk the tree is ahead on every row of the crossover table, profiles/r04/tree_vs_sweep_crossover*.txt — by
// 0-3 % for one synchronous frame of the medium scene, 7 % with frames in flight — and a probe inside r1_set_scene made the first
// timed benchmark() call, the kernel choice of each rank and a 62 MB workspace depend on a wall-clock race: VERDICT r03 / ADVICE r03.)
struct Batch;
struct Landing;
static int enqueue_frame(r1_context *c, const r1_params *p, void *d_out, int block_layout, void *d_rays, hipStream_t st, bool throughput_mode,
                         const Batch *batch = nullptr, Landing *landing = nullptr);

// ---- 4-wide nodes (R1_BVH4) -------------------------------------------------------------------------------------
// The binary tree of a small scene collapsed: a wide node starts as a binary node's two children and replaces its inner child with the
// largest box by that child's own two children until it holds four (or only leaves).  Boxes and leaves are the binary tree's, so the
// spheres offered to a ray — and with them every pixel — stay the same; only the number of dependent trips of the walk changes.
// Table: slot 0 = the binary root (4 float4 as r1_bvh.cpp writes them, child references in the 16-bit form, + 3 float4 of padding: the
// root step of bvh_advance reads it), then wide node i >= 1 at float4 7 i: children 0, 1 and children 2, 3 as two row triples of the binary
// form ({m0x m1x m0y m1y} {m0z m1z e0x e1x} {e0y e1y e0z e1z}), then {ref[4]};
// an empty slot has e = -inf (never passes) and the reference of a leaf without pairs.  Returns the most entries a lane's stack can hold.
static int build_wide(const std::vector<float> &bin, int root_leaf, std::vector<float> &out)
{
    auto ref_of = [&](uint32_t node, int k) { uint32_t v; memcpy(&v, &bin[16 * (size_t)node + 14 + k], 4); return v; };
    auto ref16 = [](uint32_t ref) { return ((ref >> 16) & 0xF000u) | (ref & 0x0FFFu); };
    struct Slot { float m[3], e[3]; uint32_t ref; };
    auto slot_of = [&](uint32_t node, int k) {
        Slot s;
        const float *q = &bin[16 * (size_t)node];
        for (int a = 0; a < 3; ++a)
            s.m[a] = q[2 * a + k], s.e[a] = q[6 + 2 * a + k];
        s.ref = ref_of(node, k);
        return s;
    };
    const size_t n_bin = bin.size() / 16;
    // Which grandchildren a wide node takes: the cut of <= 4 slots through the binary subtree that minimises the summed surface area of the
    // wide nodes below it — the expected number of wide visits of a random ray, the measure the binary tree was built by.  (Taking the
    // largest child first, top-down, leaves the bottom level of a balanced tree of odd height as two-slot nodes: 85 wide nodes for the
    // large scene's 128 binary ones instead of 43.)  <= 8 cuts per node, memoised over <= 256 nodes.
    std::vector<double> cost(n_bin, -1.0);
    std::vector<std::vector<Slot>> cut_of(n_bin);
    auto area_of = [](const Slot &q) { return std::isfinite(q.e[0] + q.e[1] + q.e[2]) ? (double)q.e[0] * q.e[1] + (double)q.e[1] * q.e[2] + (double)q.e[2] * q.e[0] : 0.0; };
    std::function<double(uint32_t)> solve = [&](uint32_t b) -> double {
        if (cost[b] >= 0.0)
            return cost[b];
        std::vector<std::vector<Slot>> cuts = {{slot_of(b, 0), slot_of(b, 1)}};
        for (size_t i = 0; i < cuts.size(); ++i)
            if (cuts[i].size() < 4)
                for (size_t k = 0; k < cuts[i].size(); ++k)
                    if (!(cuts[i][k].ref & 0x80000000u))
                    {
                        std::vector<Slot> c = cuts[i];
                        const uint32_t x = c[k].ref;
                        c[k] = slot_of(x, 0);
                        c.push_back(slot_of(x, 1));
                        cuts.push_back(c);
                    }
        double best = 1e300;
        size_t pick = 0;
        for (size_t i = 0; i < cuts.size(); ++i)
        {
            double sum = 0;
            for (const Slot &q : cuts[i])
                if (!(q.ref & 0x80000000u))
                    sum += area_of(q) + solve(q.ref);
            if (sum < best)
                best = sum, pick = i;
        }
        cut_of[b] = cuts[pick];
        return cost[b] = best;
    };
    std::vector<uint32_t> wide_of(n_bin, 0u), order; // binary inner node -> wide index (0: none yet)
    auto wide_index = [&](uint32_t b) {
        if (!wide_of[b])
            order.push_back(b), wide_of[b] = (uint32_t)order.size();
        return wide_of[b];
    };
    out.assign(28, 0.0f);
    memcpy(out.data(), bin.data(), 64);
    // the walk starts at the root's inner child (root step) — or, a tree without that shape, at the root itself
    const uint32_t start = root_leaf ? ref_of(0, root_leaf == 1 ? 1 : 0) : 0u;
    std::vector<int> above; // stack entries the ancestors of wide node i can leave
    int need = 0;
    if (!(start & 0x80000000u))
        wide_index(start), above.push_back(0);
    for (size_t i = 0; i < order.size(); ++i)
    {
        solve(order[i]);
        const std::vector<Slot> slots = cut_of[order[i]];
        const int here = above[i] + (int)slots.size() - 1;
        need = std::max(need, here);
        float w[28];
        for (int k = 0; k < 4; ++k)
        {
            const bool used = k < (int)slots.size();
            const int base = 12 * (k >> 1), j = k & 1; // children 0, 1 / 2, 3: one binary-form row triple each
            for (int a = 0; a < 3; ++a)
                w[base + 2 * a + j] = used ? slots[k].m[a] : 0.0f, w[base + 6 + 2 * a + j] = used ? slots[k].e[a] : -INFINITY;
            uint32_t r = 0x8000u; // (an empty slot never passes — e = -inf — and if it did, it is a leaf of no pairs)
            if (used)
            {
                if (slots[k].ref & 0x80000000u)
                    r = ref16(slots[k].ref);
                else
                {
                    const size_t before = order.size();
                    r = wide_index(slots[k].ref);
                    if (order.size() != before)
                        above.push_back(here);
                }
            }
            memcpy(&w[24 + k], &r, 4);
        }
        out.insert(out.end(), w, w + 28);
    }
    // slot 0's child references in the kernel's form: leaves as r1_ref16, the inner child as its wide index
    for (int k = 0; k < 2; ++k)
    {
        const uint32_t r = ref_of(0, k), r16 = (r & 0x80000000u) ? ref16(r) : wide_of[r];
        memcpy(&out[14 + k], &r16, 4);
    }
    return std::max(need, 1);
}

extern "C" int r1_set_scene(r1_context *c, const r1_scene *s, const r1_camera *cam)
{
    if (!c || !s || !cam || !s->center_x || !s->center_y || !s->center_z || !s->radius_sq || !s->inv_radius || !s->mat_type ||
        !s->albedo_r || !s->albedo_g || !s->albedo_b || !s->mat_param)
    {
        r1_set_error("r1_set_scene: null argument");
        return R1_EINVAL;
    }
    R1_HIP(hipSetDevice(c->device));

    const float *const src[9] = {s->center_x, s->center_y, s->center_z, s->radius_sq, s->inv_radius, s->albedo_r, s->albedo_g, s->albedo_b, s->mat_param};
    if (c->have_scene && c->src_mat.size() == s->count && memcmp(&c->src_cam, cam, sizeof(*cam)) == 0 &&
        (s->count == 0 || memcmp(c->src_mat.data(), s->mat_type, s->count) == 0))
    {
        bool same = true;
        for (int k = 0; k < 9 && same; ++k)
            same = s->count == 0 || memcmp(c->src_f32[k].data(), src[k], (size_t)s->count * 4) == 0;
        if (same)
            return R1_OK; // bit-identical scene and camera: everything on the device is current
    }
    c->have_scene = false;

    // active spheres: inv_radius != 0 (rayweek1.cpp:291); order preserved so that ties keep
    // the earlier index as in the reference's in-order resolve loop
    int rc_active = r1_active_spheres(s, c->active_to_scene);
    if (rc_active != R1_OK)
        return rc_active;
    for (uint32_t i : c->active_to_scene)
        if (s->mat_type[i] > R1_MAT_DIELECTRIC)
        {
            r1_set_error("r1_set_scene: sphere %u is hittable but has no material", i);
            return R1_EINVAL;
        }
    const uint32_t na = (uint32_t)c->active_to_scene.size();
    if (na > R1_MAX_ACTIVE)
    {
        r1_set_error("r1_set_scene: %u hittable spheres; this build supports up to %u", na, R1_MAX_ACTIVE);
        return R1_ELIMIT;
    }
    // level 1 of the sweep: groups of nearby spheres with a bounding sphere each
    std::vector<double> ax(na ? na : 1), ay(na ? na : 1), az(na ? na : 1), ar_(na ? na : 1);
    std::vector<char> lone(na ? na : 1, 0);
    for (uint32_t a = 0; a < na; ++a)
    {
        const uint32_t i = c->active_to_scene[a];
        ax[a] = s->center_x[i], ay[a] = s->center_y[i], az[a] = s->center_z[i];
        ar_[a] = r1_bound_radius(s->radius_sq[i], s->inv_radius[i]); // what the exact test can accept, never less
        const double r_test = s->radius_sq[i] > 0 ? sqrt((double)s->radius_sq[i]) : 0.0;
        lone[a] = !(r_test >= ar_[a] * (1.0 - 1e-3)); // SphereSOA::add keeps them within 2 ulp (soa_sphere.cpp:70-85)
    }
    std::vector<R1Group> groups = build_groups(na, ax, ay, az, ar_, lone);
    // multi-member groups first: a flagged group with id >= n_multi is a single sphere and takes
    // one member slot of the exact phase instead of R1_GROUP_MAX
    std::stable_partition(groups.begin(), groups.end(), [](const R1Group &g) { return g.n > 1; });
    const uint32_t ng = (uint32_t)groups.size();
    uint32_t n_multi = 0;
    while (n_multi < ng && groups[n_multi].n > 1)
        ++n_multi;

    // small scenes: whole 8-group chunks + one prefetch chunk; big scenes: whole LDS tiles + one
    // prefetch tile
    const bool big_scene = na > R1_MAX_ACTIVE_10BIT;
    const uint32_t ns = big_scene ? ((ng + R1_TILE_SPHERES - 1) / R1_TILE_SPHERES) * R1_TILE_SPHERES : ((ng + 7u) & ~7u);

    // sweep table: pair layout + one chunk of prefetch padding (see r1_device.h)
    const uint32_t ns_alloc = ns + (big_scene ? R1_TILE_SPHERES : 8);
    std::vector<float> sweep(4 * (size_t)ns_alloc), exact(4 * (size_t)(na ? na : 1)),
        shade(4 * (size_t)(na > R1_MAX_ACTIVE_10BIT ? na : R1_MAX_ACTIVE_10BIT + 1)), // small scenes: any 10-bit index may be read (unwind)
        mat(4 * (size_t)(na ? na : 1));
    std::vector<uint32_t> members((size_t)R1_GROUP_MAX * ns_alloc, 0xFFFFFFFFu);
    auto sweep_slot = [&](uint32_t a, int comp) -> float & { return sweep[8 * (size_t)(a >> 1) + 2 * comp + (a & 1)]; };
    for (uint32_t a = 0; a < ns_alloc; ++a) // never-candidate default
        sweep_slot(a, 0) = sweep_slot(a, 1) = sweep_slot(a, 2) = 0, sweep_slot(a, 3) = INFINITY;
    for (uint32_t g = 0; g < ng; ++g)
    {
        const R1Group &G = groups[g];
        // the bounding sphere as fp32 centre + a radius that still covers the members after the
        // centre is rounded to fp32
        const float gx = (float)G.gx, gy = (float)G.gy, gz = (float)G.gz;
        double R = 0;
        for (int k = 0; k < G.n; ++k)
        {
            const uint32_t a = G.member[k];
            const double dx = ax[a] - gx, dy = ay[a] - gy, dz = az[a] - gz;
            R = fmax(R, sqrt(dx * dx + dy * dy + dz * dz) + ar_[a]);
            members[(size_t)R1_GROUP_MAX * g + k] = a;
        }
        R *= 1.0 + 1e-12;
        const double g2 = (double)gx * gx + (double)gy * gy + (double)gz * gz;
        // Kp = (|g|^2 - R^2) - 2^-15 (C^2 + R^2), rounded down.  C^2 bounds |g|^2 and every
        // member's |c|^2.  The slack covers the fp32 error of the group test itself AND of any
        // member's reference test carried over to the bound: see sweep_prefilter.
        const double kp = (g2 - R * R) - ldexp(fmax(G.c_max2, g2) + R * R, -15) - 1e-30;
        sweep_slot(g, 0) = gx, sweep_slot(g, 1) = gy, sweep_slot(g, 2) = gz, sweep_slot(g, 3) = round_down(kp);
    }
    for (uint32_t a = 0; a < na; ++a)
    {
        const uint32_t i = c->active_to_scene[a];
        const float cx = s->center_x[i], cy = s->center_y[i], cz = s->center_z[i], rsq = s->radius_sq[i];
        exact[4 * a + 0] = cx, exact[4 * a + 1] = cy, exact[4 * a + 2] = cz, exact[4 * a + 3] = rsq;
        shade[4 * a + 0] = s->inv_radius[i], shade[4 * a + 1] = s->albedo_r[i], shade[4 * a + 2] = s->albedo_g[i],
                      shade[4 * a + 3] = s->albedo_b[i];
        uint32_t type = s->mat_type[i];
        memcpy(&mat[4 * a], &type, 4);
        const float ref_idx = s->mat_param[i];
        mat[4 * a + 1] = ref_idx;
        // Dielectric constants the reference recomputes per hit with IEEE float ops
        // (rayweek1.cpp:489 `1.0f / _refIdx`, :456-457 schlick r0): same operations, done once
        float r0 = (1 - ref_idx) / (1 + ref_idx);
        r0 = r0 * r0;
        mat[4 * a + 2] = type == R1_MAT_DIELECTRIC ? 1.0f / ref_idx : 0.0f;
        mat[4 * a + 3] = type == R1_MAT_DIELECTRIC ? r0 : 0.0f;
    }

    // the members' spheres once more, in group order (exact_trips fetches sphere and index side by side)
    std::vector<float> exact_g(4 * (size_t)R1_GROUP_MAX * ns_alloc);
    for (size_t k = 0; k < (size_t)R1_GROUP_MAX * ns_alloc; ++k)
    {
        const uint32_t a = members[k];
        for (int q = 0; q < 4; ++q)
            exact_g[4 * k + q] = a != 0xFFFFFFFFu ? exact[4 * (size_t)a + q] : (q == 3 ? -INFINITY : 0.0f);
    }

    // the optional spatial index over the same active spheres (R1_VARIANT_BVH)
    R1Bvh bvh;
    {
        std::vector<float> fx(na ? na : 1), fy(na ? na : 1), fz(na ? na : 1), fr(na ? na : 1);
        for (uint32_t a = 0; a < na; ++a)
            fx[a] = exact[4 * a + 0], fy[a] = exact[4 * a + 1], fz[a] = exact[4 * a + 2], fr[a] = exact[4 * a + 3];
        static const int leaf_env = (int)r1_knob("R1_BVH_LEAF", 0);
        // leaf size: 4 spheres (2 pairs) on the reference's scenes; 8 on big lattices (measured:
        // 100 004 spheres 3.43 ms against 3.67 ms per 1920x1080x4 frame)
        const int leaf_default = na > R1_MAX_ACTIVE_10BIT ? 2 * R1_BVH_LEAF : R1_BVH_LEAF;
        r1_build_bvh(na, fx.data(), fy.data(), fz.data(), fr.data(), ar_.data(), leaf_env > 0 ? leaf_env : leaf_default, bvh);
        if (bvh.max_depth > R1_BVH_STACK)
        {
            r1_set_error("r1_set_scene: spatial index deeper (%d) than the traversal stack (%d)", bvh.max_depth, R1_BVH_STACK);
            return R1_ELIMIT;
        }
    }

    // leaf_quad fetches a sphere index speculatively for lanes that hold no flagged sphere (slot 3 of the step, whatever the step's
    // pair count): two sentinel words behind the last pair keep that read inside the table
    bvh.ids.resize(bvh.ids.size() + 2, 0xFFFFFFFFu);
    int rc;
    if ((rc = ensure(c->bvh_nodes, bvh.nodes.size() * 4)) || (rc = ensure(c->bvh_prims, bvh.prims.size() * 4)) ||
        (rc = ensure(c->bvh_ids, bvh.ids.size() * 4)))
        return rc;
    if ((rc = ensure(c->sweep, sweep.size() * 4)) || (rc = ensure(c->exact, exact.size() * 4)) ||
        (rc = ensure(c->shade, shade.size() * 4)) || (rc = ensure(c->mat, mat.size() * 4)) ||
        (rc = ensure(c->members, members.size() * 4)) || (rc = ensure(c->exact_g, exact_g.size() * 4)))
        return rc;
    // Uploads go through the context's OWN stream (then one wait): librays1 never touches the null stream.  A process
    // that keeps K frames in flight on K contexts with GPU_MAX_HW_QUEUES = K would otherwise hand one of its K hardware
    // queues to the null stream, and two frames would share a queue and run one after the other (measured: 20 frames
    // land in 20.1 ms instead of 17.0, tools/submit_times.py, profiles/r03/stream_queue_mapping.txt).
    R1_HIP(hipStreamSynchronize(c->stream));
    R1_HIP(hipMemcpyAsync(c->sweep.p, sweep.data(), sweep.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->exact.p, exact.data(), exact.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->shade.p, shade.data(), shade.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->mat.p, mat.data(), mat.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->members.p, members.data(), members.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->exact_g.p, exact_g.data(), exact_g.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->bvh_nodes.p, bvh.nodes.data(), bvh.nodes.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->bvh_prims.p, bvh.prims.data(), bvh.prims.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipMemcpyAsync(c->bvh_ids.p, bvh.ids.data(), bvh.ids.size() * 4, hipMemcpyHostToDevice, c->stream));
    R1_HIP(hipStreamSynchronize(c->stream)); // the host vectors go out of scope
    c->bvh_nodes_host = bvh.nodes;
    c->entry_valid = false;
    c->bvh_wide_f4 = 0;
    if (R1_BVH4 && bvh.nodes.size() >= 16 && bvh.nodes.size() / 16 <= R1_NODES_LDS_MAX && !bvh.pad_local)
    {
        std::vector<float> wide;
        c->bvh_wide_stack = build_wide(bvh.nodes, bvh.root_leaf, wide);
        if (c->bvh_wide_stack <= R1_BVH_STACK && wide.size() / 4 < 7u * 4096u)
        {
            if ((rc = ensure(c->bvh_wide, wide.size() * 4)))
                return rc;
            R1_HIP(hipMemcpy(c->bvh_wide.p, wide.data(), wide.size() * 4, hipMemcpyHostToDevice));
            c->bvh_wide_f4 = (uint32_t)(wide.size() / 4);
        }
        static const int print = (int)r1_knob("R1_BVH4_PRINT", 0);
        if (print)
            fprintf(stderr, "rays1: 4-wide table: %zu binary nodes -> %zu wide nodes, stack %d entries (binary: %d)\n", bvh.nodes.size() / 16, wide.size() / 28 - 1,
                    c->bvh_wide_stack, bvh.max_depth);
    }
    c->n_bvh_nodes = (uint32_t)(bvh.nodes.size() / 16);
    c->n_bvh_leaves = bvh.n_leaves;
    c->bvh_depth = bvh.max_depth;
    for (int k = 0; k < 3; ++k)
        c->bvh_centre[k] = bvh.centre[k];
    c->bvh_pad_local = bvh.pad_local;
    c->bvh_root_leaf = bvh.root_leaf;
    for (int &o : c->occupancy)
        o = 0; // the tree kernels' LDS footprint follows the tree (depth of the traversal stack, size of the node table)
    c->n_groups = ng;
    c->n_multi = n_multi;

    c->n_active = na;
    c->n_sweep = ns;
    c->n_padded_scene = s->count;
    memcpy(c->cam.origin, cam->origin, 12);
    memcpy(c->cam.lower_left, cam->lower_left, 12);
    memcpy(c->cam.horizontal, cam->horizontal, 12);
    memcpy(c->cam.vertical, cam->vertical, 12);
    memcpy(c->cam.u, cam->u, 12);
    memcpy(c->cam.v, cam->v, 12);
    c->cam.lens_radius = cam->lens_radius;
    for (int k = 0; k < 9; ++k)
        c->src_f32[k].assign(src[k], src[k] + s->count);
    c->src_mat.assign(s->mat_type, s->mat_type + s->count);
    c->src_cam = *cam;
    c->have_scene = true;
    c->default_variant = c->default_variant_tp = 4;
    return R1_OK;
}

// ---- per-frame setup ------------------------------------------------------------------------------

static bool same_tiling(const r1_params &a, const r1_params &b)
{
    return a.width == b.width && a.height == b.height && a.spp == b.spp && a.tile_w == b.tile_w && a.tile_h == b.tile_h &&
           a.shard == b.shard && a.num_shards == b.num_shards;
}

static R1FastDiv make_div(uint32_t d)
{
    R1FastDiv r;
    uint32_t sh = 0;
    while ((2u << sh) <= d && sh < 31)
        ++sh; // floor(log2 d)
    if ((d & (d - 1)) == 0)
    {
        r.pow2 = 1, r.shift = sh, r.mul = 0;
    }
    else
    {
        r.pow2 = 0, r.shift = sh;
        r.mul = (uint32_t)((((uint64_t)1 << (32 + sh)) + d - 1) / d);
    }
    return r;
}

static int prepare_tiles(r1_context *c, const r1_params *p, int n_frames)
{
    if (c->tile_key_valid && same_tiling(c->tile_key, *p) && c->tile_frames == n_frames)
        return R1_OK;
    const int tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
    const int tiles_y = (p->height + p->tile_h - 1) / p->tile_h;
    const int total = tiles_x * tiles_y;
    const uint32_t local = total > p->shard ? (uint32_t)((total - p->shard + p->num_shards - 1) / p->num_shards) : 0u;
    const uint64_t full = (uint64_t)p->tile_w * p->tile_h * p->spp;
    if (full * local * (uint64_t)n_frames >= ((uint64_t)1 << 31) || (uint64_t)total * p->num_shards >= ((uint64_t)1 << 31))
    {
        r1_set_error("%d frame(s) of %dx%dx%d with %dx%d tiles exceed 2^31 sample slots per launch", n_frames, p->width, p->height, p->spp, p->tile_w,
                     p->tile_h);
        return R1_ELIMIT;
    }
    c->n_local_tiles = local;
    c->full = (uint32_t)full;
    c->total_samples = (uint32_t)(full * local * (uint64_t)n_frames); // the launch's queue: frame-major
    c->tile_key = *p;
    c->tile_frames = n_frames;
    c->tile_key_valid = true;
    return R1_OK;
}

// The context's counter allocation: [0, R1_COUNTER_BYTES) queue heads (set 0), ray count, diagnostic counters — the block the round-3
// kernels zero between frames; then R1_COUNTER_TAIL bytes: +0 the published ray count, +64 eight batch-argument slots, +1024 queue heads
// (set 1); then, per launch (R1_LAND): frame_rays[F] (uint64), frame_left[F] (uint32, padded), tile_cnt[F x local tiles] (uint32).
// Called by every entry point BEFORE it takes addresses inside the allocation (it may move when the launch needs more room).
static size_t land_frames_off() { return (size_t)R1_COUNTER_BYTES + R1_COUNTER_TAIL; }
static int ensure_counters(r1_context *c, const r1_params *p, int n_frames)
{
    int rc = prepare_tiles(c, p, n_frames);
    if (rc)
        return rc;
    const size_t F = (size_t)(n_frames > 0 ? n_frames : 1);
    const size_t tiles = F * (size_t)(c->n_local_tiles ? c->n_local_tiles : 1);
    const size_t need = land_frames_off() + ((F * 16 + 127) & ~(size_t)127) + tiles * 4 * R1_LAND_CNT_STRIDE;
    if (c->counters.p && need <= c->counters.cap)
        return R1_OK;
    R1_HIP(hipStreamSynchronize(c->stream)); // (a frame in flight on another stream is the caller's to order: one frame per context at a time)
    if ((rc = ensure(c->counters, need + need / 2)))
        return rc;
    R1_HIP(hipMemsetAsync(c->counters.p, 0, c->counters.cap, c->stream));
    R1_HIP(hipStreamSynchronize(c->stream));
    c->counters_clean = true;
    c->land_prev = false, c->land_armed = false, c->land_parity = 0;
    c->batch_args_slot = -1;
    return R1_OK;
}

// Frame batch of the throughput entry points: n_frames frames in one launch (r1_device.h R1TraceArgs::n_frames); frame f is
// written to d_out + f * out_stride and its uint64 ray count to d_out + f * out_stride + rays_offset.
struct Batch
{
    int n_frames = 1;
    uint32_t seed_stride = 0;
    size_t out_stride = 0, rays_offset = 0;
};

// ---- per-tile entry nodes for primary rays (R1_ENTRY) ---------------------------------------------------------
// All primary rays of a 32 x 32 tile leave a small lens disk through a small rectangle of the focal plane: a narrow beam.  For every
// tile of the launch the deepest node is found below which ALL of them stay, and a primary ray starts its walk there instead of at the
// root's inner child (bvh_advance's root step still tests the root's leaf — the ground and the big balls — and the inner child's box).
// Exactness: a subtree is left out only if no ray of the beam can pass the kernel's box test of its root, judged CONSERVATIVELY — the
// box inflated by the largest pad any of these rays gets (the pad makes the boxes conservative with respect to the reference's fp32
// sphere test, r1_bvh.cpp) plus a margin for the test's own rounding, against the four side planes of the beam pushed outwards by the
// lens radius — so the walk from the root would not have entered it either: same offers, same minimum, same pixels.
static void beam_planes(const R1DeviceCamera &cam, double s0, double s1, double t0, double t1, double n[4][3], double &d_plane)
{
    double q[4][3];
    const double ss[4] = {s0, s1, s1, s0}, tt[4] = {t0, t0, t1, t1};
    for (int k = 0; k < 4; ++k)
        for (int a = 0; a < 3; ++a)
            q[k][a] = (double)cam.lower_left[a] + ss[k] * cam.horizontal[a] + tt[k] * cam.vertical[a] - cam.origin[a]; // corner directions from the lens centre
    // side plane k contains the lens centre and corners k, k + 1; its normal points away from the opposite corner
    for (int k = 0; k < 4; ++k)
    {
        const double *a = q[k], *b = q[(k + 1) & 3], *c = q[(k + 2) & 3];
        double m[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const double len = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
        double sgn = (m[0] * c[0] + m[1] * c[1] + m[2] * c[2]) > 0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i)
            n[k][i] = len > 0 ? sgn * m[i] / len : 0.0;
    }
    // distance of the focal plane from the lens centre (along its normal): every target point is at least that far away
    double w[3] = {cam.horizontal[1] * (double)cam.vertical[2] - cam.horizontal[2] * (double)cam.vertical[1],
                   cam.horizontal[2] * (double)cam.vertical[0] - cam.horizontal[0] * (double)cam.vertical[2],
                   cam.horizontal[0] * (double)cam.vertical[1] - cam.horizontal[1] * (double)cam.vertical[0]};
    const double wl = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    d_plane = wl > 0 ? fabs((w[0] * q[0][0] + w[1] * q[0][1] + w[2] * q[0][2]) / wl) : 0.0;
}

// may any ray of the beam pass the (inflated) box {m, e}?  false only when one side plane has the whole box outside
static bool beam_may_hit(const R1DeviceCamera &cam, const double n[4][3], double d_plane, const double m[3], const double e[3])
{
    const double r = fabs((double)cam.lens_radius);
    double rel[3], far2 = 0;
    for (int a = 0; a < 3; ++a)
    {
        rel[a] = m[a] - cam.origin[a];
        const double f = fabs(rel[a]) + e[a];
        far2 += f * f;
    }
    // a point of a beam ray at parameter L (1 = the focal plane) lies within |1 - L| r of the cone from the lens CENTRE; inside the box
    // L <= (farthest corner + r) / (focal distance - r)
    const double lmax = d_plane > 2 * r ? (sqrt(far2) + r) / (d_plane - r) : 1e30;
    const double rho = r * std::max(1.0, lmax - 1.0);
    if (!(rho < 1e20))
        return true;
    for (int k = 0; k < 4; ++k)
    {
        double lo = 0;
        for (int a = 0; a < 3; ++a)
            lo += n[k][a] * rel[a] - fabs(n[k][a]) * e[a];
        if (lo > rho)
            return false;
    }
    return true;
}

// entry[j] for every tile j of the launch (frame-major as the queue; the frames of a batch share the camera): a child reference in the
// kernel's form (16-bit for the small-scene kernels); R1_BVH_DONE (all ones) where the beam misses the inner child altogether
static void compute_entries(const r1_context *c, const r1_params *p, int n_frames, bool ref16, std::vector<uint32_t> &out)
{
    const uint32_t nlt = c->n_local_tiles;
    out.assign((size_t)nlt * n_frames, 0xFFFFFFFFu);
    const size_t n_nodes = c->bvh_nodes_host.size() / 16;
    if (!c->bvh_root_leaf || n_nodes < 1 || nlt == 0)
        return;
    const float *N = c->bvh_nodes_host.data();
    auto child = [&](uint32_t node, int k) { uint32_t v; memcpy(&v, N + 16 * (size_t)node + 14 + k, 4); return v; };
    auto form = [&](uint32_t ref) { return ref16 && ref != 0xFFFFFFFFu ? (((ref >> 16) & 0xF000u) | (ref & 0x0FFFu)) : ref; }; // r1_ref16 (r1_trace.hpp)
    const uint32_t other = child(0, c->bvh_root_leaf == 1 ? 1 : 0);
    static const int entry_off = (int)r1_knob("R1_ENTRY_OFF", 0); // tuning: every tile starts at the root's inner child
    if ((other & 0x80000000u) || entry_off)
    {
        out.assign((size_t)nlt * n_frames, form(other));
        return;
    }
    const int tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
    const double r = fabs((double)c->cam.lens_radius);
    double oc2 = 0, o1 = 0;
    for (int a = 0; a < 3; ++a)
        oc2 += ((double)c->cam.origin[a] - c->bvh_centre[a]) * ((double)c->cam.origin[a] - c->bvh_centre[a]), o1 += fabs((double)c->cam.origin[a]);
    const double r2max = (sqrt(oc2) + r) * (sqrt(oc2) + r); // largest |o - C|^2 of a primary ray's origin
    long depth_sum = 0, missed = 0;
    for (uint32_t lt = 0; lt < nlt; ++lt)
    {
        const int tile = p->shard + (int)lt * p->num_shards;
        const int x0 = (tile % tiles_x) * p->tile_w, y0 = (tile / tiles_x) * p->tile_h;
        const int tw = std::min(p->tile_w, p->width - x0), th = std::min(p->tile_h, p->height - y0);
        double n[4][3], d_plane;
        // (x + jitter) / W with jitter in [0, 1); a hair of slack for the fp32 products of Camera::getRay
        beam_planes(c->cam, (x0 - 1e-3) / p->width, (x0 + tw + 1e-3) / p->width, (y0 - 1e-3) / p->height, (y0 + th + 1e-3) / p->height, n, d_plane);
        uint32_t x = other; // descend while exactly one child can be met and it is an inner node
        int levels = 0;
        for (;; ++levels)
        {
            const float *q = N + 16 * (size_t)x;
            const double A = q[12], K = q[13];
            int hits = 0, which = -1;
            for (int k = 0; k < 2; ++k)
            {
                const double m[3] = {q[0 + k], q[2 + k], q[4 + k]};
                double pad;
                if (c->bvh_pad_local)
                {
                    double s2 = 0; // |m0 + m1 - 2 o|^2 at its largest over the lens disk
                    const double sv[3] = {(double)q[0] + q[1] - 2.0 * c->cam.origin[0], (double)q[2] + q[3] - 2.0 * c->cam.origin[1], (double)q[4] + q[5] - 2.0 * c->cam.origin[2]};
                    for (int a = 0; a < 3; ++a)
                        s2 += sv[a] * sv[a];
                    pad = A * (sqrt(s2) + 2 * r) * (sqrt(s2) + 2 * r) + K;
                }
                else
                    pad = A * r2max + K;
                // margin: the slab test's rounding (relative 2^-20 of the coordinates involved is generous) and v_rcp_f32's 1 ulp
                const double margin = 1e-5 * (fabs(m[0]) + fabs(m[1]) + fabs(m[2]) + o1 + 1.0) + 1e-4;
                const double e[3] = {q[6 + k] * 1.0001 + pad * 1.001 + margin, q[8 + k] * 1.0001 + pad * 1.001 + margin, q[10 + k] * 1.0001 + pad * 1.001 + margin};
                if (beam_may_hit(c->cam, n, d_plane, m, e))
                    ++hits, which = k;
            }
            if (hits == 0)
            {
                x = 0xFFFFFFFFu; // nothing of the lattice can be met: the walk is over after the root step
                break;
            }
            if (hits == 2)
                break;
            const uint32_t cref = child(x, which);
            x = cref;
            if (cref & 0x80000000u)
                break; // a leaf: tested directly
        }
        depth_sum += levels + (x == 0xFFFFFFFFu ? 1 : 0), missed += x == 0xFFFFFFFFu;
        for (int f = 0; f < n_frames; ++f)
            out[(size_t)f * nlt + lt] = form(x);
    }
    static const int print = (int)r1_knob("R1_ENTRY_PRINT", 0);
    if (print)
        fprintf(stderr, "rays1: entry nodes: %u tiles, %.2f levels below the root's inner child on average, %ld tiles whose beam misses it\n", nlt,
                (double)depth_sum / nlt, missed);
}

// Where the caller finally wants the frame, if the device can write there (page-locked host memory): launches that resolve their own
// tiles (R1_LAND) store the pixels and the count there directly and set `used`; the entry point then enqueues no copy.
struct Landing
{
    void *out = nullptr;  // device address of the caller's pixels (row-major image, or the frame records of a batch)
    void *rays = nullptr; // ... of its ray count (single frames)
    bool used = false;
};

// device address of page-locked host memory (r1_host_alloc, hipHostMalloc, hipHostRegister), or null for anything else
static void *mapped_host(const void *ptr)
{
    if (!ptr)
        return nullptr;
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess)
    {
        (void)hipGetLastError(); // ordinary (pageable) memory is not an error here
        return nullptr;
    }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}

// Enqueues the frame (trace + resolve) on `st`.  d_out / d_rays are device addresses; d_rays == NULL stands for the context's own
// count word (counters + R1_COUNTER_BYTES; the allocation may move in here, so callers take that address afterwards).
static int enqueue_frame(r1_context *c, const r1_params *p, void *d_out, int block_layout, void *d_rays, hipStream_t st,
                         bool throughput_mode, const Batch *batch, Landing *landing)
{
    const int n_frames = batch ? batch->n_frames : 1;
    if (!c->have_scene)
    {
        r1_set_error("no scene set (call r1_set_scene first)");
        return R1_EINVAL;
    }
    int rc = r1_params_check(p);
    if (rc)
        return rc;
    // kernel selection.  DEFAULT = the box tree (a property of the build, see above); PREFILTER always forces the exhaustive
    // sweep, BVH always the tree; all of them produce the same pixels.
    int variant = 2;
    switch (p->variant)
    {
    case R1_VARIANT_REFERENCE: variant = 1; break;
    case R1_VARIANT_STATS: variant = 3; break;
    case R1_VARIANT_BVH: variant = 4; break;
    case R1_VARIANT_BVH_STATS: variant = 5; break;
    case R1_VARIANT_WAVEFRONT: variant = 6; break;
    case R1_VARIANT_DEFAULT: variant = throughput_mode ? c->default_variant_tp : c->default_variant; break;
    default: variant = 2; break;
    }
    R1_HIP(hipSetDevice(c->device));
    if ((rc = ensure_counters(c, p, n_frames)))
        return rc;
    if (!d_rays)
        d_rays = (char *)c->counters.p + R1_COUNTER_BYTES;
    // kernel mode: the host-returning entry points run in latency mode, the throughput entry point with few long-lived
    // waves per frame — per-sample records + r1_resolve_kernel either way, unless r1_set_pixel_mode chose PIXEL mode
    // for the throughput entry point (a lane owns a pixel: no sample records, no resolve launch, ~10 % slower)
    // big-scene kernels: > 1023 hittable spheres (10-bit hit indices), or — tree kernels — a node table too large for LDS, or a tree whose
    // pad is measured per node (small spheres: only the kernels that walk the table in global memory carry that arm, bvh_advance)
    const int big_scene_ = (c->n_active > R1_MAX_ACTIVE_10BIT || ((variant == 4 || variant == 5) && (c->n_bvh_nodes > R1_NODES_LDS_MAX || c->bvh_pad_local))) ? 1 : 0;
    static const int tp_mode_env = (int)r1_knob("R1_TP_MODE", -1); // tuning experiments
    const int tp_mode = c->pixel_mode ? 2 : (tp_mode_env >= 0 && tp_mode_env <= 2 ? tp_mode_env : 0);
    const int mode = variant == 6 ? 0 : r1_trace_mode(variant, big_scene_, throughput_mode ? tp_mode : 1);
    const bool pixel_mode = mode == 2;
    if (batch && (mode != 0 || variant == 6 || variant == 3 || variant == 5 || variant == 1))
    {
        r1_set_error("frame batches run through the throughput kernels only (no PIXEL mode, no diagnostic / reference-form / wavefront variant)");
        return R1_EINVAL;
    }
    if (batch && !(R1_LAND && variant == 4))
    {
        // partial ray counts of the resolve launch: one uint64 per (tile of the batch, workgroup column)
        const size_t cols = ((size_t)p->tile_w * p->tile_h + 255) / 256;
        if ((rc = ensure(c->batch_rays, (size_t)n_frames * (c->n_local_tiles ? c->n_local_tiles : 1) * cols * 8)))
            return rc;
    }
    // tiles resolved inside the trace kernel: the product kernels' launches; the diagnostic builds, the reference-form
    // sweep, the wavefront variant and PIXEL mode keep the round-3 form (records + r1_resolve_kernel, or no records at all)
    const bool land = variant == 4 && R1_LAND_MODE(mode) && c->total_samples > 0;
    if (!pixel_mode)
    {
        const size_t want = (size_t)(c->total_samples ? c->total_samples : 1) * 16;
        const bool fresh = !c->samples.p || c->samples.cap < want;
        if ((rc = ensure(c->samples, want)))
            return rc;
        if (fresh) // a record is recognised by its launch's tag: fresh memory must not carry one by accident
            R1_HIP(hipMemsetAsync(c->samples.p, 0, c->samples.cap, st));
    }

    R1TraceArgs a;
    memset(&a, 0, sizeof(a));
    a.scene.sweep = (const float4 *)c->sweep.p;
    a.scene.exact = (const float4 *)c->exact.p;
    a.scene.shade = (const float4 *)c->shade.p;
    a.scene.mat = (const float4 *)c->mat.p;
    a.scene.members = (const uint32_t *)c->members.p;
    a.scene.exact_g = (const float4 *)c->exact_g.p;
    a.scene.n_active = c->n_active;
    a.scene.n_sweep = c->n_sweep;
    a.scene.n_multi = c->n_multi;
    a.scene.bvh_nodes = (const float4 *)c->bvh_nodes.p;
    a.scene.bvh_prims = (const float4 *)c->bvh_prims.p;
    a.scene.bvh_ids = (const uint32_t *)c->bvh_ids.p;
    for (int k = 0; k < 3; ++k)
        a.scene.bvh_centre[k] = c->bvh_centre[k];
    a.scene.bvh_pad_local = (uint32_t)c->bvh_pad_local;
    a.scene.bvh_root_leaf = (uint32_t)c->bvh_root_leaf;
    a.cam = c->cam;
    a.width = p->width, a.height = p->height, a.spp = p->spp, a.max_bounces = p->max_bounces;
    a.seed = p->seed;
    a.inv_w = 1.0f / p->width;  // Vec3 inv_image_size(1.0f / td.image_w, 1.0f / td.image_h, 0) rayweek1.cpp:746
    a.inv_h = 1.0f / p->height;
    a.tile_w = p->tile_w, a.tile_h = p->tile_h;
    a.tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
    a.shard = p->shard, a.num_shards = p->num_shards;
    a.n_local_tiles = c->n_local_tiles;
    a.batch = nullptr;
    if (batch && n_frames > 1)
    {
        // the batch's numbers, in the context's counter allocation behind the published ray count (stream-ordered upload:
        // the previous launch through this context has finished reading its copy by the time this one is written)
        R1BatchArgs ba;
        ba.n_frames = (uint32_t)n_frames, ba.seed_stride = batch->seed_stride;
        ba.div_tiles = make_div(c->n_local_tiles ? c->n_local_tiles : 1u), ba.n_local_tiles = c->n_local_tiles;
        static_assert(sizeof(R1BatchArgs) == 24, "r1_launch_put6 writes the six words of R1BatchArgs");
        if (c->batch_args_slot < 0 || c->batch_args_stream != st || memcmp(&ba, &c->batch_args_last, sizeof(ba)) != 0)
        {
            // a new slot, so that a launch still reading the previous numbers is not disturbed; written in stream order by a launch of its own
            c->batch_args_slot = (c->batch_args_slot + 1) & 7;
            c->batch_args_last = ba, c->batch_args_stream = st;
            R1_HIP(r1_launch_put6((char *)c->counters.p + R1_COUNTER_BYTES + 64 + 32 * c->batch_args_slot, (const uint32_t *)&ba, st));
        }
        a.batch = (const R1BatchArgs *)((char *)c->counters.p + R1_COUNTER_BYTES + 64 + 32 * c->batch_args_slot);
    }
    a.full = c->full;
    a.div_full = make_div(c->full);
    a.div_spp = make_div((uint32_t)p->spp);
    a.div_tw = make_div((uint32_t)p->tile_w);
    a.div_tx = make_div((uint32_t)a.tiles_x);
    a.total_samples = c->total_samples;
    a.queue = (uint32_t *)((char *)c->counters.p + 1024);
    a.nq = 1;
    {
        static const int coop_env = (int)r1_knob("R1_COOP_LANES", -1);
        a.coop_lanes = coop_env >= 0 ? (uint32_t)coop_env : R1_COOP_LANES;
    }
    a.bvh_entry = nullptr, a.entry_lds = 0;
    if (R1_ENTRY && (variant == 4 || variant == 5) && c->bvh_root_leaf && c->n_local_tiles)
    {
        // per-tile entry nodes of the primary rays (compute_entries): once per (scene, camera, tiling, frames of the launch, reference form)
        const int form = big_scene_ ? 2 : 1;
        if (!c->entry_valid || c->entry_frames != n_frames * 4 + form || !same_tiling(c->entry_key, *p))
        {
            std::vector<uint32_t> tab;
            compute_entries(c, p, n_frames, !big_scene_, tab);
            R1_HIP(hipDeviceSynchronize()); // (launches still reading the previous table: a change of tiling is rare)
            if ((rc = ensure(c->bvh_entry, tab.size() * 4)))
                return rc;
            R1_HIP(hipMemcpy(c->bvh_entry.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
            c->entry_valid = true, c->entry_frames = n_frames * 4 + form, c->entry_key = *p;
        }
        a.bvh_entry = (const uint32_t *)c->bvh_entry.p;
        static const int entry_lds_env = (int)r1_knob("R1_ENTRY_LDS", 1); // tuning: 0 = the table stays in global memory
        a.entry_lds = (!big_scene_ && !batch && entry_lds_env && c->n_local_tiles <= R1_ENTRY_LDS_MAX) ? c->n_local_tiles : 0u;
    }
    a.samples = (float4 *)c->samples.p;
    // The frame's last launch (resolve) publishes the ray count and zeroes the counter block for the next frame, which
    // saves the two memset launches in front of every frame (they cost nothing to execute and ~10 us each to dispatch:
    // a rank of an 8-GPU run renders its share of a frame in 140 us).  Frames without a resolve launch, and the diagnostic
    // builds, whose counters are read back afterwards, count into the caller's word and clear with memsets.
    const bool fused_clear = !land && !pixel_mode && c->n_local_tiles && c->total_samples && variant != 3 && variant != 5;
    a.num_rays = fused_clear ? (unsigned long long *)((char *)c->counters.p + 32) : (unsigned long long *)d_rays;
    a.stats = (variant == 3 || variant == 5) ? (unsigned long long *)((char *)c->counters.p + 128) : nullptr;

    // BIG kernels: 32-bit hit indices, the attenuation stack in a global workspace (the packed
    // LDS stack holds 10-bit indices) and the tree's node table through the vector L1.  Tried for the
    // tree kernel on small scenes too (more workgroups per CU): 15 % slower.  The small-scene tree
    // kernels keep the first 3 * R1_STACK_LDS_WORDS stack entries in LDS and use the workspace beyond.
    const int big = big_scene_;
    a.bvh_depth = c->bvh_depth > 0 ? c->bvh_depth : 1;
    // the workgroups' LDS copy of the node table: all of it for small scenes, the breadth-first top for big ones
    static const int big_top_env = (int)r1_knob("R1_BIG_TOP", R1_BVH_TOP_NODES); // tuning experiments
    // (big scenes: at least node 0 — the walk's root step reads it from the LDS copy, whatever the tuning knob says)
    a.bvh_lds_f4 = !(variant == 4 || variant == 5) ? 0u : (!big ? 4u * c->n_bvh_nodes : 4u * std::min<uint32_t>(c->n_bvh_nodes, (uint32_t)std::max(1, big_top_env)));
    a.bvh_wide = nullptr;
    if (R1_BVH4 && (variant == 4 || variant == 5) && !big)
    {
        if (!c->bvh_wide_f4)
        {
            r1_set_error("this build walks 4-wide nodes (R1_BVH4) and the scene's tree has no such table");
            return R1_EINVAL;
        }
        a.bvh_wide = (const float4 *)c->bvh_wide.p, a.bvh_lds_f4 = c->bvh_wide_f4, a.bvh_depth = c->bvh_wide_stack;
    }
    const int occ_slot = variant + 8 * big + 16 * mode;
    if (c->occupancy[occ_slot] == 0)
        R1_HIP(r1_trace_occupancy(variant, big, mode,
                                  (variant == 4 || variant == 5) ? (size_t)a.bvh_depth * R1_BLOCK * (big ? 4 : 2) + (size_t)a.bvh_lds_f4 * 16 + R1_ENTRY_LDS_BYTES(a.entry_lds) : 0,
                                  &c->occupancy[occ_slot]));
    int per_cu = c->occupancy[occ_slot];
    if (per_cu < 1)
        per_cu = 1;
    if (per_cu > 8)
        per_cu = 8;
    static const int per_cu_env = (int)r1_knob("R1_BLOCKS_PER_CU", 0); // tuning experiments
    if (per_cu_env > 0 && per_cu_env < per_cu)
        per_cu = per_cu_env;
    // Persistent grid.  Latency mode (the synchronous host entry points: one frame, the caller
    // waits): as many waves as fit, every lane at least one sample.  Throughput mode (the
    // device-resident entry point, frames in flight on several streams): a wave's lanes run dry
    // one by one at the end of its share (the longest bounce chain of 64 lanes is ~20 sweeps), so
    // a wave needs many times that much work to stay full — give every lane
    // >= R1_SAMPLES_PER_LANE samples and let the other frames fill the CUs a small frame leaves.
    long long blocks = (long long)c->cus * per_cu;
    static const long long spl_env = r1_knob("R1_SAMPLES_PER_LANE", R1_SAMPLES_PER_LANE);
    static const long long minb_env = r1_knob("R1_MIN_BLOCKS", R1_MIN_BLOCKS);
    long long needed = ((long long)c->total_samples + R1_BLOCK - 1) / R1_BLOCK;
    if (throughput_mode)
    {
        // few, long-lived workgroups per frame (>= R1_SAMPLES_PER_LANE samples per lane), but not fewer
        // than R1_MIN_BLOCKS while that still leaves R1_SAMPLES_PER_LANE_MIN samples per lane
        const long long spl = p->num_shards > 1 ? R1_SAMPLES_PER_LANE_SHARD : (spl_env > 0 ? spl_env : 1);
        const long long hi = ((long long)c->total_samples + R1_BLOCK * spl - 1) / (R1_BLOCK * spl);
        const long long lo = ((long long)c->total_samples + R1_BLOCK * R1_SAMPLES_PER_LANE_MIN - 1) / (R1_BLOCK * R1_SAMPLES_PER_LANE_MIN);
        needed = std::max(hi, std::min(minb_env, lo));
    }
    if (blocks > needed)
        blocks = needed;
    if (blocks < 1)
        blocks = 1;
    // Queue chunk per atomic: guided (remaining / (2 waves)) between chunk_min and chunk_max.  Large
    // chunks keep a wave on consecutive samples (coherent primary rays, whole sample-record lines)
    // and save atomics — measured at N = 1: 256 -> 1.227 ms, 1024 -> 1.197, 4096 -> 1.231 —
    // but they must stay small against a wave's share of the frame (8 shards: 1024 costs 12 %).
    {
        static const int chunk_max_env = (int)r1_knob("R1_CHUNK", 0), chunk_min_env = (int)r1_knob("R1_CHUNK_MIN", 0);
        const long long waves = blocks * (R1_BLOCK / 64);
        long long cm = (long long)c->total_samples / (waves * 12);
        cm = cm < R1_CHUNK ? R1_CHUNK : (cm > R1_CHUNK_BIG ? R1_CHUNK_BIG : cm);
        a.chunk_max = chunk_max_env > 0 ? (uint32_t)chunk_max_env : (uint32_t)cm;
        a.chunk_min = chunk_min_env > 0 ? (uint32_t)chunk_min_env : R1_CHUNK_MIN;
        if (a.chunk_min > a.chunk_max)
            a.chunk_min = a.chunk_max;
    }
    // Latency mode: every wave of the full grid takes one wave-full of samples per atomic (what a wave
    // still holds when the queue runs dry is the frame's tail: with 256-sample chunks the waves found
    // the queue empty over a span of 0.7 ms), which one counter cannot serve: sub-queues.
    if (pixel_mode)
    {
        // the queue holds the padded pixels of the shard's tiles; chunks in pixels (a wave holds 64 pixels at a time)
        const uint32_t tp = (uint32_t)(p->tile_w * p->tile_h);
        a.full = tp;
        a.div_full = make_div(tp);
        a.total_samples = c->n_local_tiles * tp;
        a.samples = (float4 *)d_out;
        a.block_layout = block_layout;
        a.inv_spp = (float)(1.0f / p->spp); // rayweek1.cpp:765
        const long long waves = blocks * (R1_BLOCK / 64);
        long long cm = (long long)a.total_samples / (waves * 12);
        cm = cm < 16 ? 16 : (cm > 128 ? 128 : cm);
        a.chunk_max = (uint32_t)cm;
        a.chunk_min = 8;
    }
    if (mode == 1 && land)
        a.chunk_max = a.chunk_min = 64u; // (the XCDs' cursors hand out the wave-fulls: eight lines instead of one, no sub-queues)
    else if (mode == 1)
    {
        static const int nq_env = (int)r1_knob("R1_NQ", 0), ch_env = (int)r1_knob("R1_CHUNK", 0);
        long long nq = nq_env > 0 ? nq_env : R1_SUBQUEUES;
        // a wave only ever pulls from its home sub-queue (r1_trace.hpp: home = (4 (block / 8) + wave) % nq), so every
        // sub-queue needs home waves: the full groups of 8 workgroups must cover all nq residues
        if (nq > 4 * (blocks / 8))
            nq = 4 * (blocks / 8);
        if (nq > (R1_COUNTER_BYTES - 1024) / 128)
            nq = (R1_COUNTER_BYTES - 1024) / 128;
        if (nq > 1)
        {
            a.nq = (uint32_t)nq;
            a.chunk_max = a.chunk_min = ch_env > 0 ? (uint32_t)ch_env : 64u;
        }
    }
    if (land)
    {
        // Launches alternate between two sets of queue heads and wave counts; workgroup 0 zeroes the set the launch before used, which
        // nobody touches any more (a workgroup that starts late still asks its own queue for work after the frame's last tile has been
        // summed, so a launch cannot clear its own).  After anything else has run through this context both sets (and the round-3 block) are cleared here.
        if (!c->land_prev)
        {
            R1_HIP(hipMemsetAsync(c->counters.p, 0, R1_COUNTER_BYTES, st));
            R1_HIP(hipMemsetAsync((char *)c->counters.p + R1_COUNTER_BYTES + 1024, 0, R1_COUNTER_BYTES - 1024, st)); // (not the batch-argument slots in front of it)
            c->land_parity = 1;
        }
        c->land_parity ^= 1;
        char *const set0 = (char *)c->counters.p + 1024, *const set1 = (char *)c->counters.p + R1_COUNTER_BYTES + 1024;
        a.queue = (uint32_t *)(c->land_parity ? set1 : set0);
        a.land.clear_heads = (uint32_t *)(c->land_parity ? set0 : set1);
        a.land.clear_count = (R1_COUNTER_BYTES - 1024) / 128;
        static_assert((R1_COUNTER_BYTES - 1024) / 128 <= R1_BLOCK && R1_COUNTER_TAIL >= 1024 + (R1_COUNTER_BYTES - 1024), "the second set of queue heads fits the tail");
        // the launch's generation tags its sample records (1 .. 2^24 - 1; on wrap-around the records are wiped)
        c->land_gen = (c->land_gen + 1) & 0xFFFFFFu;
        if (c->land_gen == 0)
        {
            R1_HIP(hipMemsetAsync(c->samples.p, 0, c->samples.cap, st));
            c->land_gen = 1;
        }
        a.land_tag = c->land_gen << 8;
        a.land_res = 1; // (a landing launch: r1_launch_trace checks that the kernel and the launch agree)
        unsigned long long *frame_rays = (unsigned long long *)((char *)c->counters.p + land_frames_off());
        uint32_t *frame_left = (uint32_t *)(frame_rays + n_frames);
        a.land_cnt = (uint32_t *)((char *)frame_rays + (((size_t)n_frames * 16 + 127) & ~(size_t)127)); // (every countdown on a 128-byte line of its own)
        if (!c->land_armed || c->land_frames != n_frames || !same_tiling(c->land_key, *p))
        {
            R1_HIP(r1_launch_land_arm(a.land_cnt, frame_rays, frame_left, (uint32_t)n_frames, c->n_local_tiles, p->width, p->height, p->spp, p->tile_w, p->tile_h,
                                      a.tiles_x, p->shard, p->num_shards, st));
            c->land_armed = true, c->land_frames = n_frames, c->land_key = *p;
        }
        if (landing && landing->out && (batch || landing->rays))
        {
            d_out = landing->out;
            if (!batch)
                d_rays = landing->rays;
            landing->used = true;
        }
        a.land.out = (uint8_t *)d_out;
        a.land.rays_dst = (unsigned long long *)d_rays;
        a.land.out_stride = batch ? batch->out_stride : 0;
        a.land.rays_offset = batch ? batch->rays_offset : 0;
        a.land.rays_in_out = batch ? 1u : 0u;
        a.land.frame_rays = frame_rays, a.land.frame_left = frame_left;
        a.land.n_frames = (uint32_t)n_frames;
        a.land.block_layout = (uint32_t)block_layout;
        a.land.inv_spp = (float)(1.0f / p->spp); // rayweek1.cpp:765
        a.land.error = c->host_word_dev ? (uint32_t *)(c->host_word_dev + 1) : nullptr;
        {
            // every wave's list of tiles: a row as long as the launch has tiles (the cursors only move forward: a wave meets a tile at
            // most once; with uneven residency — twenty frames in flight, a frame's first workgroups do most of its work — a
            // shorter list overflowed at 250 spp)
            const size_t tiles_all = (size_t)c->n_local_tiles * n_frames;
            const size_t bytes = (size_t)blocks * (R1_BLOCK / 64) * tiles_all * 4;
            if (bytes > ((size_t)2 << 30))
            {
                r1_set_error("frames in flight: %zu tiles x %lld waves need %zu MB of tile lists; render this frame synchronously or in shards", tiles_all,
                             (long long)blocks * (R1_BLOCK / 64), bytes >> 20);
                return R1_ELIMIT;
            }
            if ((rc = ensure(c->land_spill, bytes)))
                return rc;
            a.land.owed_spill = (uint32_t *)c->land_spill.p;
            a.land.spill_stride = (uint32_t)tiles_all;
        }
        a.num_rays = nullptr;
    }
    hipEvent_t e0 = c->ev0, e1 = c->ev1, e2 = c->ev2;
    if (c->ring_on && c->ring_frames > 0)
    {
        const int slot = c->ring_used < c->ring_frames ? c->ring_used : c->ring_frames - 1;
        e0 = c->ring[3 * slot], e1 = c->ring[3 * slot + 1], e2 = c->ring[3 * slot + 2];
        if (c->ring_used < c->ring_frames)
            ++c->ring_used;
    }
    if (big || (R1_STACK_LDS_WORDS < R1_STACK_WORDS && (variant == 4 || variant == 5)))
    {
        // sized for the largest grid of this kernel (not this frame's): a frame with a bigger grid must not reallocate
        // (sized for the build that keeps the fewest words in LDS: the latency / diagnostic builds keep R1_STACK_LDS_WORDS, the throughput builds R1_STACK_LDS_WORDS_TP)
        const size_t entries = big ? R1_STACK_ENTRIES : R1_STACK_ENTRIES - 3 * (R1_STACK_LDS_WORDS < R1_STACK_LDS_WORDS_TP ? R1_STACK_LDS_WORDS : R1_STACK_LDS_WORDS_TP);
        const size_t max_blocks = std::max((size_t)blocks, (size_t)c->cus * (size_t)per_cu);
        if ((rc = ensure(c->gstack, entries * max_blocks * R1_BLOCK * 4)))
            return rc;
        a.gstack = (uint32_t *)c->gstack.p;
    }
    if (!land && !c->counters_clean)
        R1_HIP(hipMemsetAsync(c->counters.p, 0, R1_COUNTER_BYTES, st));
    c->counters_clean = false;
    if (variant == 3 || variant == 5)
    {
        c->wave_log_waves = (uint32_t)blocks * (R1_BLOCK / 64);
        if ((rc = ensure(c->wave_log, (size_t)c->wave_log_waves * 32)))
            return rc;
        R1_HIP(hipMemsetAsync(c->wave_log.p, 0, (size_t)c->wave_log_waves * 32, st));
        c->wave_log_ptr = (unsigned long long)c->wave_log.p;
        R1_HIP(hipMemcpyAsync((char *)c->counters.p + 128 + 16 * 8, &c->wave_log_ptr, 8, hipMemcpyHostToDevice, st));
    }
    if (!fused_clear && !land)
        R1_HIP(hipMemsetAsync(d_rays, 0, 8, st));
    R1_HIP(hipEventRecord(e0, st));
    if (c->total_samples && variant != 6)
        R1_HIP(r1_launch_trace(&a, variant, big, mode, (int)blocks, st));
    if (c->total_samples && variant == 6)
    {
        // wavefront variant: path state, per-level queues and the attenuation stack live in HBM
        const size_t n = c->total_samples;
        if (n > ((size_t)1 << 24))
        {
            r1_set_error("R1_VARIANT_WAVEFRONT keeps every path of the frame in memory: %zu sample slots > 2^24", n);
            return R1_ELIMIT;
        }
        if ((rc = ensure(c->wf_paths, 3 * n * 16)) || (rc = ensure(c->wf_hits, n * 8)) || (rc = ensure(c->wf_queue, 2 * n * 4)) ||
            (rc = ensure(c->wf_counts, (R1_STACK_ENTRIES + 2) * 4)) || (rc = ensure(c->gstack, (size_t)R1_STACK_ENTRIES * n * 4)))
            return rc;
        R1WaveArgs w;
        memset(&w, 0, sizeof(w));
        w.t = a;
        w.t.gstack = (uint32_t *)c->gstack.p;
        w.paths = (float4 *)c->wf_paths.p;
        w.hits = (float2 *)c->wf_hits.p;
        w.queue[0] = (uint32_t *)c->wf_queue.p;
        w.queue[1] = (uint32_t *)c->wf_queue.p + n;
        w.counts = (uint32_t *)c->wf_counts.p;
        w.n_paths = (uint32_t)n;
        long long wb = (long long)((n + R1_BLOCK - 1) / R1_BLOCK);
        if (wb > (long long)c->cus * 8)
            wb = (long long)c->cus * 8;
        R1_HIP(hipMemsetAsync(c->wf_counts.p, 0, (R1_STACK_ENTRIES + 2) * 4, st));
        R1_HIP(r1_launch_wavefront(&w, (int)wb, st));
        blocks = wb;
    }
    R1_HIP(hipEventRecord(e1, st));

    R1ResolveArgs r;
    memset(&r, 0, sizeof(r));
    r.samples = (const float4 *)c->samples.p;
    r.full = c->full;
    r.width = p->width, r.height = p->height, r.spp = p->spp;
    r.tile_w = p->tile_w, r.tile_h = p->tile_h, r.tiles_x = a.tiles_x;
    r.shard = p->shard, r.num_shards = p->num_shards;
    r.n_local_tiles = c->n_local_tiles;
    r.inv_spp = (float)(1.0f / p->spp); // rayweek1.cpp:765
    r.out = (uint8_t *)d_out;
    r.block_layout = block_layout;
    r.n_frames = (uint32_t)n_frames;
    if (batch)
    {
        r.out_stride = batch->out_stride, r.rays_offset = batch->rays_offset;
        r.frame_rays = (unsigned long long *)c->batch_rays.p;
    }
    if (fused_clear)
    {
        r.rays_src = (const unsigned long long *)((char *)c->counters.p + 32);
        r.rays_dst = (unsigned long long *)d_rays;
        r.reset = (uint32_t *)c->counters.p;
    }
    static const int resolve_rows = (int)r1_knob("R1_RESOLVE_ROWS", R1_RESOLVE_ROWS_TP); // tuning experiments
    if (land)
        ; // the trace launch resolved its tiles itself
    else if (c->n_local_tiles && !pixel_mode)
        R1_HIP(r1_launch_resolve(&r, throughput_mode ? resolve_rows : 0, st));
    else if (batch) // a shard without tiles: its frames' counts are zero
        for (int f = 0; f < n_frames; ++f)
            R1_HIP(hipMemsetAsync((char *)d_out + (size_t)f * batch->out_stride + batch->rays_offset, 0, 8, st));
    c->counters_clean = fused_clear;
    c->land_prev = land;
    R1_HIP(hipEventRecord(e2, st));
    c->last0 = e0, c->last1 = e1, c->last2 = e2;
    c->timing_valid = true;

    c->info.blocks = (int32_t)blocks;
    c->info.tiles_in_kernel = land ? 1 : 0;
    c->info.threads_per_block = R1_BLOCK;
    c->info.spheres_active = (int32_t)c->n_active;
    c->info.spheres_padded = (int32_t)c->n_padded_scene;
    c->info.groups = (int32_t)c->n_groups;
    c->info.samples = c->total_samples;
    c->info.kernel = variant; // internal numbering = the public enum (DEFAULT resolved)
    c->info.bvh_nodes = (int32_t)c->n_bvh_nodes;
    c->info.bvh_leaves = (int32_t)c->n_bvh_leaves;
    c->info.bvh_depth = c->bvh_depth;
    return R1_OK;
}

// ---- public render entry points ---------------------------------------------------------------------

// a resolver of an earlier launch gave up waiting (R1_LAND_MAX_WAIT): the frames of that launch are not valid
static int land_check(r1_context *c)
{
    if (c->host_word && ((volatile uint32_t *)c->host_word)[2])
    {
        ((volatile uint32_t *)c->host_word)[2] = 0;
        r1_set_error("a launch did not resolve all of its tiles (a resolver workgroup gave up waiting): its frames are not valid");
        return R1_EHIP;
    }
    return R1_OK;
}

static int render_host(r1_context *c, const r1_params *p, uint8_t *rgb_out, uint64_t *num_rays_out, double *device_seconds_out,
                       float *samples_out)
{
    if (!c || !p || !rgb_out)
    {
        r1_set_error("r1_render: null argument");
        return R1_EINVAL;
    }
    int rc = r1_params_check(p);
    if (rc)
        return rc;
    if (samples_out && p->num_shards != 1)
    {
        r1_set_error("r1_render_samples needs num_shards == 1");
        return R1_EINVAL;
    }
    R1_HIP(hipSetDevice(c->device));
    const bool sharded = p->num_shards > 1;
    const size_t img_bytes = (size_t)p->width * p->height * 3;
    const size_t out_bytes = sharded ? r1_shard_block_bytes(p) : img_bytes;
    if ((rc = ensure(c->image, out_bytes + 64)))
        return rc;
    // the ray count: stored by the frame's last launch straight into the context's page-locked word (no second copy to
    // enqueue and wait for); the diagnostic builds count with atomics and keep a device word + copy
    const bool stats = p->variant == R1_VARIANT_STATS || p->variant == R1_VARIANT_BVH_STATS;
    const bool direct = !stats && c->host_word_dev;
    // a page-locked pixel buffer (r1_host_alloc) receives the tiles straight from the trace kernel's resolvers: no copy either
    Landing land_to;
    if (direct && !sharded)
        land_to.out = mapped_host(rgb_out), land_to.rays = c->host_word_dev;
    if ((rc = enqueue_frame(c, p, c->image.p, sharded ? 1 : 0, direct ? (void *)c->host_word_dev : nullptr, c->stream, false, nullptr, &land_to)))
        return rc;
    void *const d_rays = direct ? (void *)c->host_word_dev : (void *)((char *)c->counters.p + R1_COUNTER_BYTES); // (behind the block the frame's last launch zeroes)

    uint64_t rays = 0;
    if (!sharded)
    {
        if (!land_to.used)
            R1_HIP(hipMemcpyAsync(rgb_out, c->image.p, img_bytes, hipMemcpyDeviceToHost, c->stream));
        if (!direct)
            R1_HIP(hipMemcpyAsync(&rays, d_rays, 8, hipMemcpyDeviceToHost, c->stream));
        R1_HIP(hipStreamSynchronize(c->stream));
    }
    else
    {
        std::vector<uint8_t> block(out_bytes);
        R1_HIP(hipMemcpyAsync(block.data(), c->image.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
        if (!direct)
            R1_HIP(hipMemcpyAsync(&rays, d_rays, 8, hipMemcpyDeviceToHost, c->stream));
        R1_HIP(hipStreamSynchronize(c->stream));
        const int tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
        for (uint32_t lt = 0; lt < c->n_local_tiles; ++lt)
        {
            const int t = p->shard + (int)lt * p->num_shards;
            const int x0 = (t % tiles_x) * p->tile_w, y0 = (t / tiles_x) * p->tile_h;
            const int tw = p->tile_w < p->width - x0 ? p->tile_w : p->width - x0;
            const int th = p->tile_h < p->height - y0 ? p->tile_h : p->height - y0;
            for (int ly = 0; ly < th; ++ly)
                memcpy(rgb_out + ((size_t)(y0 + ly) * p->width + x0) * 3,
                       block.data() + ((size_t)lt * p->tile_h * p->tile_w + (size_t)ly * p->tile_w) * 3, (size_t)tw * 3);
        }
    }
    if ((rc = land_check(c)))
        return rc;
    if (direct)
        rays = *(volatile unsigned long long *)c->host_word;
    if (num_rays_out)
        *num_rays_out = rays;
    if (device_seconds_out)
    {
        float ms = 0;
        R1_HIP(hipEventElapsedTime(&ms, c->last0, c->last2)); // the events THIS frame recorded (a ring slot while r1_timing_begin is on)
        *device_seconds_out = ms * 1e-3;
    }
    if (samples_out)
    {
        // device order is [padded tile][sample][pixel in tile]; the ABI order is
        // ((y*width + x)*spp + s)
        std::vector<float> tmp((size_t)c->total_samples * 4);
        R1_HIP(hipMemcpyAsync(tmp.data(), c->samples.p, tmp.size() * 4, hipMemcpyDeviceToHost, c->stream));
        R1_HIP(hipStreamSynchronize(c->stream));
        const int tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
        for (uint32_t lt = 0; lt < c->n_local_tiles; ++lt)
        {
            const int x0 = ((int)lt % tiles_x) * p->tile_w, y0 = ((int)lt / tiles_x) * p->tile_h;
            const int tw = p->tile_w < p->width - x0 ? p->tile_w : p->width - x0;
            const int th = p->tile_h < p->height - y0 ? p->tile_h : p->height - y0;
            const float *src = tmp.data() + (size_t)lt * c->full * 4;
            const size_t tile_px = (size_t)p->tile_w * p->tile_h;
            for (int ly = 0; ly < th; ++ly)
                for (int lx = 0; lx < tw; ++lx)
                    for (int sm = 0; sm < p->spp; ++sm)
                    {
                        float *dst = samples_out + (((size_t)(y0 + ly) * p->width + (x0 + lx)) * p->spp + sm) * 4;
                        memcpy(dst, src + ((size_t)sm * tile_px + (size_t)(ly * p->tile_w + lx)) * 4, 16);
                        if (c->land_prev) // the launch tagged its records' ray-count words (R1_LAND): the ABI's word is the count alone
                        {
                            uint32_t wv;
                            memcpy(&wv, dst + 3, 4);
                            wv &= 255u;
                            memcpy(dst + 3, &wv, 4);
                        }
                    }
        }
    }
    return R1_OK;
}

extern "C" int r1_render(r1_context *c, const r1_params *p, uint8_t *rgb_out, uint64_t *num_rays_out, double *device_seconds_out)
{
    return render_host(c, p, rgb_out, num_rays_out, device_seconds_out, nullptr);
}

extern "C" int r1_render_samples(r1_context *c, const r1_params *p, uint8_t *rgb_out, uint64_t *num_rays_out, float *samples_out)
{
    if (!samples_out)
    {
        r1_set_error("r1_render_samples: null samples_out");
        return R1_EINVAL;
    }
    return render_host(c, p, rgb_out, num_rays_out, nullptr, samples_out);
}

// Pipelined form of r1_render (frames in flight, results on the HOST): the frame is enqueued with the throughput
// kernels on `hip_stream` (or the context's stream), followed by the copies of the row-major image and of the ray
// count into the caller's buffers.  Nothing is waited for: the buffers are valid once the stream is idle (r1_sync for
// the context's stream).  One frame per context at a time — a caller keeps K frames in flight with K contexts, as
// bench.py does.  Page-locked buffers (r1_host_alloc) let the copies overlap the other frames' kernels; pageable
// memory works but makes each copy wait for its frame.
extern "C" int r1_render_async(r1_context *c, const r1_params *p, uint8_t *rgb_out, uint64_t *num_rays_out, void *hip_stream)
{
    if (!c || !p || ((rgb_out == nullptr) != (num_rays_out == nullptr)))
    {
        r1_set_error("r1_render_async: null argument (rgb_out and num_rays_out are given together, or both NULL)");
        return R1_EINVAL;
    }
    int rc = r1_params_check(p);
    if (rc)
        return rc;
    if (p->num_shards != 1)
    {
        r1_set_error("r1_render_async renders whole frames (num_shards == 1); shards go through r1_render_shard_device");
        return R1_EINVAL;
    }
    R1_HIP(hipSetDevice(c->device));
    const size_t img_bytes = (size_t)p->width * p->height * 3;
    if ((rc = ensure(c->image, img_bytes + 64)))
        return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    // page-locked buffers (r1_host_alloc) receive the tiles and the count straight from the trace kernel's resolvers: nothing to copy
    Landing land_to;
    if (rgb_out && ((uintptr_t)num_rays_out & 7u) == 0)
        land_to.out = mapped_host(rgb_out), land_to.rays = mapped_host(num_rays_out);
    if ((rc = enqueue_frame(c, p, c->image.p, 0, nullptr, st, true, nullptr, &land_to)))
        return rc;
    if (rgb_out && !land_to.used) // (both NULL: the frame stays in the context's device buffers — a measurement aid, bench.py's value_device_resident)
    {
        R1_HIP(hipMemcpyAsync(rgb_out, c->image.p, img_bytes, hipMemcpyDeviceToHost, st));
        R1_HIP(hipMemcpyAsync(num_rays_out, (char *)c->counters.p + R1_COUNTER_BYTES, 8, hipMemcpyDeviceToHost, st));
    }
    return R1_OK;
}

// n_frames frames of the same scene, camera and size in ONE launch (frame f seeded params->seed + f * seed_stride): the
// persistent waves flow from one frame into the next, so the ramp and drain of a launch are paid once per batch.
// Whole frames (num_shards == 1): host_frames receives n_frames frame records (r1_frame_record_bytes each) with ONE copy.
extern "C" int r1_render_batch_async(r1_context *c, const r1_params *p, int32_t n_frames, uint32_t seed_stride, void *host_frames, void *hip_stream)
{
    if (!c || !p || n_frames < 1)
    {
        r1_set_error("r1_render_batch_async: bad argument");
        return R1_EINVAL;
    }
    int rc = r1_params_check(p);
    if (rc)
        return rc;
    if (p->num_shards != 1)
    {
        r1_set_error("r1_render_batch_async renders whole frames (num_shards == 1); shards go through r1_render_shard_device_batch");
        return R1_EINVAL;
    }
    R1_HIP(hipSetDevice(c->device));
    const size_t frame = r1_frame_record_bytes(p);
    if ((rc = ensure(c->image, frame * (size_t)n_frames)))
        return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    Batch b;
    b.n_frames = n_frames, b.seed_stride = seed_stride, b.out_stride = frame, b.rays_offset = frame - 8;
    Landing land_to;
    if (host_frames && ((uintptr_t)host_frames & 7u) == 0)
        land_to.out = mapped_host(host_frames);
    if ((rc = enqueue_frame(c, p, c->image.p, 0, nullptr, st, true, &b, &land_to)))
        return rc;
    if (host_frames && !land_to.used) // (NULL: the frames stay in the context's device buffer — a measurement aid)
        R1_HIP(hipMemcpyAsync(host_frames, c->image.p, frame * (size_t)n_frames, hipMemcpyDeviceToHost, st));
    return R1_OK;
}

// The same for one shard of n_frames frames: d_records receives n_frames records (r1_shard_record_bytes each: dense tile
// block + uint64 ray count), device memory — what a rank hands to ONE all-gather per batch.
extern "C" int r1_render_shard_device_batch(r1_context *c, const r1_params *p, int32_t n_frames, uint32_t seed_stride, void *d_records, void *hip_stream)
{
    if (!c || !p || !d_records || n_frames < 1 || ((uintptr_t)d_records & 7u))
    {
        r1_set_error("r1_render_shard_device_batch: bad argument (d_records must be 8-byte aligned)");
        return R1_EINVAL;
    }
    int rc = r1_params_check(p);
    if (rc)
        return rc;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const size_t record = r1_shard_record_bytes(p);
    Batch b;
    b.n_frames = n_frames, b.seed_stride = seed_stride, b.out_stride = record, b.rays_offset = record - 8;
    return enqueue_frame(c, p, d_records, 1, nullptr, st, true, &b);
}

extern "C" int r1_host_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0)
        return R1_EINVAL;
    *out = nullptr;
    if (r1_device_count() <= 0)
        return R1_ENODEVICE;
    R1_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return R1_OK;
}

extern "C" void r1_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

extern "C" int r1_render_shard_device(r1_context *c, const r1_params *p, void *d_block, void *d_num_rays, void *hip_stream)
{
    if (!c || !p || !d_block || !d_num_rays)
    {
        r1_set_error("r1_render_shard_device: null argument");
        return R1_EINVAL;
    }
    if ((uintptr_t)d_num_rays & 7u)
    {
        r1_set_error("r1_render_shard_device: d_num_rays must be 8-byte aligned (a uint64 the kernels store and add to atomically)");
        return R1_EINVAL;
    }
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return enqueue_frame(c, p, d_block, 1, d_num_rays, st, true);
}

extern "C" int r1_render_shard_device_once(r1_context *c, const r1_params *p, void *d_block, void *d_num_rays, void *hip_stream)
{
    if (!c || !p || !d_block || !d_num_rays)
    {
        r1_set_error("r1_render_shard_device_once: null argument");
        return R1_EINVAL;
    }
    if ((uintptr_t)d_num_rays & 7u)
    {
        r1_set_error("r1_render_shard_device_once: d_num_rays must be 8-byte aligned");
        return R1_EINVAL;
    }
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    return enqueue_frame(c, p, d_block, 1, d_num_rays, st, false);
}

static int assemble_common(r1_context *c, const r1_params *p, const void *d_blocks, size_t shard_stride_bytes, void *d_rgb, void *d_total_rays,
                           void *hip_stream, int n_frames = 1, size_t frame_in = 0, size_t frame_out = 0)
{
    if (!c || !p || !d_blocks || !d_rgb)
    {
        r1_set_error("r1_assemble_device: null argument");
        return R1_EINVAL;
    }
    int32_t total = 0, per = 0;
    int rc = r1_tile_count(p, &total, &per);
    if (rc)
        return rc;
    const size_t tight = (size_t)per * p->tile_w * p->tile_h * 3;
    if (shard_stride_bytes == 0)
        shard_stride_bytes = tight;
    if (shard_stride_bytes < tight)
    {
        r1_set_error("r1_assemble_device: shard stride %zu smaller than a shard block (%zu bytes)", shard_stride_bytes, tight);
        return R1_EINVAL;
    }
    if (d_total_rays && (((uintptr_t)d_total_rays | (uintptr_t)d_blocks | shard_stride_bytes | frame_in | frame_out) & 7u))
    {
        r1_set_error("r1_assemble_device_records: records and totals must be 8-byte aligned");
        return R1_EINVAL;
    }
    R1_HIP(hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const int tiles_x = (p->width + p->tile_w - 1) / p->tile_w;
    const size_t record = r1_shard_record_bytes(p);
    R1_HIP(r1_launch_assemble(d_blocks, d_rgb, p->width, p->height, p->tile_w, p->tile_h, tiles_x, p->num_shards, shard_stride_bytes, n_frames, frame_in,
                              frame_out, record - 8, d_total_rays ? (long long)((char *)d_total_rays - (char *)d_rgb) : 0, d_total_rays ? 1 : 0, st));
    return R1_OK;
}

extern "C" int r1_assemble_device_strided(r1_context *c, const r1_params *p, const void *d_blocks, size_t shard_stride_bytes, void *d_rgb,
                                          void *hip_stream)
{
    return assemble_common(c, p, d_blocks, shard_stride_bytes, d_rgb, nullptr, hip_stream);
}

extern "C" int r1_assemble_device_records(r1_context *c, const r1_params *p, const void *d_records, void *d_rgb, void *d_total_rays, void *hip_stream)
{
    if (!d_total_rays)
    {
        r1_set_error("r1_assemble_device_records: null d_total_rays");
        return R1_EINVAL;
    }
    return assemble_common(c, p, d_records, r1_shard_record_bytes(p), d_rgb, d_total_rays, hip_stream);
}

// Batches: d_gathered = what one all-gather of every shard's n_frames records returns, [shard][frame][record]; d_frames receives
// n_frames frame records (r1_frame_record_bytes each: row-major image, padded to 8 bytes, + the frame's uint64 ray count).
extern "C" int r1_assemble_device_records_batch(r1_context *c, const r1_params *p, int32_t n_frames, const void *d_gathered, void *d_frames,
                                                void *hip_stream)
{
    if (n_frames < 1 || !d_frames)
    {
        r1_set_error("r1_assemble_device_records_batch: bad argument");
        return R1_EINVAL;
    }
    const size_t record = r1_shard_record_bytes(p), frame = r1_frame_record_bytes(p);
    if (!record)
        return R1_EINVAL;
    return assemble_common(c, p, d_gathered, record * (size_t)n_frames, d_frames, (char *)d_frames + frame - 8, hip_stream, n_frames, record, frame);
}

extern "C" int r1_assemble_device(r1_context *c, const r1_params *p, const void *d_blocks, void *d_rgb, void *hip_stream)
{
    return r1_assemble_device_strided(c, p, d_blocks, 0, d_rgb, hip_stream);
}

extern "C" int r1_set_pixel_mode(r1_context *c, int32_t on)
{
    if (!c)
        return R1_EINVAL;
    c->pixel_mode = on != 0;
    return R1_OK;
}

extern "C" int r1_sync(r1_context *c)
{
    if (!c)
        return R1_EINVAL;
    R1_HIP(hipSetDevice(c->device));
    R1_HIP(hipStreamSynchronize(c->stream));
    return land_check(c);
}

extern "C" int r1_last_timing(r1_context *c, double *trace_kernel_ms, double *total_ms)
{
    if (!c || !c->timing_valid)
    {
        r1_set_error("r1_last_timing: nothing rendered yet");
        return R1_EINVAL;
    }
    R1_HIP(hipEventSynchronize(c->last2));
    float a = 0, b = 0;
    R1_HIP(hipEventElapsedTime(&a, c->last0, c->last1));
    R1_HIP(hipEventElapsedTime(&b, c->last0, c->last2));
    if (trace_kernel_ms)
        *trace_kernel_ms = a;
    if (total_ms)
        *total_ms = b;
    return R1_OK;
}

extern "C" int r1_last_stats(r1_context *c, uint64_t *out16)
{
    if (!c || !out16 || !c->counters.p)
        return R1_EINVAL;
    R1_HIP(hipSetDevice(c->device));
    R1_HIP(hipStreamSynchronize(c->stream));
    R1_HIP(hipMemcpyAsync(out16, (char *)c->counters.p + 128, 128, hipMemcpyDeviceToHost, c->stream));
    R1_HIP(hipStreamSynchronize(c->stream));
    return R1_OK;
}

extern "C" int r1_last_wave_log(r1_context *c, uint64_t *out, size_t cap_waves, uint32_t *waves)
{
    if (!c || !waves)
        return R1_EINVAL;
    *waves = c->wave_log_waves;
    if (!out)
        return R1_OK;
    if (cap_waves < c->wave_log_waves || !c->wave_log.p)
        return R1_EINVAL;
    R1_HIP(hipSetDevice(c->device));
    R1_HIP(hipStreamSynchronize(c->stream));
    R1_HIP(hipMemcpyAsync(out, c->wave_log.p, (size_t)c->wave_log_waves * 32, hipMemcpyDeviceToHost, c->stream));
    R1_HIP(hipStreamSynchronize(c->stream));
    return R1_OK;
}

extern "C" int r1_timing_begin(r1_context *c, int32_t max_frames)
{
    if (!c || max_frames < 1 || max_frames > 100000)
    {
        r1_set_error("r1_timing_begin: bad argument");
        return R1_EINVAL;
    }
    R1_HIP(hipSetDevice(c->device));
    while ((int)c->ring.size() < 3 * max_frames)
    {
        hipEvent_t e;
        R1_HIP(hipEventCreate(&e));
        c->ring.push_back(e);
    }
    c->ring_frames = max_frames;
    c->ring_used = 0;
    c->ring_on = true;
    return R1_OK;
}

extern "C" int r1_timing_end(r1_context *c, double *trace_ms_sum, double *total_ms_sum, int32_t *frames)
{
    if (!c || !c->ring_on)
    {
        r1_set_error("r1_timing_end without r1_timing_begin");
        return R1_EINVAL;
    }
    c->ring_on = false;
    double a = 0, b = 0;
    for (int i = 0; i < c->ring_used; ++i)
    {
        float x = 0, y = 0;
        R1_HIP(hipEventSynchronize(c->ring[3 * i + 2]));
        R1_HIP(hipEventElapsedTime(&x, c->ring[3 * i], c->ring[3 * i + 1]));
        R1_HIP(hipEventElapsedTime(&y, c->ring[3 * i], c->ring[3 * i + 2]));
        a += x, b += y;
    }
    if (trace_ms_sum)
        *trace_ms_sum = a;
    if (total_ms_sum)
        *total_ms_sum = b;
    if (frames)
        *frames = c->ring_used;
    return R1_OK;
}

extern "C" int r1_last_launch_info(r1_context *c, r1_launch_info *out)
{
    if (!c || !out)
        return R1_EINVAL;
    *out = c->info;
    return R1_OK;
}

// internal (tools/land_debug.py): the context's counter allocation as it is after the stream has drained
extern "C" int r1_debug_dump_counters(r1_context *c, void *out, size_t bytes, size_t *have)
{
    if (!c || !out || !c->counters.p)
        return R1_EINVAL;
    R1_HIP(hipSetDevice(c->device));
    R1_HIP(hipStreamSynchronize(c->stream));
    const size_t n = bytes < c->counters.cap ? bytes : c->counters.cap;
    R1_HIP(hipMemcpy(out, c->counters.p, n, hipMemcpyDeviceToHost));
    if (have)
        *have = n;
    return R1_OK;
}
