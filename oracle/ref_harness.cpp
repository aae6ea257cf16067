// oracle/ref_harness.cpp — TEST INFRASTRUCTURE, not product code.
//
// Drives the reference's OWN step13 functions (create_*_scene, myrand01_x4,
// Camera::getRay, color, render_tile, TileRenderScheduler) compiled from where they
// lie under /root/reference (the translation unit is #include'd by absolute path from
// the Makefile's -DREF_STEP13_TU; nothing is copied into this repo).  It exists to
//   (1) generate the golden fixtures under tests/golden/ (oracle/gen_golden.py),
//   (2) pin oracle/r1_oracle.c against the real reference, bit for bit,
//   (3) time the reference's own scheduler/render_tile as bench.py's
//       cpu_baseline {"kind": "reference"}.
// Only this container can build it (the GPU box has no /root/reference); the built
// binary lives in oracle/_ref/ (git-ignored, travels with gpurun).
//
// What the harness adds on top of the reference (and why):
//   * runtime width/height/spp: the reference bakes SCREEN_W/H/NUM_SAMPLES_PER_PIXEL
//     in as macros (src/common/common.h:19-28) but render_tile()/ThreadData carry them
//     as runtime fields (rayweek1.cpp:79-95, :722-782), so the harness fills ThreadData
//     itself and re-runs Camera::init (rayweek1.cpp:366-379) with the runtime aspect;
//   * per-sample seeding (include/rays1_seed.h) for the `samples`/`frame` commands.
// The per-scene camera constants below are restated from rayweek1.cpp:557-564,
// :589-596, :661-668 and self-checked against the reference's own camera at its
// native 1280x720 aspect on every run.

#include <stdio.h>
#include <stdlib.h>
#include <assert.h>
#include <float.h>
#include <ctime>
#include <mutex>
#include <thread>
#include <atomic>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <immintrin.h>
#include <chrono>
#include <mm_malloc.h>
#include <vector>
#include <string>

// Dielectric::_refIdx is private by `class` default access (rayweek1.cpp:461-463) and the
// scene dump needs it; every standard header the TU uses is already included above, so
// the keyword swap only touches the reference's own declarations.
#define class struct
#define main ref_main
#include REF_STEP13_TU
#undef main
#undef class

#include "../include/rays1_seed.h"

// ------------------------------------------------------------------------------------

struct SceneDesc
{
    const char *name;
    Scene *(*create)();
    float from[3], at[3];
    float focus, aperture;
};

static const SceneDesc g_scenes[3] = {
    {"small", create_small_scene, {2, 1, 2}, {0, 0, 0}, 5.0f, 0.1f},
    {"medium", create_medium_scene, {0, 2, 3}, {0, 0, 0}, 3.0f, 0.1f * 0.2f},
    {"large", create_large_scene, {3, 8, 15}, {0, 0, 0}, 10.0f, 0.1f},
};

static const SceneDesc *find_scene(const char *name)
{
    for (const SceneDesc &s : g_scenes)
        if (strcmp(s.name, name) == 0)
            return &s;
    fprintf(stderr, "unknown scene '%s'\n", name);
    exit(2);
}

static bool same_vec(Vec3 a, Vec3 b)
{
    return memcmp(&a.m, &b.m, 12) == 0;
}

// Builds the reference scene, then re-initialises its camera for a runtime aspect.
static Scene *make_scene(const SceneDesc *d, int w, int h)
{
    Scene *scene = d->create();

    // self-check: our restated camera constants reproduce the reference camera
    Camera chk;
    chk.init(Vec3(d->from[0], d->from[1], d->from[2]), Vec3(d->at[0], d->at[1], d->at[2]), Vec3(0, 1, 0), 60,
             (float)SCREEN_W / (float)SCREEN_H, d->aperture, d->focus);
    if (!same_vec(chk._lowerLeftCorner, scene->camera._lowerLeftCorner) || !same_vec(chk._horizontal, scene->camera._horizontal) ||
        !same_vec(chk._vertical, scene->camera._vertical) || !same_vec(chk._origin, scene->camera._origin) ||
        !same_vec(chk._u, scene->camera._u) || !same_vec(chk._v, scene->camera._v) || chk._lensRadius != scene->camera._lensRadius)
    {
#ifdef __FAST_MATH__
        // the -ffast-math timing builds may fold/reassociate the two evaluations differently;
        // they only serve `bench`, which re-initialises the camera below anyway
        static bool warned = false;
        if (!warned)
            fprintf(stderr, "note: camera self-check differs under -ffast-math (scene %s)\n", d->name);
        warned = true;
#else
        fprintf(stderr, "camera self-check failed for scene %s\n", d->name);
        exit(3);
#endif
    }

    scene->camera.init(Vec3(d->from[0], d->from[1], d->from[2]), Vec3(d->at[0], d->at[1], d->at[2]), Vec3(0, 1, 0), 60,
                       (float)w / (float)h, d->aperture, d->focus);
    return scene;
}

// ------------------------------------------------------------------------------------
// tiny tagged binary writer: [tag 8 bytes][dtype 1 byte: f,u,b,d,q][count u64][payload]

struct Writer
{
    FILE *f;
    explicit Writer(const char *path)
    {
        f = fopen(path, "wb");
        if (!f)
        {
            fprintf(stderr, "cannot open %s\n", path);
            exit(4);
        }
        fwrite("R1GOLD01", 1, 8, f);
    }
    ~Writer() { fclose(f); }
    void put(const char *tag, char dtype, const void *data, uint64_t count, size_t elem)
    {
        char t[8] = {0};
        strncpy(t, tag, 8);
        fwrite(t, 1, 8, f);
        fwrite(&dtype, 1, 1, f);
        fwrite(&count, 8, 1, f);
        fwrite(data, elem, count, f);
    }
    void f32(const char *tag, const float *p, uint64_t n) { put(tag, 'f', p, n, 4); }
    void u32(const char *tag, const uint32_t *p, uint64_t n) { put(tag, 'u', p, n, 4); }
    void u8(const char *tag, const uint8_t *p, uint64_t n) { put(tag, 'b', p, n, 1); }
    void u64(const char *tag, const uint64_t *p, uint64_t n) { put(tag, 'q', p, n, 8); }
};

static void vec3_out(float *dst, Vec3 v)
{
    dst[0] = v.getX();
    dst[1] = v.getY();
    dst[2] = v.getZ();
}

// ------------------------------------------------------------------------------------
// scene dump

static int cmd_scene(int argc, const char **argv)
{
    // scene <scene> <w> <h> <out>
    if (argc < 4)
        return 1;
    const SceneDesc *d = find_scene(argv[0]);
    int w = atoi(argv[1]), h = atoi(argv[2]);
    Scene *scene = make_scene(d, w, h);
    const SphereSOA::InstanceData *s = scene->hitables->_soa_spheres.getData();
    uint32_t n = s->_count;

    // material classification without RTTI: compare vtable pointers
    Lambertian l(Vec3(0, 0, 0));
    Metal m(Vec3(0, 0, 0), 0);
    Dielectric g(1);
    void *vl = *(void **)&l, *vm = *(void **)&m, *vg = *(void **)&g;

    std::vector<uint8_t> type(n);
    std::vector<float> ar(n), ag(n), ab(n), par(n);
    for (uint32_t i = 0; i < n; ++i)
    {
        Material *mat = s->material[i];
        ar[i] = ag[i] = ab[i] = par[i] = 0;
        if (!mat)
        {
            type[i] = 255;
            continue;
        }
        void *v = *(void **)mat;
        if (v == vl)
        {
            type[i] = 0;
            Vec3 a = ((Lambertian *)mat)->albedo;
            ar[i] = a.getX(), ag[i] = a.getY(), ab[i] = a.getZ();
        }
        else if (v == vm)
        {
            type[i] = 1;
            Vec3 a = ((Metal *)mat)->albedo;
            ar[i] = a.getX(), ag[i] = a.getY(), ab[i] = a.getZ();
            par[i] = ((Metal *)mat)->fuzz;
        }
        else if (v == vg)
        {
            type[i] = 2;
            ar[i] = ag[i] = ab[i] = 1;
            par[i] = ((Dielectric *)mat)->_refIdx;
        }
        else
        {
            fprintf(stderr, "unknown material vtable\n");
            return 5;
        }
    }

    float cam[22];
    vec3_out(cam + 0, scene->camera._origin);
    vec3_out(cam + 3, scene->camera._lowerLeftCorner);
    vec3_out(cam + 6, scene->camera._horizontal);
    vec3_out(cam + 9, scene->camera._vertical);
    vec3_out(cam + 12, scene->camera._u);
    vec3_out(cam + 15, scene->camera._v);
    vec3_out(cam + 18, scene->camera._w);
    cam[21] = scene->camera._lensRadius;

    Writer wr(argv[3]);
    uint32_t dims[3] = {(uint32_t)w, (uint32_t)h, n};
    wr.u32("dims", dims, 3);
    wr.f32("cx", s->center_x, n);
    wr.f32("cy", s->center_y, n);
    wr.f32("cz", s->center_z, n);
    wr.f32("rsq", s->radius_sq, n);
    wr.f32("invr", s->inv_radius, n);
    wr.u8("mtype", type.data(), n);
    wr.f32("alb_r", ar.data(), n);
    wr.f32("alb_g", ag.data(), n);
    wr.f32("alb_b", ab.data(), n);
    wr.f32("mparam", par.data(), n);
    wr.f32("camera", cam, 22);
    delete scene;
    return 0;
}

// ------------------------------------------------------------------------------------
// per-sample evaluation with the seeding contract; mirrors rayweek1.cpp:757-763

struct SampleOut
{
    float r, g, b;
    uint32_t rays;
};

static inline SampleOut eval_sample(Scene *scene, int w, int h, uint32_t seed, int x, int y, int s)
{
    ThreadData td;
    memset(&td, 0, sizeof(td));
    td.scene = scene;
    td.image_w = w;
    td.image_h = h;
    r1_sample_seed sd = r1_seed_sample(seed, (uint32_t)(y * w + x), (uint32_t)s);
    td.state = sd.scalar;
    td.state4 = _mm_set_epi32((int)R1_SEED_LANE3, (int)sd.lane2, (int)sd.lane1, (int)sd.lane0);
    td.out_num_rays = 0;

    Vec3 inv_image_size(1.0f / w, 1.0f / h, 0);
    Vec3 xy((float)x, (float)y, 0);
    Vec3 uv = (Vec3(myrand01_x4(td.state4)) + xy) * inv_image_size;
    Ray r = scene->camera.getRay(uv.getX(), uv.getY(), td.state);
    Vec3 col = color(r, scene->hitables, 0, &td);

    SampleOut o;
    o.r = col.getX();
    o.g = col.getY();
    o.b = col.getZ();
    o.rays = (uint32_t)td.out_num_rays;
    return o;
}

static int cmd_samples(int argc, const char **argv)
{
    // samples <scene> <w> <h> <spp> <seed> <n> <sel_seed> <out>
    if (argc < 8)
        return 1;
    const SceneDesc *d = find_scene(argv[0]);
    int w = atoi(argv[1]), h = atoi(argv[2]), spp = atoi(argv[3]);
    uint32_t seed = (uint32_t)strtoul(argv[4], 0, 0);
    uint64_t n = strtoull(argv[5], 0, 0);
    uint32_t sel = (uint32_t)strtoul(argv[6], 0, 0);
    Scene *scene = make_scene(d, w, h);

    std::vector<uint32_t> xs(n), ys(n), ss(n), rays(n);
    std::vector<float> rgb(3 * n);
    uint32_t st = r1_nonzero(r1_mix32(sel));
    for (uint64_t i = 0; i < n; ++i)
    {
        st = r1_mix32(st + 0x9E3779B9u);
        xs[i] = st % (uint32_t)w;
        st = r1_mix32(st + 0x9E3779B9u);
        ys[i] = st % (uint32_t)h;
        st = r1_mix32(st + 0x9E3779B9u);
        ss[i] = st % (uint32_t)spp;
        SampleOut o = eval_sample(scene, w, h, seed, (int)xs[i], (int)ys[i], (int)ss[i]);
        rgb[3 * i] = o.r, rgb[3 * i + 1] = o.g, rgb[3 * i + 2] = o.b;
        rays[i] = o.rays;
    }
    Writer wr(argv[7]);
    uint32_t hdr[5] = {(uint32_t)w, (uint32_t)h, (uint32_t)spp, seed, sel};
    wr.u32("hdr", hdr, 5);
    wr.u32("x", xs.data(), n);
    wr.u32("y", ys.data(), n);
    wr.u32("s", ss.data(), n);
    wr.f32("rgb", rgb.data(), 3 * n);
    wr.u32("rays", rays.data(), n);
    delete scene;
    return 0;
}

// full frame under the seeding contract: per-sample results summed in sample order and
// resolved exactly as rayweek1.cpp:765-775
static int cmd_frame(int argc, const char **argv)
{
    // frame <scene> <w> <h> <spp> <seed> <threads> <out> [dump_samples]
    if (argc < 7)
        return 1;
    const SceneDesc *d = find_scene(argv[0]);
    int w = atoi(argv[1]), h = atoi(argv[2]), spp = atoi(argv[3]);
    uint32_t seed = (uint32_t)strtoul(argv[4], 0, 0);
    int nthreads = atoi(argv[5]);
    bool dump = argc > 7 && atoi(argv[7]) != 0;
    Scene *scene = make_scene(d, w, h);

    std::vector<uint8_t> img((size_t)w * h * 3);
    std::vector<uint64_t> row_rays(h, 0);
    std::vector<float> samp;
    if (dump)
        samp.resize((size_t)w * h * spp * 4);
    std::atomic<int> next_row{0};
    auto worker = [&]() {
        int y;
        while ((y = next_row.fetch_add(1)) < h)
        {
            uint64_t rays = 0;
            for (int x = 0; x < w; ++x)
            {
                Vec3 col(0, 0, 0);
                for (int s = 0; s < spp; ++s)
                {
                    SampleOut o = eval_sample(scene, w, h, seed, x, y, s);
                    col += Vec3(o.r, o.g, o.b);
                    rays += o.rays;
                    if (dump)
                    {
                        float *p = &samp[(((size_t)y * w + x) * spp + s) * 4];
                        p[0] = o.r, p[1] = o.g, p[2] = o.b;
                        memcpy(p + 3, &o.rays, 4);
                    }
                }
                col *= (float)(1.0f / spp);
                col = Vec3(sqrtf(col.getX()), sqrtf(col.getY()), sqrtf(col.getZ()));
                uint8_t *px = &img[((size_t)y * w + x) * 3];
                px[0] = (uint8_t)(int)(col.getX() * 255.99f);
                px[1] = (uint8_t)(int)(col.getY() * 255.99f);
                px[2] = (uint8_t)(int)(col.getZ() * 255.99f);
            }
            row_rays[y] = rays;
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < nthreads; ++i)
        th.emplace_back(worker);
    for (auto &t : th)
        t.join();
    uint64_t total = 0;
    for (int y = 0; y < h; ++y)
        total += row_rays[y];

    Writer wr(argv[6]);
    uint32_t hdr[4] = {(uint32_t)w, (uint32_t)h, (uint32_t)spp, seed};
    wr.u32("hdr", hdr, 4);
    wr.u8("image", img.data(), img.size());
    wr.u64("rays", &total, 1);
    wr.u64("rowrays", row_rays.data(), h);
    if (dump)
        wr.f32("samples", samp.data(), samp.size());
    printf("{\"scene\": \"%s\", \"w\": %d, \"h\": %d, \"spp\": %d, \"seed\": %u, \"rays\": %llu}\n", d->name, w, h, spp, seed,
           (unsigned long long)total);
    delete scene;
    return 0;
}

// the reference's deterministic single-thread path (rayweek1.cpp:879-888): its own
// render_tile() over tiles 0..T-1 with the sequential streams 10001 / (1001..1007)
static int cmd_seq(int argc, const char **argv)
{
    // seq <scene> <w> <h> <spp> <out>
    if (argc < 5)
        return 1;
    const SceneDesc *d = find_scene(argv[0]);
    int w = atoi(argv[1]), h = atoi(argv[2]), spp = atoi(argv[3]);
    Scene *scene = make_scene(d, w, h);
    std::vector<Pix> pixels((size_t)w * h);

    ThreadData td;
    memset(&td, 0, sizeof(td));
    td.scene = scene;
    td.image = pixels.data();
    td.image_w = w;
    td.image_h = h;
    td.tile_w_in_pixels = 32 > w ? w : 32;
    td.tile_h_in_pixels = 32 > h ? h : 32;
    td.samples_per_pixel = spp;
    td.out_num_rays = 0;
    td.state = 10001;
    td.state4 = _mm_set_epi32(1001, 1003, 1005, 1007);
    int num_tiles = tiles_required(td.tile_w_in_pixels, w) * tiles_required(td.tile_h_in_pixels, h);
    uint64_t total = 0;
    for (int i = 0; i < num_tiles; ++i)
    {
        render_tile(i, &td);
        total += td.out_num_rays;
        td.out_num_rays = 0;
    }
    Writer wr(argv[4]);
    uint32_t hdr[3] = {(uint32_t)w, (uint32_t)h, (uint32_t)spp};
    wr.u32("hdr", hdr, 3);
    wr.u8("image", (const uint8_t *)pixels.data(), pixels.size() * 3);
    wr.u64("rays", &total, 1);
    printf("{\"scene\": \"%s\", \"w\": %d, \"h\": %d, \"spp\": %d, \"mode\": \"seq\", \"rays\": %llu}\n", d->name, w, h, spp,
           (unsigned long long)total);
    delete scene;
    return 0;
}

// the reference's own multi-threaded path (rayweek1.cpp:851-891) with runtime sizes
static int cmd_bench(int argc, const char **argv)
{
    // bench <scene> <w> <h> <spp> <threads (0 = hardware_concurrency)> <runs> [out_image]
    if (argc < 6)
        return 1;
    const SceneDesc *d = find_scene(argv[0]);
    int w = atoi(argv[1]), h = atoi(argv[2]), spp = atoi(argv[3]);
    int threads = atoi(argv[4]), runs = atoi(argv[5]);
    if (threads <= 0)
        threads = (int)std::thread::hardware_concurrency();
    std::vector<Pix> pixels((size_t)w * h);
    for (int run = 0; run < runs; ++run)
    {
        Scene *scene = make_scene(d, w, h);
        Timer timer;
        ThreadData td;
        memset(&td, 0, sizeof(td));
        td.scene = scene;
        td.image = pixels.data();
        td.image_w = w;
        td.image_h = h;
        td.tile_w_in_pixels = 32 > w ? w : 32;
        td.tile_h_in_pixels = 32 > h ? h : 32;
        td.samples_per_pixel = spp;
        td.out_num_rays = 0;
        int num_tiles = tiles_required(td.tile_w_in_pixels, w) * tiles_required(td.tile_h_in_pixels, h);
        TileRenderScheduler scheduler;
        uint64_t rays = scheduler.run(num_tiles, threads, &td);
        double el = timer.elapsed();
        printf("{\"scene\": \"%s\", \"w\": %d, \"h\": %d, \"spp\": %d, \"threads\": %d, \"run\": %d, \"rays\": %llu, \"seconds\": %.6f, "
               "\"mrays_per_s\": %.4f}\n",
               d->name, w, h, spp, threads, run, (unsigned long long)rays, el, rays / el / 1e6);
        fflush(stdout);
        delete scene;
    }
    if (argc > 6)
    {
        Writer wr(argv[6]);
        uint32_t hdr[3] = {(uint32_t)w, (uint32_t)h, (uint32_t)spp};
        wr.u32("hdr", hdr, 3);
        wr.u8("image", (const uint8_t *)pixels.data(), pixels.size() * 3);
    }
    return 0;
}

// RNG known-answer vectors (mymath.h:17-73)
static int cmd_kat(int argc, const char **argv)
{
    if (argc < 1)
        return 1;
    const int N = 16;
    uint32_t st = 10001;
    uint32_t raw[N];
    float r01[N], r02[N];
    for (int i = 0; i < N; ++i)
        raw[i] = XorShift32(st);
    st = 10001;
    for (int i = 0; i < N; ++i)
        r01[i] = myrand01(st);
    st = 10001;
    for (int i = 0; i < N; ++i)
        r02[i] = myrand02(st);
    float x01[4 * N], x02[4 * N];
    uint32_t xstate[4 * N];
    __m128i s4 = _mm_set_epi32(10001, 10003, 10005, 10007);
    for (int i = 0; i < N; ++i)
    {
        __m128 v = myrand01_x4(s4);
        _mm_storeu_ps(x01 + 4 * i, v);
        _mm_storeu_si128((__m128i *)(xstate + 4 * i), s4);
    }
    s4 = _mm_set_epi32(10001, 10003, 10005, 10007);
    for (int i = 0; i < N; ++i)
    {
        __m128 v = myrand02_x4(s4);
        _mm_storeu_ps(x02 + 4 * i, v);
    }
    Writer wr(argv[0]);
    wr.u32("raw", raw, N);
    wr.f32("r01", r01, N);
    wr.f32("r02", r02, N);
    wr.f32("x01", x01, 4 * N);
    wr.f32("x02", x02, 4 * N);
    wr.u32("xstate", xstate, 4 * N);
    return 0;
}

#ifdef R1_DROPIN
// ------------------------------------------------------------------------------------
// The binding of INTEGRATION.md, for real: the reference's OWN Scene object (create_*_scene of
// rayweek1.cpp, untouched) is handed to librays1.so through the C-ABI exactly as a maintainer's
// benchmark() would do it — the SoA arrays are passed in place (soa_sphere.h:38-53), only the
// polymorphic material column is flattened — and the same frame is rendered by the reference's
// own TileRenderScheduler on the host cores.  Output: one JSON line.  (Needs a GPU at run time;
// built here because only this container has the reference sources.)
#include "../include/rays1.h"

static uint64_t fnv1a(const uint8_t *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i)
        h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

static int cmd_dropin(int argc, const char **argv)
{
    // dropin <scene> <w> <h> <spp> <seed> <variant> [out.bin]
    if (argc < 6)
        return 1;
    const SceneDesc *d = find_scene(argv[0]);
    const int w = atoi(argv[1]), h = atoi(argv[2]), spp = atoi(argv[3]);
    const uint32_t seed = (uint32_t)strtoul(argv[4], 0, 10);
    const int variant = atoi(argv[5]);
    Scene *scene = make_scene(d, w, h);
    const SphereSOA::InstanceData *s = scene->hitables->_soa_spheres.getData();
    const uint32_t n = s->_count;

    Lambertian l(Vec3(0, 0, 0));
    Metal m(Vec3(0, 0, 0), 0);
    Dielectric g(1);
    void *vl = *(void **)&l, *vm = *(void **)&m, *vg = *(void **)&g;
    std::vector<uint8_t> type(n, 255);
    std::vector<float> ar(n, 0), ag(n, 0), ab(n, 0), par(n, 0);
    for (uint32_t i = 0; i < n; ++i)
    {
        Material *mat = s->material[i];
        if (!mat)
            continue;
        void *v = *(void **)mat;
        Vec3 a(1, 1, 1);
        if (v == vl)
            type[i] = R1_MAT_LAMBERTIAN, a = ((Lambertian *)mat)->albedo;
        else if (v == vm)
            type[i] = R1_MAT_METAL, a = ((Metal *)mat)->albedo, par[i] = ((Metal *)mat)->fuzz;
        else if (v == vg)
            type[i] = R1_MAT_DIELECTRIC, par[i] = ((Dielectric *)mat)->_refIdx;
        ar[i] = a.getX(), ag[i] = a.getY(), ab[i] = a.getZ();
    }
    r1_scene rs = {n, s->center_x, s->center_y, s->center_z, s->radius_sq, s->inv_radius, type.data(), ar.data(), ag.data(), ab.data(), par.data()};
    r1_camera rc;
    vec3_out(rc.origin, scene->camera._origin);
    vec3_out(rc.lower_left, scene->camera._lowerLeftCorner);
    vec3_out(rc.horizontal, scene->camera._horizontal);
    vec3_out(rc.vertical, scene->camera._vertical);
    vec3_out(rc.u, scene->camera._u);
    vec3_out(rc.v, scene->camera._v);
    vec3_out(rc.w, scene->camera._w);
    rc.lens_radius = scene->camera._lensRadius;

    r1_context *ctx = 0;
    if (r1_create(0, &ctx) != R1_OK || r1_set_scene(ctx, &rs, &rc) != R1_OK)
    {
        fprintf(stderr, "dropin: %s\n", r1_last_error());
        return 6;
    }
    r1_params p = {w, h, spp, MAX_BOUNCES, seed, 32, 32, 0, 1, variant};
    std::vector<Pix> gpu_pixels((size_t)w * h), cpu_pixels((size_t)w * h);
    uint64_t gpu_rays = 0;
    double device_seconds = 0;
    Timer tg;
    if (r1_render(ctx, &p, &gpu_pixels[0].r, &gpu_rays, &device_seconds) != R1_OK)
    {
        fprintf(stderr, "dropin: %s\n", r1_last_error());
        return 6;
    }
    const double gpu_elapsed = tg.elapsed();
    r1_destroy(ctx);

    // the reference's own multi-threaded path on the same Scene object (rayweek1.cpp:851-877)
    Timer tc;
    ThreadData td;
    memset(&td, 0, sizeof(td));
    td.scene = scene;
    td.image = cpu_pixels.data();
    td.image_w = w;
    td.image_h = h;
    td.tile_w_in_pixels = 32 > w ? w : 32;
    td.tile_h_in_pixels = 32 > h ? h : 32;
    td.samples_per_pixel = spp;
    const int num_tiles = tiles_required(td.tile_w_in_pixels, w) * tiles_required(td.tile_h_in_pixels, h);
    TileRenderScheduler scheduler;
    const int threads = (int)std::thread::hardware_concurrency();
    const uint64_t cpu_rays = scheduler.run(num_tiles, threads, &td);
    const double cpu_elapsed = tc.elapsed();

    double gsum = 0, csum = 0, adiff = 0;
    const uint8_t *gp = &gpu_pixels[0].r, *cp = &cpu_pixels[0].r;
    for (size_t i = 0; i < (size_t)w * h * 3; ++i)
        gsum += gp[i], csum += cp[i], adiff += fabs((double)gp[i] - (double)cp[i]);
    const double npx = (double)w * h * 3;
    printf("{\"scene\": \"%s\", \"w\": %d, \"h\": %d, \"spp\": %d, \"seed\": %u, \"variant\": %d, \"spheres\": %u, "
           "\"gpu_rays\": %llu, \"gpu_image_fnv1a\": \"%016llx\", \"gpu_image_mean\": %.6f, \"gpu_seconds\": %.6f, \"gpu_device_seconds\": %.6f, "
           "\"ref_cpu_rays\": %llu, \"ref_cpu_image_mean\": %.6f, \"ref_cpu_seconds\": %.6f, \"ref_cpu_threads\": %d, "
           "\"mean_abs_pixel_difference\": %.6f}\n",
           d->name, w, h, spp, seed, variant, n, (unsigned long long)gpu_rays, (unsigned long long)fnv1a(gp, (size_t)w * h * 3), gsum / npx,
           gpu_elapsed, device_seconds, (unsigned long long)cpu_rays, csum / npx, cpu_elapsed, threads, adiff / npx);
    if (argc > 6)
    {
        Writer wr(argv[6]);
        uint32_t hdr[3] = {(uint32_t)w, (uint32_t)h, (uint32_t)spp};
        wr.u32("hdr", hdr, 3);
        wr.u8("image", gp, (size_t)w * h * 3);
    }
    delete scene;
    return 0;
}
#endif

int main(int argc, const char **argv)
{
    if (argc < 2)
    {
        fprintf(stderr, "usage: %s scene|samples|frame|seq|bench|kat ...\n", argv[0]);
        return 1;
    }
    const char *cmd = argv[1];
    int rc = 1;
    if (!strcmp(cmd, "scene"))
        rc = cmd_scene(argc - 2, argv + 2);
    else if (!strcmp(cmd, "samples"))
        rc = cmd_samples(argc - 2, argv + 2);
    else if (!strcmp(cmd, "frame"))
        rc = cmd_frame(argc - 2, argv + 2);
    else if (!strcmp(cmd, "seq"))
        rc = cmd_seq(argc - 2, argv + 2);
    else if (!strcmp(cmd, "bench"))
        rc = cmd_bench(argc - 2, argv + 2);
    else if (!strcmp(cmd, "kat"))
        rc = cmd_kat(argc - 2, argv + 2);
#ifdef R1_DROPIN
    else if (!strcmp(cmd, "dropin"))
        rc = cmd_dropin(argc - 2, argv + 2);
#endif
    if (rc == 1)
        fprintf(stderr, "bad arguments for '%s'\n", cmd);
    return rc;
}
