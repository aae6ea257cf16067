// rayweek1_hip.cpp — the drop-in host program: the reference's entry points for the hot path
//     Scene *create_small_scene() / create_medium_scene() / create_large_scene()
//                                       (/root/reference/src/step13/rayweek1.cpp:552 / :582 / :654)
//     RESULT benchmark(Scene *scene, Pix *pixels, bool write_tga, const char *scene_name)
//                                       (rayweek1.cpp:845-927)
//     main: -w, -n N                    (rayweek1.cpp:930-988)
// kept with the same signatures, ownership (benchmark() deletes the scene it is given,
// rayweek1.cpp:905), stdout block (:895-902), out_<scene>.txt (common.h:47-77) and TGA
// (common.h:86-122), but rendering through librays1.so (include/rays1.h) on a MI355X.
//
// What the reference fixes at compile time (common.h:19-28) is a run-time option here:
//     --width W --height H --spp S --seed N --device D --devices N --variant V
//       (V = R1_VARIANT_*: 0 default, 1 reference-form sweep, 2 exhaustive sweep, 4 box tree, 6 wavefront)
// Defaults are the reference's multi-threaded defaults: 1280x720, 250 spp.
// --devices N splits the frame over N HIP devices inside this one process, tile t -> device
// t % N (the in-process twin of the reference's thread pool, rayweek1.cpp:785-842):
//     --gather rccl (default when the N devices are distinct GPUs): r1_multi_render — every device
//         renders its tiles, ONE ncclAllGather of the tile blocks + ray counts over xGMI, device 0
//         assembles the image and copies it to the host once (the join of rayweek1.cpp:804-813)
//     --gather host: one host thread + r1_context per device, each device copies its own tiles
//         into the caller's pixel buffer (also works oversubscribed: N contexts on fewer GPUs)
// bench.py's one-process-per-GPU form (torch.distributed) gathers the same records.
// --backend hip (default) | cpu-step1 | cpu-step12: the reference's two single-thread CPU stages
// (r1_cpu_backends.cpp) behind the same benchmark(), for the README-style table
// (README.md:38-84).  Named backends of this program only — never a fallback: with --backend hip and no
// usable device the program exits with an error.

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rays1.h"

// ---- common.h types ----------------------------------------------------------------------------

struct RESULT // common.h:36-45
{
    double elapsed_seconds;
    uint64_t num_rays;
    double get_mrays_per_sec() const { return elapsed_seconds ? (num_rays / elapsed_seconds / 1000000.0) : 0; }
};

struct Pix // common.h:80-83
{
    uint8_t r, g, b;
};

// ---- run-time configuration (the reference's SCREEN_W / SCREEN_H / NUM_SAMPLES_PER_PIXEL) --------

static int g_screen_w = 1280;
static int g_screen_h = 720;
static int g_spp = 10 * 25;
static int g_max_bounces = 50;
static uint32_t g_seed = 10001;
static int g_device = 0;
static int g_variant = R1_VARIANT_DEFAULT;
static int g_devices = 1;
static int g_backend = 0; // 0 hip, 1 cpu-step1, 12 cpu-step12
static int g_pipeline = 0; // --pipeline FRAMES: after the benchmark() runs, FRAMES frames per scene with several in flight (r1_render_async)
static int g_inflight = 20;
int r1cpu_step12_render(const r1_scene *scene, const r1_camera *cam, int width, int height, int spp, int max_bounces, uint8_t *rgb,
                        uint64_t *num_rays); // r1_cpu_backends.cpp
int r1cpu_step1_render(int scene_kind, const r1_scene *scene, const r1_camera *cam, int width, int height, int spp, uint8_t *rgb,
                       uint64_t *num_rays);
static std::vector<r1_context *> g_ctx; // one per device in use (--gather host)
static r1_multi *g_multi = nullptr;     // --gather rccl
static int g_gather = -1;               // -1 auto, 0 host, 1 rccl
static std::vector<double> g_device_seconds; // per benchmark() call, for the JSON record
static r1_launch_info g_last_info;

// ---- Scene ---------------------------------------------------------------------------------------

class Scene // rayweek1.cpp:539-549: owns its spheres/materials, deleted by benchmark()
{
  public:
    r1_host_scene *host = nullptr;
    int kind = 0; // R1_SCENE_*
    const r1_scene *hitables = nullptr;
    const r1_camera *camera = nullptr;
    ~Scene() { r1_host_scene_destroy(host); }
};

static Scene *make_scene(int kind)
{
    Scene *s = new Scene;
    if (r1_host_scene_create(kind, g_screen_w, g_screen_h, 0, 0, &s->host) != R1_OK)
    {
        fprintf(stderr, "scene build failed: %s\n", r1_last_error());
        exit(1);
    }
    s->kind = kind;
    s->hitables = r1_host_scene_spheres(s->host);
    s->camera = r1_host_scene_camera(s->host);
    return s;
}

Scene *create_small_scene() { return make_scene(R1_SCENE_SMALL); }
Scene *create_medium_scene() { return make_scene(R1_SCENE_MEDIUM); }
Scene *create_large_scene() { return make_scene(R1_SCENE_LARGE); }

// ---- benchmark -------------------------------------------------------------------------------------

RESULT benchmark(Scene *scene, Pix *pixels, bool write_tga, const char *scene_name)
{
    RESULT result = {0, 0};
    auto t0 = std::chrono::high_resolution_clock::now(); // Timer, rayweek1.cpp:848

    r1_params p;
    memset(&p, 0, sizeof(p));
    p.width = g_screen_w, p.height = g_screen_h, p.spp = g_spp, p.max_bounces = g_max_bounces;
    p.seed = g_seed;
    p.tile_w = 32, p.tile_h = 32; // rayweek1.cpp:855-856
    p.shard = 0, p.num_shards = 1;
    p.variant = g_variant;

    if (g_backend != 0)
    {
        // the reference's single-thread CPU stages (r1_cpu_backends.cpp), same timer span and report
        if (g_backend == 12)
            r1cpu_step12_render(scene->hitables, scene->camera, g_screen_w, g_screen_h, g_spp, g_max_bounces, &pixels[0].r, &result.num_rays);
        else
            r1cpu_step1_render(scene->kind, scene->hitables, scene->camera, g_screen_w, g_screen_h, g_spp, &pixels[0].r, &result.num_rays);
        result.elapsed_seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        g_device_seconds.push_back(0.0);
        memset(&g_last_info, 0, sizeof(g_last_info));
        g_last_info.spheres_padded = (int32_t)scene->hitables->count;
        printf("%s\n", scene_name);
        printf("elapsed time:   %.3fs\n", result.elapsed_seconds);
        printf("total samples:  %llu\n", (unsigned long long)((uint64_t)g_screen_w * g_screen_h * g_spp));
        printf("total rays:     %llu\n", (unsigned long long)result.num_rays);
        printf("mrays/s:        %0.2f\n", result.get_mrays_per_sec());
        printf("backend:        %s (1 host thread, no GPU)\n", g_backend == 12 ? "cpu-step12" : "cpu-step1");
        printf("\n");
        delete scene; // rayweek1.cpp:905
        if (write_tga)
        {
            char filename[128];
            snprintf(filename, sizeof(filename), "out_%s.tga", scene_name);
            r1_tga_write_rgb24(filename, g_screen_w, g_screen_h, &pixels[0].r);
        }
        return result;
    }
    if (g_multi)
    {
        // tile split over the devices + one RCCL all-gather, behind one call
        double device_seconds = 0;
        int rc = r1_multi_set_scene(g_multi, scene->hitables, scene->camera);
        if (rc == R1_OK)
            rc = r1_multi_render(g_multi, &p, &pixels[0].r, &result.num_rays, &device_seconds);
        if (rc != R1_OK)
        {
            fprintf(stderr, "%s: render failed (%d): %s\n", scene_name, rc, r1_last_error());
            result.num_rays = 0;
            g_device_seconds.push_back(0.0);
            delete scene;
            return result;
        }
        result.elapsed_seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count(); // :891
        int32_t nd = 0, ver = 0;
        r1_launch_info li;
        memset(&li, 0, sizeof(li));
        r1_multi_info(g_multi, &nd, &ver, &li);
        g_last_info = li;
        g_device_seconds.push_back(device_seconds);
        printf("%s\n", scene_name);
        printf("elapsed time:   %.3fs\n", result.elapsed_seconds);
        printf("total samples:  %llu\n", (unsigned long long)((uint64_t)g_screen_w * g_screen_h * g_spp));
        printf("total rays:     %llu\n", (unsigned long long)result.num_rays);
        printf("mrays/s:        %0.2f\n", result.get_mrays_per_sec());
        printf("devices:        %d (first hip:%d, %d CUs, %d workgroups x %d threads), gather: RCCL %d all-gather\n", nd, g_device, li.compute_units,
               li.blocks, li.threads_per_block, ver);
        printf("device time:    %.3fms (%0.2f mrays/s)\n", device_seconds * 1e3, device_seconds ? result.num_rays / device_seconds / 1e6 : 0.0);
        printf("\n");
        delete scene; // rayweek1.cpp:905
        if (write_tga)
        {
            char filename[128];
            snprintf(filename, sizeof(filename), "out_%s.tga", scene_name);
            r1_tga_write_rgb24(filename, g_screen_w, g_screen_h, &pixels[0].r);
        }
        return result;
    }

    // one host thread per device; with one device this is a plain call
    const int nd = (int)g_ctx.size();
    std::vector<int> rcs(nd, R1_OK);
    std::vector<uint64_t> rays(nd, 0);
    std::vector<double> dev_s(nd, 0.0);
    std::vector<std::string> errs(nd);
    auto worker = [&](int i) {
        r1_params q = p;
        q.shard = i, q.num_shards = nd;
        int rc = r1_set_scene(g_ctx[i], scene->hitables, scene->camera);
        if (rc == R1_OK)
            rc = r1_render(g_ctx[i], &q, &pixels[0].r, &rays[i], &dev_s[i]);
        rcs[i] = rc;
        if (rc != R1_OK)
            errs[i] = r1_last_error(); // thread-local in the library
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nd; ++i)
        th.emplace_back(worker, i);
    worker(0);
    for (auto &t : th)
        t.join();
    double device_seconds = 0;
    int rc = R1_OK;
    for (int i = 0; i < nd; ++i)
    {
        result.num_rays += rays[i]; // rayweek1.cpp:809-813
        device_seconds = dev_s[i] > device_seconds ? dev_s[i] : device_seconds;
        if (rcs[i] != R1_OK && rc == R1_OK)
        {
            rc = rcs[i];
            fprintf(stderr, "%s: device %d: %s\n", scene_name, i, errs[i].c_str());
        }
    }
    if (rc != R1_OK)
    {
        // the reference has no error convention: report and return RESULT{0,0}
        fprintf(stderr, "%s: render failed (%d)\n", scene_name, rc);
        result.num_rays = 0;
        g_device_seconds.push_back(0.0); // one entry per benchmark() call, failed ones included (log_json indexes by run)
        delete scene;
        return result;
    }
    result.elapsed_seconds = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count(); // :891

    r1_launch_info li;
    memset(&li, 0, sizeof(li));
    r1_last_launch_info(g_ctx[0], &li);
    g_last_info = li;
    g_device_seconds.push_back(device_seconds);
    uint64_t total_samples = (uint64_t)g_screen_w * g_screen_h * g_spp;

    printf("%s\n", scene_name);
    printf("elapsed time:   %.3fs\n", result.elapsed_seconds);
    printf("total samples:  %llu\n", (unsigned long long)total_samples);
    printf("total rays:     %llu\n", (unsigned long long)result.num_rays);
    printf("mrays/s:        %0.2f\n", result.get_mrays_per_sec());
    printf("devices:        %d (first hip:%d, %d CUs, %d workgroups x %d threads)\n", nd, g_device, li.compute_units, li.blocks,
           li.threads_per_block);
    static const char *const kernel_names[] = {"default", "reference-form sweep", "grouped exhaustive sweep", "grouped exhaustive sweep + counters",
                                               "box tree", "box tree + counters", "wavefront"};
    printf("kernel:         %s (%d hittable spheres, %d inner nodes)\n", li.kernel >= 0 && li.kernel <= 6 ? kernel_names[li.kernel] : "?",
           li.spheres_active, li.bvh_nodes);
    printf("device time:    %.3fms (%0.2f mrays/s)\n", device_seconds * 1e3, device_seconds ? result.num_rays / device_seconds / 1e6 : 0.0);
    printf("\n");

    delete scene; // rayweek1.cpp:905

    if (write_tga)
    {
        char filename[128];
        snprintf(filename, sizeof(filename), "out_%s.tga", scene_name);
        r1_tga_write_rgb24(filename, g_screen_w, g_screen_h, &pixels[0].r); // swaps R/B in `pixels`, as the reference
    }
    return result;
}

static void log_results(const char *version, const char *scene, const RESULT *results, int num_runs) // common.h:47-77
{
    double el[32];
    uint64_t rays[32];
    for (int i = 0; i < num_runs; ++i)
        el[i] = results[i].elapsed_seconds, rays[i] = results[i].num_rays;
    r1_log_results(version, scene, el, rays, num_runs);
}

// next to the reference's out_<scene>.txt (parsed by update_readme.py) a JSON
// record with what the HIP backend can add: device count, per-run device time, and the
// roofline fractions in the survey's accounting (16 B and 16 flop per ray-sphere test).
static void log_json(const char *version, const char *scene, const RESULT *results, int num_runs)
{
    char filename[128];
    snprintf(filename, sizeof(filename), "out_%s.json", scene);
    if (g_device_seconds.size() < (size_t)num_runs)
        return;
    FILE *f = fopen(filename, "wt");
    if (!f)
        return;
    double el = 0, dev = 0;
    uint64_t rays = 0;
    const size_t first = g_device_seconds.size() - (size_t)num_runs;
    fprintf(f, "{\"version\": \"%s\", \"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp\": %d, \"devices\": %d,\n \"runs\": [", version, scene,
            g_screen_w, g_screen_h, g_spp, g_devices);
    for (int i = 0; i < num_runs; ++i)
    {
        fprintf(f, "%s{\"elapsed_seconds\": %.6f, \"num_rays\": %llu, \"device_seconds\": %.6f}", i ? ", " : "", results[i].elapsed_seconds,
                (unsigned long long)results[i].num_rays, g_device_seconds[first + i]);
        el += results[i].elapsed_seconds, dev += g_device_seconds[first + i], rays += results[i].num_rays;
    }
    const double per_ray = 16.0 * g_last_info.spheres_padded; // bytes == flop per ray in the survey's model
    const double rays_per_dev_s = dev > 0 ? rays / dev : 0;
    fprintf(f, "],\n \"mrays_per_s\": %.3f, \"device_mrays_per_s\": %.3f, \"spheres_padded\": %d, \"spheres_active\": %d,\n", el > 0 ? rays / el / 1e6 : 0.0,
            rays_per_dev_s / 1e6, g_last_info.spheres_padded, g_last_info.spheres_active);
    // the two device-only fractions are null for the CPU stages (devices == 0): nothing ran on a GPU
    if (g_devices > 0)
        fprintf(f, " \"algorithmic_bytes_per_ray\": %.0f, \"hbm_algorithmic_fraction_of_8TBs\": %.4f, \"fp32_vector_fraction_of_157TFs\": %.4f}\n", per_ray,
                rays_per_dev_s * per_ray / 8.0e12 / (double)g_devices, rays_per_dev_s * per_ray / 157.3e12 / (double)g_devices);
    else
        fprintf(f, " \"algorithmic_bytes_per_ray\": %.0f, \"hbm_algorithmic_fraction_of_8TBs\": null, \"fp32_vector_fraction_of_157TFs\": null}\n", per_ray);
    fclose(f);
}

// ---- frames in flight from the C++ host (no reference counterpart: benchmark() renders ONE frame and waits) --------
// The `-n` runs of main (rayweek1.cpp:969-984) are independent frames.  Rendered one after the other each pays the
// tail of its own longest bounce chains; kept in flight — one context, stream and page-locked pixel buffer per frame,
// r1_render_async — the next frames fill the chip while one drains, and every frame still ends with its pixels and
// ray count on the host (the Timer span of rayweek1.cpp:848 -> :891, pipelined).  Prints one line per scene.
static int pipelined(const char *scene_name, int kind, int frames)
{
    const int k = g_inflight < frames ? g_inflight : frames;
    const bool multi = g_gather == 1 && g_devices >= 1 && g_multi; // --gather rccl: every frame in flight is split over the N devices (r1_multi)
    r1_host_scene *hs = nullptr;
    if (r1_host_scene_create(kind, g_screen_w, g_screen_h, 0, 0, &hs) != R1_OK)
        return 1;
    r1_params p;
    memset(&p, 0, sizeof(p));
    p.width = g_screen_w, p.height = g_screen_h, p.spp = g_spp, p.max_bounces = g_max_bounces, p.seed = g_seed;
    p.tile_w = 32, p.tile_h = 32, p.shard = 0, p.num_shards = 1, p.variant = g_variant;
    const size_t rec = r1_frame_record_bytes(&p); // image, padded to 8 bytes, + uint64 ray count
    std::vector<r1_context *> ctx((size_t)k, nullptr);
    std::vector<r1_multi *> mul((size_t)k, nullptr);
    std::vector<uint8_t *> host((size_t)k, nullptr);
    int rc = R1_OK;
    const int visible = r1_device_count();
    std::vector<int32_t> devs;
    for (int i = 0; i < g_devices; ++i)
        devs.push_back(g_device + i);
    for (int i = 0; i < k && rc == R1_OK; ++i) // every context first, the scenes afterwards: the streams get their own hardware queues
        rc = multi ? r1_multi_create(g_devices, devs.data(), &mul[(size_t)i]) : r1_create(g_device % (visible > 0 ? visible : 1), &ctx[(size_t)i]);
    for (int i = 0; i < k && rc == R1_OK; ++i)
    {
        rc = multi ? r1_multi_set_scene(mul[(size_t)i], r1_host_scene_spheres(hs), r1_host_scene_camera(hs))
                   : r1_set_scene(ctx[(size_t)i], r1_host_scene_spheres(hs), r1_host_scene_camera(hs));
        if (rc == R1_OK)
            rc = r1_host_alloc(rec, (void **)&host[(size_t)i]);
    }
    auto enqueue = [&](size_t s) {
        return multi ? r1_multi_render_async(mul[s], &p, host[s]) : r1_render_async(ctx[s], &p, host[s], (uint64_t *)(host[s] + rec - 8), nullptr);
    };
    auto wait = [&](size_t s) { return multi ? r1_multi_sync(mul[s]) : r1_sync(ctx[s]); };
    uint64_t rays = 0;
    double secs = 0;
    if (rc == R1_OK)
    {
        for (int pass = 0; pass < 2 && rc == R1_OK; ++pass) // pass 0: workspaces and queues (not timed)
        {
            rays = 0;
            const int n = pass ? frames : k;
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int f = 0; f < n && rc == R1_OK; ++f)
            {
                const size_t s = (size_t)(f % k);
                if (f >= k) // the slot's previous frame has to have landed before its buffer is reused
                {
                    rc = wait(s);
                    rays += *(const uint64_t *)(host[s] + rec - 8);
                }
                if (rc == R1_OK)
                    rc = enqueue(s);
            }
            for (int i = 0; i < k && rc == R1_OK; ++i)
            {
                rc = wait((size_t)i);
                if (i < n)
                    rays += *(const uint64_t *)(host[(size_t)i] + rec - 8);
            }
            secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
        }
    }
    if (rc == R1_OK)
        printf("%s pipelined:  %d frames, %d in flight%s, %.3f ms per frame, %llu rays, %0.2f mrays/s (pixels + count on the host)\n", scene_name, frames,
               k, multi ? " (each split over the devices, RCCL all-gather)" : "", secs / frames * 1e3, (unsigned long long)rays, rays / secs / 1e6);
    else
        fprintf(stderr, "pipelined %s: %s\n", scene_name, r1_last_error());
    for (int i = 0; i < k; ++i)
    {
        r1_host_free(host[(size_t)i]);
        r1_destroy(ctx[(size_t)i]);
        r1_multi_destroy(mul[(size_t)i]);
    }
    r1_host_scene_destroy(hs);
    return rc == R1_OK ? 0 : 1;
}

int main(int argc, const char *argv[])
{
    bool write_tga = false;
    int num_runs = 1;
    const static int MAX_NUMS = 32;
    RESULT results[MAX_NUMS];

    for (int i = 1; i < argc; ++i)
    {
        if (strcmp(argv[i], "-w") == 0)
            write_tga = true;
        else if (strcmp(argv[i], "-n") == 0 && i + 1 < argc)
        {
            int n = atoi(argv[++i]);
            if (n >= 1 && n < MAX_NUMS)
                num_runs = n;
            else
                printf("Invalid num_runs parameter: %d\n", n);
        }
        else if (strcmp(argv[i], "--width") == 0 && i + 1 < argc)
            g_screen_w = atoi(argv[++i]);
        else if (strcmp(argv[i], "--height") == 0 && i + 1 < argc)
            g_screen_h = atoi(argv[++i]);
        else if (strcmp(argv[i], "--spp") == 0 && i + 1 < argc)
            g_spp = atoi(argv[++i]);
        else if (strcmp(argv[i], "--seed") == 0 && i + 1 < argc)
            g_seed = (uint32_t)strtoul(argv[++i], 0, 0);
        else if (strcmp(argv[i], "--device") == 0 && i + 1 < argc)
            g_device = atoi(argv[++i]);
        else if (strcmp(argv[i], "--devices") == 0 && i + 1 < argc)
            g_devices = atoi(argv[++i]);
        else if (strcmp(argv[i], "--variant") == 0 && i + 1 < argc)
            g_variant = atoi(argv[++i]);
        else if (strcmp(argv[i], "--pipeline") == 0 && i + 1 < argc)
            g_pipeline = atoi(argv[++i]);
        else if (strcmp(argv[i], "--inflight") == 0 && i + 1 < argc)
            g_inflight = atoi(argv[++i]);
        else if (strcmp(argv[i], "--backend") == 0 && i + 1 < argc)
        {
            const char *b = argv[++i];
            g_backend = strcmp(b, "hip") == 0 ? 0 : (strcmp(b, "cpu-step1") == 0 ? 1 : (strcmp(b, "cpu-step12") == 0 ? 12 : -1));
        }
        else if (strcmp(argv[i], "--gather") == 0 && i + 1 < argc)
        {
            const char *g = argv[++i];
            g_gather = strcmp(g, "rccl") == 0 ? 1 : (strcmp(g, "host") == 0 ? 0 : -2);
        }
    }
    if (g_screen_w <= 0 || g_screen_h <= 0 || g_spp <= 0 || g_devices < 1 || g_devices > 64 || g_gather == -2 || g_backend < 0)
    {
        fprintf(stderr, "bad --width/--height/--spp/--devices/--gather/--backend\n");
        return 1;
    }

    if (g_pipeline > 0 && g_backend == 0)
    {
        // Frames in flight only overlap when their streams sit on different hardware queues; the ROCm default is 4.  One queue
        // per frame in flight + one for the context benchmark() renders through (two streams that share a queue run their frames
        // one after the other); at most 23: beyond that the process runs out of them.  Must be set before the first HIP call.
        char q[16];
        snprintf(q, sizeof(q), "%d", g_inflight < 1 ? 2 : (g_inflight > 22 ? 23 : g_inflight + 1));
        setenv("GPU_MAX_HW_QUEUES", q, 0);
    }
    // HIP context creation stays outside the timed region and is shared by all -n runs.
    // Devices wrap around the visible ones, so --devices 2 also works (oversubscribed) on one GPU.
    const int visible = g_backend == 0 ? r1_device_count() : 0;
    if (g_backend != 0)
        g_devices = 0, g_gather = 0; // the CPU stages touch no device
    if (g_gather == -1)
        g_gather = g_devices > 1 && g_device + g_devices <= visible ? 1 : 0; // RCCL needs distinct devices
    if (g_gather == 1)
    {
        std::vector<int32_t> devs;
        for (int i = 0; i < g_devices; ++i)
            devs.push_back(g_device + i);
        if (visible <= 0 || r1_multi_create(g_devices, devs.data(), &g_multi) != R1_OK)
        {
            fprintf(stderr, "cannot create the RCCL device group: %s\n", r1_last_error());
            return 2;
        }
    }
    for (int i = 0; i < g_devices && !g_multi; ++i)
    {
        r1_context *c = nullptr;
        if (visible <= 0 || r1_create((g_device + i) % visible, &c) != R1_OK)
        {
            fprintf(stderr, "cannot create HIP context: %s\n", r1_last_error());
            return 2;
        }
        g_ctx.push_back(c);
    }

    // The pixel buffer benchmark() fills (rayweek1.cpp:961 allocates it with new[]).  With the HIP backend it is page-locked memory
    // (r1_host_alloc): the trace launch then stores every finished tile straight into it and r1_render copies nothing
    // (include/rays1.h; any other memory works too and costs one copy of the image per frame).
    Pix *pixels = nullptr;
    bool pixels_pinned = false;
    if (g_backend == 0)
    {
        void *mem = nullptr;
        if (r1_host_alloc((size_t)g_screen_w * g_screen_h * sizeof(Pix), &mem) == R1_OK)
            pixels = (Pix *)mem, pixels_pinned = true;
    }
    if (!pixels)
        pixels = new Pix[(size_t)g_screen_w * g_screen_h];
    memset(pixels, 0, (size_t)g_screen_w * g_screen_h * sizeof(pixels[0]));

    const char *version = g_backend == 0 ? "hip" : (g_backend == 12 ? "cpu-step12" : "cpu-step1");

    for (int i = 0; i < num_runs; ++i)
        results[i] = benchmark(create_small_scene(), pixels, write_tga, "small");
    log_results(version, "small", results, num_runs);
    log_json(version, "small", results, num_runs);

    for (int i = 0; i < num_runs; ++i)
        results[i] = benchmark(create_medium_scene(), pixels, write_tga, "medium");
    log_results(version, "medium", results, num_runs);
    log_json(version, "medium", results, num_runs);

    for (int i = 0; i < num_runs; ++i)
        results[i] = benchmark(create_large_scene(), pixels, write_tga, "large");
    log_results(version, "large", results, num_runs);
    log_json(version, "large", results, num_runs);

    int rc_pipe = 0;
    if (g_pipeline > 0 && g_backend == 0 && g_inflight >= 1 && g_inflight <= 64)
    {
        rc_pipe |= pipelined("small", R1_SCENE_SMALL, g_pipeline);
        rc_pipe |= pipelined("medium", R1_SCENE_MEDIUM, g_pipeline);
        rc_pipe |= pipelined("large", R1_SCENE_LARGE, g_pipeline);
    }

    if (pixels_pinned)
        r1_host_free(pixels);
    else
        delete[] pixels;
    for (r1_context *c : g_ctx)
        r1_destroy(c);
    r1_multi_destroy(g_multi);
    return rc_pipe ? 3 : 0;
}
