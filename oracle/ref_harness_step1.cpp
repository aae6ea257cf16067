// oracle/ref_harness_step1.cpp — TEST INFRASTRUCTURE, not product code.
//
// BASELINE config 1 ("small scene, 200x100, 1 spp, step1 CPU baseline"): drives the
// reference's OWN step1 functions (create_small_scene, Camera::getRay, color, myrand)
// compiled from /root/reference/src/step1/rayweek1.cpp (-DREF_STEP1_TU, nothing copied).
// step1's benchmark() (step1/rayweek1.cpp:689-765) bakes SCREEN_W/H/spp in as macros, so
// the pixel loop below restates ITS loop (:701-727) with runtime sizes around the
// reference's own calls, and re-runs the reference's Camera::init with the runtime aspect.
// Self-check: at the macro size the loop must reproduce benchmark()'s ray count.

#include <stdio.h>
#include <stdlib.h>
#include <assert.h>
#include <float.h>
#include <ctime>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <vector>

#define main ref_main
#include REF_STEP1_TU
#undef main

struct SceneDesc
{
    const char *name;
    Scene *(*create)();
    float from[3];
    float focus, aperture;
};
static const SceneDesc g_scenes[3] = {
    {"small", create_small_scene, {2, 1, 2}, 5.0f, 0.1f},
    {"medium", create_medium_scene, {0, 2, 3}, 3.0f, 0.1f * 0.2f},
    {"large", create_large_scene, {3, 8, 15}, 10.0f, 0.1f},
};

static uint32_t render(Scene *scene, int nx, int ny, int spp, Pix *pixels)
{
    xorshift_state = 1236787; // :691
    uint32_t num_rays = 0;
    for (int y = ny - 1; y >= 0; --y)
        for (int x = 0; x < nx; ++x)
        {
            Vec3 col(0, 0, 0);
            for (int s = 0; s < spp; ++s)
            {
                float u = (float)((x + myrand()) / nx);
                float v = (float)((y + myrand()) / ny);
                Ray r = scene->camera.getRay(u, v);
                col += color(r, scene->hitables, 0, &num_rays);
            }
            col /= (float)spp;
            col = Vec3(sqrtf(col.x), sqrtf(col.y), sqrtf(col.z));
            pixels[y * nx + x].r = (uint8_t)(int)(col.x * 255.99f);
            pixels[y * nx + x].g = (uint8_t)(int)(col.y * 255.99f);
            pixels[y * nx + x].b = (uint8_t)(int)(col.z * 255.99f);
        }
    return num_rays;
}

int main(int argc, const char **argv)
{
    // frame1 <scene> <w> <h> <spp> <out.bin>   |   selfcheck
    if (argc >= 2 && !strcmp(argv[1], "selfcheck"))
    {
        // the reference's own benchmark() at its macro size vs the loop above
        std::vector<Pix> a((size_t)SCREEN_W * SCREEN_H), b((size_t)SCREEN_W * SCREEN_H);
        RESULT r = benchmark(create_small_scene(), a.data(), false, "small");
        Scene *sc = create_small_scene();
        uint32_t rays = render(sc, SCREEN_W, SCREEN_H, NUM_SAMPLES_PER_PIXEL, b.data());
        delete sc;
        bool same = rays == r.num_rays && memcmp(a.data(), b.data(), a.size() * 3) == 0;
        printf("selfcheck %s: benchmark() rays %llu, harness loop rays %u\n", same ? "OK" : "FAILED", (unsigned long long)r.num_rays, rays);
        return same ? 0 : 3;
    }
    if (argc < 7 || strcmp(argv[1], "frame1"))
    {
        fprintf(stderr, "usage: %s frame1 <scene> <w> <h> <spp> <out.bin> | selfcheck\n", argv[0]);
        return 1;
    }
    const SceneDesc *d = 0;
    for (const SceneDesc &s : g_scenes)
        if (!strcmp(s.name, argv[2]))
            d = &s;
    if (!d)
        return 2;
    int w = atoi(argv[3]), h = atoi(argv[4]), spp = atoi(argv[5]);
    Scene *scene = d->create();
    scene->camera.init(Vec3(d->from[0], d->from[1], d->from[2]), Vec3(0, 0, 0), Vec3(0, 1, 0), 60, (float)w / (float)h, d->aperture, d->focus);
    std::vector<Pix> pixels((size_t)w * h);
    uint32_t rays = render(scene, w, h, spp, pixels.data());
    delete scene;

    FILE *f = fopen(argv[6], "wb");
    if (!f)
        return 4;
    fwrite("R1GOLD01", 1, 8, f);
    auto put = [&](const char *tag, char dt, const void *p, uint64_t n, size_t el) {
        char t[8] = {0};
        strncpy(t, tag, 8);
        fwrite(t, 1, 8, f);
        fwrite(&dt, 1, 1, f);
        fwrite(&n, 8, 1, f);
        fwrite(p, el, n, f);
    };
    uint32_t hdr[3] = {(uint32_t)w, (uint32_t)h, (uint32_t)spp};
    put("hdr", 'u', hdr, 3, 4);
    put("image", 'b', pixels.data(), pixels.size() * 3, 1);
    uint64_t r64 = rays;
    put("rays", 'q', &r64, 1, 8);
    fclose(f);
    printf("{\"scene\": \"%s\", \"w\": %d, \"h\": %d, \"spp\": %d, \"mode\": \"step1\", \"rays\": %u}\n", d->name, w, h, spp, rays);
    return 0;
}
