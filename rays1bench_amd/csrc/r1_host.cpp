// r1_host.cpp — host-side half of librays1.so that needs no GPU: the scene builders,
// the tile/shard arithmetic and the reference's output formats, behind the C-ABI of
// include/rays1.h.
//
// Scene content follows the reference (citations: /root/reference/src/step13/):
//   create_small_scene  rayweek1.cpp:552-579      create_medium_scene rayweek1.cpp:582-651
//   create_large_scene  rayweek1.cpp:654-719      Camera::init        rayweek1.cpp:366-379
//   SphereSOA::add      soa_sphere.cpp:70-85      placeholders        rayweek1.cpp:574-576
// but the container is this build's own: one flat struct-of-arrays plus a flat material
// table (type, albedo, parameter) instead of `Material*`, because the device cannot follow
// host vtables.  Arithmetic that feeds the fixtures (tests/golden/scene_*.bin) is written
// operation by operation and this file is compiled with -ffp-contract=off.

#include "../../include/rays1.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

extern "C" void r1_set_error(const char *fmt, ...); // r1_capi.cpp

namespace
{

struct V3
{
    float x, y, z;
};

inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 scale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// mymath.h:205-207 — lanes are summed as (x + y) + z
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// mymath.h:211
inline V3 unit(V3 v) { return scale(v, 1.0f / sqrtf(dot(v, v))); }
// mymath.h:190-197 after the lane shuffles
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

inline void store(float *dst, V3 v)
{
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

} // namespace

struct r1_host_scene
{
    std::vector<float> cx, cy, cz, rsq, invr, ar, ag, ab, param;
    std::vector<uint8_t> mtype;
    r1_scene view;
    r1_camera cam;

    // SphereSOA::add, soa_sphere.cpp:70-85
    void add(V3 c, float radius, uint8_t type, V3 albedo, float p)
    {
        cx.push_back(c.x);
        cy.push_back(c.y);
        cz.push_back(c.z);
        rsq.push_back(radius * radius);
        invr.push_back(radius > 0 ? (1.0f / radius) : 0);
        mtype.push_back(type);
        ar.push_back(albedo.x);
        ag.push_back(albedo.y);
        ab.push_back(albedo.z);
        param.push_back(p);
    }
    void lambertian(V3 c, float r, V3 a) { add(c, r, R1_MAT_LAMBERTIAN, a, 0); }
    // Metal::Metal clamps fuzz to <= 1, rayweek1.cpp:422-425
    void metal(V3 c, float r, V3 a, float fuzz) { add(c, r, R1_MAT_METAL, a, fuzz < 1 ? fuzz : 1); }
    // Dielectric attenuation is (1,1,1), rayweek1.cpp:472
    void dielectric(V3 c, float r, float ref_idx) { add(c, r, R1_MAT_DIELECTRIC, {1, 1, 1}, ref_idx); }

    // "make sure num spheres is multiple of SIMD width", rayweek1.cpp:574-576 (SIMD_WIDTH 8)
    void pad_to_simd_width()
    {
        while (cx.size() % 8 != 0)
            add({999999999.f, 999999999.f, 999999999.f}, 0, R1_MAT_NONE, {0, 0, 0}, 0);
    }

    // Camera::init, rayweek1.cpp:366-379
    void camera_init(V3 lookfrom, V3 lookat, V3 vup, float vfov, float aspect, float aperture, float focus_dist)
    {
        cam.lens_radius = aperture / 2;
        // The reference calls tanf(theta / 2).  Its value for vfov = 60 depends on who evaluates
        // it: g++ folds the call at compile time (MPFR, correctly rounded: 0x1.279a74p-1, which
        // is what the fixture binary and any -O2/-O3 build of the reference contain) while
        // glibc's run-time tanf returns 0x1.279a76p-1.  Evaluating in double and rounding once
        // gives the correctly rounded value on every toolchain.
        float theta = vfov * (float)M_PI / 180;
        float half_height = (float)tan((double)(theta / 2));
        float half_width = aspect * half_height;
        V3 w = unit(sub(lookfrom, lookat));
        V3 u = unit(cross(vup, w));
        V3 v = cross(w, u);
        V3 ll = sub(sub(sub(lookfrom, scale(u, half_width * focus_dist)), scale(v, half_height * focus_dist)), scale(w, focus_dist));
        store(cam.origin, lookfrom);
        store(cam.lower_left, ll);
        store(cam.horizontal, scale(u, 2 * half_width * focus_dist));
        store(cam.vertical, scale(v, 2 * half_height * focus_dist));
        store(cam.u, u);
        store(cam.v, v);
        store(cam.w, w);
    }

    void finish()
    {
        view.count = (uint32_t)cx.size();
        view.center_x = cx.data();
        view.center_y = cy.data();
        view.center_z = cz.data();
        view.radius_sq = rsq.data();
        view.inv_radius = invr.data();
        view.mat_type = mtype.data();
        view.albedo_r = ar.data();
        view.albedo_g = ag.data();
        view.albedo_b = ab.data();
        view.mat_param = param.data();
    }
};

namespace
{

void build_small(r1_host_scene &s, float aspect)
{
    s.camera_init({2, 1, 2}, {0, 0, 0}, {0, 1, 0}, 60, aspect, 0.1f, 5.0f);
    s.lambertian({0, 0, -1}, 0.5f, {0.1f, 0.2f, 0.5f});
    s.lambertian({0, -100.5f, -1}, 100.0f, {0.8f, 0.8f, 0});
    s.metal({1, 0, -1}, 0.5f, {0.8f, 0.6f, 0.2f}, 0.3f);
    s.dielectric({-1, 0, -1}, 0.5f, 1.5f);
    s.dielectric({-1, 0, -1}, -0.45f, 1.5f); // negative radius => inv_radius 0 => never hit (rayweek1.cpp:291)
}

void build_medium(r1_host_scene &s, float aspect)
{
    s.camera_init({0, 2, 3}, {0, 0, 0}, {0, 1, 0}, 60, aspect, 0.1f * 0.2f, 3);
    s.lambertian({0, -100.5, -1}, 100, {0.8f, 0.8f, 0.8f});
    s.lambertian({2, 0, -1}, 0.5f, {0.8f, 0.4f, 0.4f});
    s.lambertian({0, 0, -1}, 0.5f, {0.4f, 0.8f, 0.4f});
    s.metal({-2, 0, -1}, 0.5f, {0.4f, 0.4f, 0.8f}, 0);
    s.metal({2, 0, 1}, 0.5f, {0.4f, 0.8f, 0.4f}, 0);
    s.metal({0, 0, 1}, 0.5f, {0.4f, 0.8f, 0.4f}, 0.2f);
    s.metal({-2, 0, 1}, 0.5f, {0.4f, 0.8f, 0.4f}, 0.6f);
    s.dielectric({0.5f, 1, 0.5f}, 0.5f, 1.5f);
    s.lambertian({-1.5f, 1.5f, 0.f}, 0.3f, {0.8f, 0.6f, 0.2f});
    // four rows of nine spheres at z = -3 .. -6, x = 4 .. -4 (rayweek1.cpp:607-642)
    static const float grey[9] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f, 0.9f};
    static const V3 hue[9] = {{0.8f, 0.1f, 0.1f}, {0.8f, 0.5f, 0.1f}, {0.8f, 0.8f, 0.1f}, {0.4f, 0.8f, 0.1f}, {0.1f, 0.8f, 0.1f},
                              {0.1f, 0.8f, 0.5f}, {0.1f, 0.8f, 0.8f}, {0.1f, 0.1f, 0.8f}, {0.5f, 0.1f, 0.8f}};
    for (int i = 0; i < 9; ++i)
        s.lambertian({(float)(4 - i), 0, -3}, 0.5f, {grey[i], grey[i], grey[i]});
    for (int i = 0; i < 9; ++i)
        s.metal({(float)(4 - i), 0, -4}, 0.5f, {grey[i], grey[i], grey[i]}, 0);
    for (int i = 0; i < 9; ++i)
        s.metal({(float)(4 - i), 0, -5}, 0.5f, hue[i], 0);
    for (int i = 0; i < 8; ++i)
        s.lambertian({(float)(4 - i), 0, -6}, 0.5f, hue[i]);
    s.metal({-4, 0, -6}, 0.5f, hue[8], 0); // the last one of the z = -6 row is metal (rayweek1.cpp:642)
    s.lambertian({1.5f, 1.5f, -2}, 0.3f, {0.1f, 0.2f, 0.5f});
}

// create_large_scene generalised to a gw x gh grid of small spheres.  At gw = 30, gh = 16
// (scale == 1.0f exactly) every value equals the reference's (rayweek1.cpp:670-712).
// Larger grids (BASELINE config 5) shrink spacing, radius and the metal y-offset by 30/gw
// so that footprint and camera stay put; the dielectric index rule wraps at the
// reference's 480 spheres.  Albedos come from glibc rand() after srand(111), as in the
// reference (rayweek1.cpp:673, :682-684): same libc => same colours.
void build_grid(r1_host_scene &s, float aspect, int gw, int gh)
{
    s.camera_init({3, 8, 15}, {0, 0, 0}, {0, 1, 0}, 60, aspect, 0.1f, 10.0f);
    const float k = 30.0f / (float)gw;
    srand(111);
    for (int y = 0; y < gh; ++y)
        for (int x = 0; x < gw; ++x)
        {
            V3 pos = {(x - gw / 2) * 1.1f * k, 0, (y - gh / 2) * 1.1f * k};
            float r = (rand() & 0xff) / 255.0f;
            float g = (rand() & 0xff) / 255.0f;
            float b = (rand() & 0xff) / 255.0f;
            int i = x + y * gw;
            float radius = 0.45f * k;
            if (i % 20 == 0)
                s.dielectric(pos, radius, 1.2f + (i % 480) * 0.05f);
            else if (i % 10 == 0)
            {
                pos = add(pos, {0, 0.1f * k, 0});
                s.metal(pos, radius, {r, g, b}, 0.01f + 0.5f * y / (float)(gh));
            }
            else
                s.lambertian(pos, radius, {r, g, b});
        }
    s.lambertian({0, -1000.5f, 0}, 1000, {0.5f, 0.5f, 0.5f});
    s.metal({5, 3, 0}, 2, {0.5f, 0.5f, 0.8f}, 0.65f);
    s.dielectric({0, 3, 0}, 2, 1.5f);
    s.metal({-5, 3, 0}, 2, {0.8f, 0.2f, 0.2f}, 0.05f);
}

} // namespace

extern "C" int r1_host_scene_create(int kind, int32_t width, int32_t height, int32_t grid_w, int32_t grid_h, r1_host_scene **out)
{
    if (!out)
        return R1_EINVAL;
    *out = nullptr;
    if (width <= 0 || height <= 0 || kind < R1_SCENE_SMALL || kind > R1_SCENE_GRID)
    {
        r1_set_error("r1_host_scene_create: bad kind/size (%d, %dx%d)", kind, width, height);
        return R1_EINVAL;
    }
    r1_host_scene *s = new (std::nothrow) r1_host_scene();
    if (!s)
        return R1_ENOMEM;
    const float aspect = (float)width / (float)height; // rayweek1.cpp:564
    switch (kind)
    {
    case R1_SCENE_SMALL:
        build_small(*s, aspect);
        break;
    case R1_SCENE_MEDIUM:
        build_medium(*s, aspect);
        break;
    case R1_SCENE_LARGE:
        build_grid(*s, aspect, 30, 16);
        break;
    default:
        if (grid_w <= 0)
            grid_w = 30;
        if (grid_h <= 0)
            grid_h = 16;
        if ((int64_t)grid_w * grid_h > (1 << 21) - 16)
        {
            delete s;
            r1_set_error("r1_host_scene_create: grid %dx%d too large", grid_w, grid_h);
            return R1_ELIMIT;
        }
        build_grid(*s, aspect, grid_w, grid_h);
        break;
    }
    s->pad_to_simd_width();
    s->finish();
    *out = s;
    return R1_OK;
}

extern "C" void r1_host_scene_destroy(r1_host_scene *hs) { delete hs; }
extern "C" const r1_scene *r1_host_scene_spheres(const r1_host_scene *hs) { return hs ? &hs->view : nullptr; }
extern "C" const r1_camera *r1_host_scene_camera(const r1_host_scene *hs) { return hs ? &hs->cam : nullptr; }

// ---- tiles and shards -----------------------------------------------------------------

static int tiles_required(int tile, int size) // rayweek1.cpp:61-68
{
    int n = size / tile;
    if (n * tile < size)
        n++;
    return n;
}

extern "C" int r1_params_check(const r1_params *p)
{
    if (!p || p->width <= 0 || p->height <= 0 || p->spp <= 0 || p->tile_w <= 0 || p->tile_h <= 0 || p->num_shards < 1 ||
        p->shard < 0 || p->shard >= p->num_shards || p->max_bounces < 1 || p->max_bounces > R1_MAX_BOUNCES_LIMIT ||
        p->variant < R1_VARIANT_DEFAULT || p->variant > R1_VARIANT_WAVEFRONT)
    {
        r1_set_error("bad r1_params (size %dx%dx%d, tile %dx%d, shard %d/%d, max_bounces %d, variant %d)", p ? p->width : 0,
                     p ? p->height : 0, p ? p->spp : 0, p ? p->tile_w : 0, p ? p->tile_h : 0, p ? p->shard : 0, p ? p->num_shards : 0,
                     p ? p->max_bounces : 0, p ? p->variant : 0);
        return R1_EINVAL;
    }
    if ((int64_t)p->width * p->height * p->spp >= (int64_t)1 << 31 || p->width > 65535 || p->height > 65535)
    {
        r1_set_error("image %dx%dx%d exceeds 2^31 samples", p->width, p->height, p->spp);
        return R1_ELIMIT;
    }
    return R1_OK;
}

extern "C" int r1_tile_count(const r1_params *p, int32_t *tiles_total, int32_t *tiles_per_shard)
{
    int rc = r1_params_check(p);
    if (rc)
        return rc;
    int t = tiles_required(p->tile_w, p->width) * tiles_required(p->tile_h, p->height);
    if (tiles_total)
        *tiles_total = t;
    if (tiles_per_shard)
        *tiles_per_shard = (t + p->num_shards - 1) / p->num_shards;
    return R1_OK;
}

extern "C" size_t r1_shard_block_bytes(const r1_params *p)
{
    int32_t per = 0;
    if (r1_tile_count(p, nullptr, &per))
        return 0;
    return (size_t)per * p->tile_w * p->tile_h * 3;
}

// One shard's gather record: the dense tile block, padded to a multiple of 8 bytes, then the shard's uint64 ray
// count (8-byte aligned for any tile size: the kernels store it as one 64-bit word).
extern "C" size_t r1_shard_record_bytes(const r1_params *p)
{
    const size_t block = r1_shard_block_bytes(p);
    return block ? ((block + 7u) & ~(size_t)7u) + 8u : 0u;
}

// One whole frame as the batch entry points deliver it: the row-major image, padded to a multiple of 8 bytes, then the
// frame's uint64 ray count.
extern "C" size_t r1_frame_record_bytes(const r1_params *p)
{
    if (r1_params_check(p))
        return 0;
    return (((size_t)p->width * p->height * 3 + 7u) & ~(size_t)7u) + 8u;
}

// ---- output formats (src/common/common.h) ---------------------------------------------

extern "C" int r1_tga_write_rgb24(const char *filename, int32_t width, int32_t height, uint8_t *pixels)
{
    // common.h:86-122: 18-byte header, origin bits 0 (bottom-left), BGR payload; the
    // caller's buffer is left R/B-swapped, as in the reference.
    if (!filename || !pixels || width <= 0 || height <= 0)
        return R1_EINVAL;
    FILE *f = fopen(filename, "wb");
    if (!f)
        return R1_EINVAL;
    uint8_t header[18] = {0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, (uint8_t)(width & 0x00FF), (uint8_t)((width & 0xFF00) >> 8),
                          (uint8_t)(height & 0x00FF), (uint8_t)((height & 0xFF00) >> 8), 24, 0};
    for (int64_t i = 0; i < (int64_t)width * height; ++i)
    {
        uint8_t tmp = pixels[3 * i];
        pixels[3 * i] = pixels[3 * i + 2];
        pixels[3 * i + 2] = tmp;
    }
    fwrite(header, 1, sizeof(header), f);
    fwrite(pixels, 3, (size_t)width * height, f);
    fclose(f);
    return R1_OK;
}

extern "C" int r1_log_results(const char *version, const char *scene, const double *elapsed_seconds, const uint64_t *num_rays,
                              int32_t num_runs)
{
    // common.h:47-77: averages, then `version|%.3fs|%llu|%0.3f mrays/s|`
    if (!version || !scene || !elapsed_seconds || !num_rays || num_runs < 1 || strlen(scene) > 100)
        return R1_EINVAL;
    double el = 0;
    uint64_t rays = 0;
    for (int i = 0; i < num_runs; ++i)
    {
        el += elapsed_seconds[i];
        rays += num_rays[i];
    }
    el /= num_runs;
    rays /= (uint64_t)num_runs;
    char filename[128];
    snprintf(filename, sizeof(filename), "out_%s.txt", scene);
    FILE *f = fopen(filename, "wt");
    if (!f)
        return R1_EINVAL;
    fprintf(f, "%s|", version);
    fprintf(f, "%.3fs|", el);
    fprintf(f, "%llu|", (unsigned long long)rays);
    fprintf(f, "%0.3f mrays/s|", el ? (rays / el / 1000000.0) : 0); // RESULT::get_mrays_per_sec common.h:41-44
    fclose(f);
    return R1_OK;
}
