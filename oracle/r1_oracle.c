/* oracle/r1_oracle.c — TEST INFRASTRUCTURE: CPU restatement of the reference's hot path.
 *
 * NOT product code: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg load libr1_oracle.so, and only as the checker (see r1_oracle.h).
 *
 * Parity status: PINNED against the reference's own functions, bit for bit, through the
 * fixtures in tests/golden/ (generator oracle/gen_golden.py, harness oracle/ref_harness*.cpp).
 *
 * Arithmetic contract.  The reference's Vec3 is a 4-lane SSE register whose operators are
 * lane-wise IEEE mul/add/sub/div (src/step13/mymath.h:154-177); the only fused operations
 * are the explicit FMAs of the sphere sweep (rayweek1.cpp:196, :199 via mymath.h:280-283).
 * This file restates that with scalar floats, is compiled with -ffp-contract=off, and uses
 * fmaf() exactly where the reference uses fma().  All citations are /root/reference paths.
 */
#include "r1_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ---------------------------------------------------------------- RNG (mymath.h:17-73) */

uint32_t r1o_xorshift32(uint32_t *state)
{
    uint32_t x = *state; /* mymath.h:19-24: shifts 13 / 17 / 15 */
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 15;
    *state = x;
    return x;
}

float r1o_rand01(uint32_t *state)
{
    /* mymath.h:29: (x & 0xFFFFFF) * float(1.0 / 16777216.0) */
    return (float)(r1o_xorshift32(state) & 0xFFFFFFu) * (float)(1.0 / 16777216.0);
}

float r1o_rand02(uint32_t *state)
{
    /* mymath.h:34: (x & 0xFFFFFF) / (float)(0xFFFFFF/2 + 1) */
    return (float)(r1o_xorshift32(state) & 0xFFFFFFu) / (float)(0xFFFFFF / 2 + 1);
}

void r1o_rand01_x4(uint32_t state4[4], float out[4])
{
    /* mymath.h:41-56: same generator per lane, cvtepi32_ps, * (float)(1.0/(0xFFFFFF+1)) */
    for (int i = 0; i < 4; ++i)
        out[i] = (float)(int32_t)(r1o_xorshift32(&state4[i]) & 0xFFFFFFu) * (float)(1.0 / (0xFFFFFF + 1));
}

void r1o_rand02_x4(uint32_t state4[4], float out[4])
{
    /* mymath.h:58-73: * (float)(1.0 / (0xFFFFFF / 2 + 1)) */
    for (int i = 0; i < 4; ++i)
        out[i] = (float)(int32_t)(r1o_xorshift32(&state4[i]) & 0xFFFFFFu) * (float)(1.0 / (0xFFFFFF / 2 + 1));
}

/* ------------------------------------------------------- Vec3 (mymath.h:85-127, :154-216) */

typedef struct
{
    float x, y, z;
} v3;

static inline v3 V(float x, float y, float z)
{
    v3 r = {x, y, z};
    return r;
}
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); } /* Vec3*float == float*Vec3 */
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }                     /* mymath.h:179 sign flip */
/* mymath.h:205-207: sum(a*b) = (x + y) + z */
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* mymath.h:211: v * (1.0f / length(v)) */
static inline v3 vunit(v3 v) { return vscale(v, 1.0f / sqrtf(vdot(v, v))); }

static inline v3 v3_from(const float *p) { return V(p[0], p[1], p[2]); }

/* --------------------------------------------------------------------------- tracer */

typedef struct
{
    uint32_t scalar;   /* ThreadData::state  (rayweek1.cpp:91) */
    uint32_t lanes[4]; /* ThreadData::state4 (rayweek1.cpp:92), lanes 0..3 */
} streams;

typedef struct
{
    const r1_scene *sc;
    const r1_camera *cam;
    int32_t max_bounces;
    streams st;
    uint64_t rays; /* ThreadData::out_num_rays */
    /* pass-1 scratch (rayweek1.cpp:174-178; sized to the scene instead of MAX_SPHERES) */
    float *nb;
    float *discr;
    uint32_t *cand;
} tracer;

typedef struct
{
    v3 o, d;
} ray;

/* Ray::Ray normalises (rayweek1.cpp:104-108) */
static inline ray make_ray(v3 o, v3 dir)
{
    ray r;
    r.o = o;
    r.d = vunit(dir);
    return r;
}

/* mymath.h:224-235 */
static v3 random_in_unit_sphere(tracer *t)
{
    v3 p;
    float f[4];
    do
    {
        r1o_rand02_x4(t->st.lanes, f);
        p = vsub(V(f[0], f[1], f[2]), V(1.0f, 1.0f, 1.0f));
    } while (vdot(p, p) >= 1);
    return p;
}

/* rayweek1.cpp:353-362.  g++ evaluates the constructor arguments right to left, so the
 * FIRST draw lands in y and the SECOND in x (pinned by the fixtures). */
static v3 random_in_unit_disk(tracer *t)
{
    v3 p;
    do
    {
        float second_arg = r1o_rand02(&t->st.scalar);
        float first_arg = r1o_rand02(&t->st.scalar);
        p = vsub(V(first_arg, second_arg, 0), V(1, 1, 0));
    } while (vdot(p, p) >= 1.0f);
    return p;
}

/* Camera::getRay rayweek1.cpp:381-386 */
static ray camera_get_ray(tracer *t, float s, float tt)
{
    const r1_camera *c = t->cam;
    v3 rd = vscale(random_in_unit_disk(t), c->lens_radius);
    v3 offset = vadd(vscale(v3_from(c->u), rd.x), vscale(v3_from(c->v), rd.y));
    v3 origin = vadd(v3_from(c->origin), offset);
    v3 dir = vsub(vsub(vadd(vadd(v3_from(c->lower_left), vscale(v3_from(c->horizontal), s)), vscale(v3_from(c->vertical), tt)),
                       v3_from(c->origin)),
                  offset);
    return make_ray(origin, dir);
}

typedef struct
{
    float t;
    v3 p, normal;
    uint32_t index;
} hit_record;

/* Hitable::hit rayweek1.cpp:152-339 */
static int world_hit(tracer *t, ray r, float t_min, float t_max, hit_record *rec)
{
    const r1_scene *sc = t->sc;
    const uint32_t n = sc->count;
    const float *cx = sc->center_x, *cy = sc->center_y, *cz = sc->center_z, *rsq = sc->radius_sq;
    float *all_nb = t->nb, *discriminants = t->discr;

    /* pass 1, rayweek1.cpp:190-202 */
    for (uint32_t i = 0; i < n; ++i)
    {
        const float cox = cx[i] - r.o.x;
        const float coy = cy[i] - r.o.y;
        const float coz = cz[i] - r.o.z;
        const float nb = fmaf(coz, r.d.z, fmaf(coy, r.d.y, cox * r.d.x));
        const float c = fmaf(coz, coz, fmaf(coy, coy, cox * cox)) - rsq[i];
        all_nb[i] = nb;
        discriminants[i] = nb * nb - c;
    }
    /* candidates = sign bit clear, rayweek1.cpp:204-225 */
    uint32_t num_positives = 0;
    for (uint32_t i = 0; i < n; ++i)
    {
        uint32_t bits;
        memcpy(&bits, &discriminants[i], 4);
        if (!(bits >> 31))
            t->cand[num_positives++] = i;
    }

    /* pass 2, rayweek1.cpp:281-314 */
    int hit_index = -1;
    float hit_t = 0;
    for (uint32_t k = 0; k < num_positives; ++k)
    {
        const uint32_t idx = t->cand[k];
        if (sc->inv_radius[idx] == 0)
            continue;
        const float discr_sq = sqrtf(discriminants[idx]);
        const float nb = all_nb[idx];
        float temp = nb - discr_sq;
        if (temp < t_max && temp > t_min)
        {
            t_max = temp;
            hit_t = temp;
            hit_index = (int)idx;
            continue;
        }
        temp = nb + discr_sq;
        if (temp < t_max && temp > t_min)
        {
            t_max = temp;
            hit_t = temp;
            hit_index = (int)idx;
            continue;
        }
    }
    if (hit_index != -1)
    {
        /* rayweek1.cpp:316-322 */
        rec->t = hit_t;
        rec->p = vadd(r.o, vscale(r.d, hit_t));
        rec->normal = vscale(vsub(rec->p, V(cx[hit_index], cy[hit_index], cz[hit_index])), sc->inv_radius[hit_index]);
        rec->index = (uint32_t)hit_index;
    }
    return hit_index != -1;
}

/* rayweek1.cpp:414-417: v - 2 * dot(v, n) * n */
static inline v3 reflect(v3 v, v3 n) { return vsub(v, vscale(n, 2 * vdot(v, n))); }

/* rayweek1.cpp:439-452 */
static int refract(v3 uv, v3 n, float ni_over_nt, v3 *refracted)
{
    float dt = vdot(uv, n);
    float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (discriminant > 0)
    {
        *refracted = vsub(vscale(vsub(uv, vscale(n, dt)), ni_over_nt), vscale(n, sqrtf(discriminant)));
        return 1;
    }
    return 0;
}

/* rayweek1.cpp:454-459 */
static float schlick(float cosine, float ref_idx)
{
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * (float)powf((1 - cosine), 5);
}

/* Material::scatter for the three classes, rayweek1.cpp:403-409 / :427-433 / :470-511 */
static int scatter(tracer *t, ray in, const hit_record *rec, v3 *attenuation, ray *scattered)
{
    const r1_scene *sc = t->sc;
    const uint32_t i = rec->index;
    switch (sc->mat_type[i])
    {
    case R1_MAT_LAMBERTIAN:
    {
        v3 target = vadd(vadd(rec->p, rec->normal), random_in_unit_sphere(t));
        *scattered = make_ray(rec->p, vsub(target, rec->p));
        *attenuation = V(sc->albedo_r[i], sc->albedo_g[i], sc->albedo_b[i]);
        return 1;
    }
    case R1_MAT_METAL:
    {
        v3 reflected = reflect(in.d, rec->normal);
        *scattered = make_ray(rec->p, vadd(reflected, vscale(random_in_unit_sphere(t), sc->mat_param[i])));
        *attenuation = V(sc->albedo_r[i], sc->albedo_g[i], sc->albedo_b[i]);
        return vdot(scattered->d, rec->normal) > 0;
    }
    case R1_MAT_DIELECTRIC:
    {
        const float ref_idx = sc->mat_param[i];
        *attenuation = V(1, 1, 1);
        v3 outward_normal;
        v3 reflected = reflect(in.d, rec->normal);
        float ni_over_nt;
        v3 refracted = V(0, 0, 0);
        float reflect_prob;
        float cosine;
        if (vdot(in.d, rec->normal) > 0)
        {
            outward_normal = vneg(rec->normal);
            ni_over_nt = ref_idx;
            cosine = ref_idx * vdot(in.d, rec->normal);
        }
        else
        {
            outward_normal = rec->normal;
            ni_over_nt = 1.0f / ref_idx;
            cosine = -vdot(in.d, rec->normal);
        }
        if (refract(in.d, outward_normal, ni_over_nt, &refracted))
            reflect_prob = schlick(cosine, ref_idx);
        else
            reflect_prob = 1;
        if (r1o_rand01(&t->st.scalar) < reflect_prob)
            *scattered = make_ray(rec->p, reflected);
        else
            *scattered = make_ray(rec->p, refracted);
        return 1;
    }
    default:
        return 0; /* unreachable: placeholders are never hit (rayweek1.cpp:291) */
    }
}

/* color rayweek1.cpp:515-536 */
static v3 color(tracer *t, ray r, int depth)
{
    ++t->rays;
    hit_record rec;
    if (world_hit(t, r, 0.001f, FLT_MAX, &rec))
    {
        v3 attenuation;
        ray scattered;
        if (depth < t->max_bounces && scatter(t, r, &rec, &attenuation, &scattered))
            return vmul(attenuation, color(t, scattered, depth + 1));
        return V(0, 0, 0);
    }
    float tt = 0.5f * (r.d.y + 1.0f);
    /* lerp mymath.h:212-216: (1 - t) * a + t * b */
    return vadd(vscale(V(1.0f, 1.0f, 1.0f), 1 - tt), vscale(V(0.5f, 0.7f, 1.0f), tt));
}

/* one iteration of the sample loop, rayweek1.cpp:759-762 */
static v3 trace_one(tracer *t, int32_t width, int32_t height, int32_t x, int32_t y)
{
    float j[4];
    r1o_rand01_x4(t->st.lanes, j);
    float u = (j[0] + (float)x) * (1.0f / width);
    float v = (j[1] + (float)y) * (1.0f / height);
    ray r = camera_get_ray(t, u, v);
    return color(t, r, 0);
}

static int tracer_init(tracer *t, const r1_scene *sc, const r1_camera *cam, int32_t max_bounces)
{
    memset(t, 0, sizeof(*t));
    t->sc = sc;
    t->cam = cam;
    t->max_bounces = max_bounces;
    size_t n = sc->count ? sc->count : 1;
    t->nb = (float *)malloc(n * sizeof(float));
    t->discr = (float *)malloc(n * sizeof(float));
    t->cand = (uint32_t *)malloc(n * sizeof(uint32_t));
    return (t->nb && t->discr && t->cand) ? 0 : -1;
}

static void tracer_free(tracer *t)
{
    free(t->nb);
    free(t->discr);
    free(t->cand);
}

/* rayweek1.cpp:765-775 */
static void resolve_pixel(v3 col, int32_t spp, uint8_t *px)
{
    col = vscale(col, (float)(1.0f / spp));
    col = V(sqrtf(col.x), sqrtf(col.y), sqrtf(col.z));
    px[0] = (uint8_t)(int)(col.x * 255.99f);
    px[1] = (uint8_t)(int)(col.y * 255.99f);
    px[2] = (uint8_t)(int)(col.z * 255.99f);
}

static void seed_streams(streams *st, uint32_t seed, uint32_t pixel, uint32_t sample)
{
    r1_sample_seed sd = r1_seed_sample(seed, pixel, sample);
    st->scalar = sd.scalar;
    st->lanes[0] = sd.lane0;
    st->lanes[1] = sd.lane1;
    st->lanes[2] = sd.lane2;
    st->lanes[3] = R1_SEED_LANE3;
}

void r1o_trace_sample(const r1_scene *scene, const r1_camera *cam, int32_t width, int32_t height, int32_t max_bounces,
                      uint32_t seed, int32_t x, int32_t y, int32_t s, float rgb[3], uint32_t *rays)
{
    tracer t;
    if (tracer_init(&t, scene, cam, max_bounces))
        return;
    seed_streams(&t.st, seed, (uint32_t)(y * width + x), (uint32_t)s);
    v3 c = trace_one(&t, width, height, x, y);
    rgb[0] = c.x, rgb[1] = c.y, rgb[2] = c.z;
    *rays = (uint32_t)t.rays;
    tracer_free(&t);
}

/* ------------------------------------------------------------- frame (seeding contract) */

typedef struct
{
    const r1_scene *sc;
    const r1_camera *cam;
    const r1_params *p;
    uint8_t *rgb;
    float *samples;
    int next_row; /* atomic */
    uint64_t rays; /* atomic */
    int failed;
} frame_job;

static void *frame_worker(void *arg)
{
    frame_job *job = (frame_job *)arg;
    const r1_params *p = job->p;
    tracer t;
    if (tracer_init(&t, job->sc, job->cam, p->max_bounces))
    {
        job->failed = 1;
        return 0;
    }
    const int ntx = (p->width + p->tile_w - 1) / p->tile_w;
    int y;
    while ((y = __atomic_fetch_add(&job->next_row, 1, __ATOMIC_RELAXED)) < p->height)
    {
        for (int x = 0; x < p->width; ++x)
        {
            if (p->num_shards > 1)
            {
                int tile = (y / p->tile_h) * ntx + (x / p->tile_w);
                if (tile % p->num_shards != p->shard)
                    continue;
            }
            v3 col = V(0, 0, 0);
            const uint32_t pixel = (uint32_t)(y * p->width + x);
            for (int s = 0; s < p->spp; ++s)
            {
                seed_streams(&t.st, p->seed, pixel, (uint32_t)s);
                const uint64_t before = t.rays;
                v3 c = trace_one(&t, p->width, p->height, x, y);
                col = vadd(col, c);
                if (job->samples)
                {
                    float *o = job->samples + ((size_t)pixel * p->spp + s) * 4;
                    uint32_t nr = (uint32_t)(t.rays - before);
                    o[0] = c.x, o[1] = c.y, o[2] = c.z;
                    memcpy(o + 3, &nr, 4);
                }
            }
            resolve_pixel(col, p->spp, job->rgb + (size_t)pixel * 3);
        }
    }
    __atomic_fetch_add(&job->rays, t.rays, __ATOMIC_RELAXED);
    tracer_free(&t);
    return 0;
}

static int hw_threads(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

int r1o_render_frame(const r1_scene *scene, const r1_camera *cam, const r1_params *params, uint8_t *rgb_out,
                     uint64_t *num_rays_out, float *samples_out, int32_t nthreads)
{
    if (!scene || !cam || !params || !rgb_out || params->width <= 0 || params->height <= 0 || params->spp <= 0)
        return -1;
    r1_params p = *params;
    if (p.num_shards < 1)
        p.num_shards = 1;
    if (p.tile_w <= 0)
        p.tile_w = 32;
    if (p.tile_h <= 0)
        p.tile_h = 32;
    if (nthreads <= 0)
        nthreads = hw_threads();
    if (nthreads > 256)
        nthreads = 256;
    frame_job job = {scene, cam, &p, rgb_out, samples_out, 0, 0, 0};
    pthread_t th[256];
    for (int i = 0; i < nthreads; ++i)
        pthread_create(&th[i], 0, frame_worker, &job);
    for (int i = 0; i < nthreads; ++i)
        pthread_join(th[i], 0);
    if (num_rays_out)
        *num_rays_out = job.rays;
    return job.failed ? -2 : 0;
}

/* ----------------------------------------- sequential-stream paths (reference seeding) */

static int tiles_required(int tile_w, int width) /* rayweek1.cpp:61-68 */
{
    int n = width / tile_w;
    if (n * tile_w < width)
        n++;
    return n;
}

/* render_tile rayweek1.cpp:722-782: streams run on across samples, pixels and tiles */
static void render_tile(tracer *t, int tile_index, int width, int height, int tile_w, int tile_h, int spp, uint8_t *image)
{
    int num_tiles_x = tiles_required(tile_w, width);
    int tile_x = tile_index % num_tiles_x;
    int tile_y = tile_index / num_tiles_x;
    int y0 = tile_y * tile_h, y1 = y0 + tile_h;
    int x0 = tile_x * tile_w, x1 = x0 + tile_w;
    if (x1 > width)
        x1 = width;
    if (y1 > height)
        y1 = height;
    for (int y = y1 - 1; y >= y0; --y)
        for (int x = x0; x < x1; ++x)
        {
            v3 col = V(0, 0, 0);
            for (int s = 0; s < spp; ++s)
                col = vadd(col, trace_one(t, width, height, x, y));
            resolve_pixel(col, spp, image + ((size_t)y * width + x) * 3);
        }
}

int r1o_render_sequential(const r1_scene *scene, const r1_camera *cam, int32_t width, int32_t height, int32_t spp,
                          int32_t max_bounces, uint8_t *rgb_out, uint64_t *num_rays_out)
{
    tracer t;
    if (tracer_init(&t, scene, cam, max_bounces))
        return -2;
    /* rayweek1.cpp:880-881: state = 10001; state4 = _mm_set_epi32(1001, 1003, 1005, 1007) */
    t.st.scalar = 10001;
    t.st.lanes[0] = 1007, t.st.lanes[1] = 1005, t.st.lanes[2] = 1003, t.st.lanes[3] = 1001;
    int tile_w = 32 > width ? width : 32, tile_h = 32 > height ? height : 32; /* rayweek1.cpp:855-864 */
    int num_tiles = tiles_required(tile_w, width) * tiles_required(tile_h, height);
    for (int i = 0; i < num_tiles; ++i)
        render_tile(&t, i, width, height, tile_w, tile_h, spp, rgb_out);
    if (num_rays_out)
        *num_rays_out = t.rays;
    tracer_free(&t);
    return 0;
}

typedef struct
{
    const r1_scene *sc;
    const r1_camera *cam;
    int width, height, spp, max_bounces, tile_w, tile_h, num_tiles;
    uint8_t *rgb;
    int *next_tile;
    uint32_t thread_index;
    uint64_t rays;
    int failed;
} mt_job;

static void *mt_worker(void *arg)
{
    mt_job *job = (mt_job *)arg;
    tracer t;
    if (tracer_init(&t, job->sc, job->cam, job->max_bounces))
    {
        job->failed = 1;
        return 0;
    }
    /* rayweek1.cpp:800-802 */
    const uint32_t i = job->thread_index;
    t.st.scalar = 200 * i + 10001;
    t.st.lanes[0] = 200 * i + 10007, t.st.lanes[1] = 200 * i + 10005, t.st.lanes[2] = 200 * i + 10003, t.st.lanes[3] = 200 * i + 10001;
    int tile;
    while ((tile = __atomic_fetch_add(job->next_tile, 1, __ATOMIC_SEQ_CST)) < job->num_tiles) /* rayweek1.cpp:830-838 */
        render_tile(&t, tile, job->width, job->height, job->tile_w, job->tile_h, job->spp, job->rgb);
    job->rays = t.rays;
    tracer_free(&t);
    return 0;
}

int r1o_render_threads(const r1_scene *scene, const r1_camera *cam, int32_t width, int32_t height, int32_t spp,
                       int32_t max_bounces, int32_t nthreads, uint8_t *rgb_out, uint64_t *num_rays_out)
{
    if (nthreads <= 0)
        nthreads = hw_threads();
    if (nthreads > 256)
        nthreads = 256;
    int tile_w = 32 > width ? width : 32, tile_h = 32 > height ? height : 32;
    int num_tiles = tiles_required(tile_w, width) * tiles_required(tile_h, height);
    int next_tile = 0;
    mt_job jobs[256];
    pthread_t th[256];
    for (int i = 0; i < nthreads; ++i)
    {
        mt_job j = {scene, cam, width, height, spp, max_bounces, tile_w, tile_h, num_tiles, rgb_out, &next_tile, (uint32_t)i, 0, 0};
        jobs[i] = j;
        pthread_create(&th[i], 0, mt_worker, &jobs[i]);
    }
    uint64_t rays = 0;
    int failed = 0;
    for (int i = 0; i < nthreads; ++i)
    {
        pthread_join(th[i], 0);
        rays += jobs[i].rays; /* rayweek1.cpp:809-813 */
        failed |= jobs[i].failed;
    }
    if (num_rays_out)
        *num_rays_out = rays;
    return failed ? -2 : 0;
}

/* ------------------------------------------------ BASELINE config 1: step1 semantics */
/* All citations in this section: /root/reference/src/step1/rayweek1.cpp */

static uint32_t s1_state; /* :32 */

static float s1_rand(void) /* :44-47 */
{
    return (float)(r1o_xorshift32(&s1_state) & 0xFFFFFFu) / 16777216.0f;
}

static inline float s1_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* :137-140 */
static inline v3 s1_unit(v3 v)                                                        /* :73-77 */
{
    float k = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return V(v.x * k, v.y * k, v.z * k);
}
static inline v3 s1_cross(v3 a, v3 b) /* :142-145 */
{
    return V(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}

static v3 s1_random_in_unit_sphere(void) /* :158-170; g++ draws z, y, x */
{
    v3 p;
    do
    {
        float a3 = s1_rand(), a2 = s1_rand(), a1 = s1_rand();
        p = vsub(vscale(V(a1, a2, a3), 2.0f), V(1, 1, 1));
    } while ((p.x * p.x + p.y * p.y + p.z * p.z) >= 1);
    return p;
}

static v3 s1_random_in_unit_disk(void) /* :321-330; g++ draws y, x */
{
    v3 p;
    do
    {
        float a2 = s1_rand(), a1 = s1_rand();
        p = vsub(vscale(V(a1, a2, 0), 2.0f), V(1, 1, 0));
    } while (s1_dot(p, p) >= 1.0f);
    return p;
}

typedef struct
{
    v3 center;
    float radius;
    int mat;
    v3 albedo;
    float param;
} s1_sphere;

typedef struct
{
    v3 origin, lower_left, horizontal, vertical, u, v, w;
    float lens_radius;
} s1_camera;

static void s1_camera_init(s1_camera *c, v3 lookfrom, v3 lookat, v3 vup, float vfov, float aspect, float aperture, float focus_dist)
{
    /* :334-347 */
    c->lens_radius = aperture / 2;
    /* The fixture binary calls libm's tanf at run time (Camera::init is not inlined there);
     * a compile-time fold (MPFR) differs from glibc by 1 ulp for vfov = 60, so keep the
     * argument opaque to the optimiser. */
    volatile float theta = vfov * (float)M_PI / 180;
    float half_height = tanf(theta / 2);
    float half_width = aspect * half_height;
    c->origin = lookfrom;
    c->w = s1_unit(vsub(lookfrom, lookat));
    c->u = s1_unit(s1_cross(vup, c->w));
    c->v = s1_cross(c->w, c->u);
    c->lower_left = vsub(vsub(vsub(c->origin, vscale(c->u, half_width * focus_dist)), vscale(c->v, half_height * focus_dist)),
                         vscale(c->w, focus_dist));
    c->horizontal = vscale(c->u, 2 * half_width * focus_dist);
    c->vertical = vscale(c->v, 2 * half_height * focus_dist);
}

static ray s1_get_ray(const s1_camera *c, float s, float t) /* :348-353, direction NOT normalised (:181-185) */
{
    v3 rd = vscale(s1_random_in_unit_disk(), c->lens_radius);
    v3 offset = vadd(vscale(c->u, rd.x), vscale(c->v, rd.y));
    ray r;
    r.o = vadd(c->origin, offset);
    r.d = vsub(vsub(vadd(vadd(c->lower_left, vscale(c->horizontal, s)), vscale(c->vertical, t)), c->origin), offset);
    return r;
}

typedef struct
{
    float t;
    v3 p, normal;
    const s1_sphere *sph;
} s1_hit;

static int s1_sphere_hit(const s1_sphere *sp, ray r, float t_min, float t_max, s1_hit *rec) /* :249-280 */
{
    v3 oc = vsub(r.o, sp->center);
    float a = s1_dot(r.d, r.d);
    float b = s1_dot(oc, r.d);
    float c = s1_dot(oc, oc) - sp->radius * sp->radius;
    float discriminant = b * b - a * c;
    if (discriminant > 0)
    {
        float temp = (-b - sqrtf(b * b - a * c)) / a;
        if (temp < t_max && temp > t_min)
        {
            rec->t = temp;
            rec->p = vadd(r.o, vscale(r.d, temp));
            rec->normal = vscale(vsub(rec->p, sp->center), 1.0f / sp->radius);
            rec->sph = sp;
            return 1;
        }
        temp = (-b + sqrtf(b * b - a * c)) / a;
        if (temp < t_max && temp > t_min)
        {
            rec->t = temp;
            rec->p = vadd(r.o, vscale(r.d, temp));
            rec->normal = vscale(vsub(rec->p, sp->center), 1.0f / sp->radius);
            rec->sph = sp;
            return 1;
        }
    }
    return 0;
}

static int s1_list_hit(const s1_sphere *list, int n, ray r, float t_min, float t_max, s1_hit *rec) /* :301-318 */
{
    s1_hit temp;
    int hit_anything = 0;
    float closest = t_max;
    for (int i = 0; i < n; ++i)
        if (s1_sphere_hit(&list[i], r, t_min, closest, &temp))
        {
            hit_anything = 1;
            closest = temp.t;
            *rec = temp;
        }
    return hit_anything;
}

static inline v3 s1_reflect(v3 v, v3 n) { return vsub(v, vscale(n, 2 * s1_dot(v, n))); } /* :373-376 */

static int s1_refract(v3 v, v3 n, float ni_over_nt, v3 *refracted) /* :398-410 */
{
    v3 uv = s1_unit(v);
    float dt = s1_dot(uv, n);
    float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1 - dt * dt);
    if (discriminant > 0)
    {
        *refracted = vsub(vscale(vsub(uv, vscale(n, dt)), ni_over_nt), vscale(n, sqrtf(discriminant)));
        return 1;
    }
    return 0;
}

static inline float s1_length(v3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); } /* :63-66 */

static int s1_scatter(ray in, const s1_hit *rec, v3 *att, ray *sc)
{
    const s1_sphere *sp = rec->sph;
    if (sp->mat == R1_MAT_LAMBERTIAN) /* :362-368 */
    {
        v3 target = vadd(vadd(rec->p, rec->normal), s1_random_in_unit_sphere());
        sc->o = rec->p;
        sc->d = vsub(target, rec->p);
        *att = sp->albedo;
        return 1;
    }
    if (sp->mat == R1_MAT_METAL) /* :386-392 */
    {
        v3 reflected = s1_reflect(s1_unit(in.d), rec->normal);
        sc->o = rec->p;
        sc->d = vadd(reflected, vscale(s1_random_in_unit_sphere(), sp->param));
        *att = sp->albedo;
        return s1_dot(sc->d, rec->normal) > 0;
    }
    /* Dielectric :429-474 */
    const float ref = sp->param;
    *att = V(1, 1, 1);
    v3 outward_normal;
    v3 reflected = s1_reflect(in.d, rec->normal);
    float ni_over_nt;
    v3 refracted = V(0, 0, 0);
    float reflect_prob, cosine;
    if (s1_dot(in.d, rec->normal) > 0)
    {
        outward_normal = vneg(rec->normal);
        ni_over_nt = ref;
        cosine = ref * s1_dot(in.d, rec->normal) / s1_length(in.d);
    }
    else
    {
        outward_normal = rec->normal;
        ni_over_nt = 1.0f / ref;
        cosine = -s1_dot(in.d, rec->normal) / s1_length(in.d);
    }
    if (s1_refract(in.d, outward_normal, ni_over_nt, &refracted))
        reflect_prob = schlick(cosine, ref); /* :412-417, same formula as step13 */
    else
        reflect_prob = 1;
    sc->o = rec->p;
    if (s1_rand() < reflect_prob)
        sc->d = reflected;
    else
        sc->d = refracted;
    return 1;
}

static v3 s1_color(const s1_sphere *list, int n, ray r, int depth, uint32_t *ray_count) /* :498-519 */
{
    ++(*ray_count);
    s1_hit rec;
    if (s1_list_hit(list, n, r, 0.001f, FLT_MAX, &rec))
    {
        v3 att;
        ray sc;
        if (depth < 50 && s1_scatter(r, &rec, &att, &sc))
            return vmul(att, s1_color(list, n, sc, depth + 1, ray_count));
        return V(0, 0, 0);
    }
    v3 ud = s1_unit(r.d);
    float t = 0.5f * (ud.y + 1.0f);
    return vadd(vscale(V(1.0f, 1.0f, 1.0f), 1 - t), vscale(V(0.5f, 0.7f, 1.0f), t)); /* lerp :152-155 */
}

int r1o_step1_small(int32_t width, int32_t height, int32_t spp, uint8_t *rgb_out, uint32_t *num_rays_out)
{
    /* create_small_scene :537-558 */
    s1_sphere list[5] = {
        {{0, 0, -1}, 0.5f, R1_MAT_LAMBERTIAN, {0.1f, 0.2f, 0.5f}, 0},
        {{0, -100.5f, -1}, 100.0f, R1_MAT_LAMBERTIAN, {0.8f, 0.8f, 0}, 0},
        {{1, 0, -1}, 0.5f, R1_MAT_METAL, {0.8f, 0.6f, 0.2f}, 0.3f},
        {{-1, 0, -1}, 0.5f, R1_MAT_DIELECTRIC, {1, 1, 1}, 1.5f},
        {{-1, 0, -1}, -0.45f, R1_MAT_DIELECTRIC, {1, 1, 1}, 1.5f},
    };
    s1_camera cam;
    s1_camera_init(&cam, V(2, 1, 2), V(0, 0, 0), V(0, 1, 0), 60, (float)width / (float)height, 0.1f, 5.0f);

    /* benchmark :689-727 */
    s1_state = 1236787;
    const int nx = width, ny = height;
    uint32_t num_rays = 0;
    for (int y = ny - 1; y >= 0; --y)
        for (int x = 0; x < nx; ++x)
        {
            v3 col = V(0, 0, 0);
            for (int s = 0; s < spp; ++s)
            {
                float u = (float)((x + s1_rand()) / nx);
                float v = (float)((y + s1_rand()) / ny);
                ray r = s1_get_ray(&cam, u, v);
                col = vadd(col, s1_color(list, 5, r, 0, &num_rays));
            }
            float k = 1.0f / (float)spp; /* operator/= :87-94 */
            col = V(col.x * k, col.y * k, col.z * k);
            col = V(sqrtf(col.x), sqrtf(col.y), sqrtf(col.z));
            uint8_t *px = rgb_out + ((size_t)y * nx + x) * 3;
            px[0] = (uint8_t)(int)(col.x * 255.99f);
            px[1] = (uint8_t)(int)(col.y * 255.99f);
            px[2] = (uint8_t)(int)(col.z * 255.99f);
        }
    if (num_rays_out)
        *num_rays_out = num_rays;
    return 0;
}
