// r1_trace.hpp — the device side of the hot path (gfx950, wave64): device functions + the persistent trace kernel template.
// Included by the translation units that instantiate the kernel (r1_trace_tree_small.hip, r1_trace_tree_big.hip,
// r1_trace_sweep_small.hip, r1_trace_sweep_big.hip) and by r1_aux_kernels.hip (wavefront variant, resolve, assemble), so that an
// experiment on one kernel family rebuilds one file (hygiene) and `make -j` builds the families side by side.
// Everything here has internal linkage (anonymous namespace / templates).
#ifndef R1_TRACE_HPP
#define R1_TRACE_HPP

// r1_trace.hpp — hand-written HIP for gfx950 (MI355X, wave64): the per-pixel/per-sample
// path-tracing hot path of rays1bench step13.
//
// What it replaces (all /root/reference/src/step13/):
//   render_tile's pixel x sample loop      rayweek1.cpp:722-782   -> r1_trace_kernel + r1_resolve_kernel
//   Camera::getRay / random_in_unit_disk   rayweek1.cpp:381-386 / :353-362
//   color() bounce recursion               rayweek1.cpp:515-536   -> flattened register loop
//   Hitable::hit (AVX2 sweep + resolve)    rayweek1.cpp:152-339   -> sweep_prefilter / sweep_reference / sweep_bvh
//   Lambertian/Metal/Dielectric::scatter   rayweek1.cpp:403-409 / :427-433 / :470-511
//   TileRenderScheduler (atomic tile pop)  rayweek1.cpp:785-842   -> persistent waves + global sample queue
//
// Design (DESIGN.md §4 has the long form and the measurements):
//   * one LANE per live path; a wave keeps its 64 lanes busy by refilling finished lanes from a
//     global sample queue (guided chunks of R1_CHUNK_MIN..R1_CHUNK samples per atomic), so the
//     bounce-count divergence of color() (1..51 rays per sample) costs no lanes;
//   * three interchangeable hit tests, bit-identical results (include/rays1.h R1_VARIANT_*): the
//     reference's form over every sphere (sweep_reference), the grouped exhaustive sweep
//     (sweep_prefilter, next two points) and a per-lane walk of a conservative box tree in front
//     of the reference's per-sphere test (sweep_bvh; the reference has no such structure);
//   * exhaustive sweep: the sphere table is wave-uniform: it is read with SCALAR loads (two alternating sets of
//     s_load_dwordx16) and fed to the VALU as SGPR-pair operands of v_pk_fma_f32 — two spheres
//     per instruction, no LDS round trip, no VGPRs.  (Scenes above 1023 spheres stream the table
//     through LDS tiles instead: the scalar cache cannot sustain a 3 MB stream.);
//   * pass 1 is a conservative prefilter, 7 FMA + 1 compare per ray-sphere test (11 in the
//     reference's form) = 4.5 VALU instructions per sphere per 64 rays; flagged (ray, sphere)
//     pairs of the whole wave are compacted in LDS and re-tested 64 at a time in the reference's
//     exact arithmetic, so results are bit-identical to the reference's candidate rule (sign bit
//     of its discriminant) and closest-hit rule (strict compares, lowest index on ties);
//   * attenuation is applied in the reference's right-nested order a0*(a1*(...*sky)) by
//     keeping the hit indices of a path in a packed per-lane LDS stack;
//   * per-sample radiance goes to HBM (16 B/sample) and a second kernel sums the samples of
//     a pixel in sample order — the reference's order — which keeps pixels deterministic.
//
// Arithmetic contract: identical operation order to oracle/r1_oracle.c (which is pinned to
// the reference bit-for-bit); no implicit FMA contraction; IEEE sqrt and division.  The only
// known difference is powf(x,5) (glibc, <1 ulp) vs an exactly rounded x^5 here.

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "r1_device.h"
#include "../../include/rays1_seed.h"

#pragma clang fp contract(off)

namespace
{

struct V3
{
    float x, y, z;
};

// read-only tables are addressed through the constant address space so that a wave-uniform
// index turns into scalar loads (SMEM) instead of vector loads
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const f4 __attribute__((address_space(4))) *cf4_ptr;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef const f16 __attribute__((address_space(4))) *cf16_ptr;
typedef uint32_t __attribute__((address_space(1))) r1_gu32; // a word of global memory (explicit: keeps rarely used stores / loads from becoming generic ones)

// The kernel's arguments read again from the kernarg segment (scalar loads, constant cache) at the place of use, through a pointer the
// optimiser cannot trace back to the prologue's loads.  hipcc keeps every argument field the loop uses in an SGPR from the prologue on;
// this kernel needs ~190 of them, ~90 live in lanes of two VGPRs, and every use of such a one is a v_readlane — a VALU issue slot, and
// whole 16-register tuples at a time (the fast-division constants of start_sample alone: ~90 v_readlane per loop iteration, 478 in the
// kernel).  A field read through R1_FRESH_ARGS is an s_load where it is used and occupies an SGPR only there.
#ifndef R1_FRESH
#define R1_FRESH 1
#endif
typedef const __attribute__((address_space(4))) uint32_t *r1_kargs_ptr;
static_assert(sizeof(R1TraceArgs) % 4 == 0 && __is_trivially_copyable(R1TraceArgs), "fresh_args copies the argument block word by word");
union R1ArgWords
{
    R1TraceArgs a;
    uint32_t w[sizeof(R1TraceArgs) / 4];
    __device__ R1ArgWords() {}
};
// (a pointer that arrives as two loaded words is a generic pointer to the optimiser — flat loads.  The kernel's pointer arguments all
//  point to global memory: they are fetched as what they are, pointers into address space 1, the way clang passes them to a kernel)
template <typename T>
__device__ __forceinline__ T *global_ptr_at(r1_kargs_ptr k, const size_t offset)
{
    typedef __attribute__((address_space(1))) T *gptr;
    return (T *)*(const __attribute__((address_space(4))) gptr *)((const __attribute__((address_space(4))) char *)k + offset);
}
__device__ __forceinline__ void fresh_args(R1ArgWords &u)
{
    r1_kargs_ptr k = (r1_kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr(); // (the kernel's one argument starts the segment)
    asm volatile("" : "+s"(k));
#pragma unroll
    for (uint32_t i = 0; i < sizeof(R1TraceArgs) / 4; ++i)
        u.w[i] = k[i]; // (only the words that are used afterwards are fetched)
#define R1_G(f) u.a.f = global_ptr_at<__typeof__(*u.a.f)>(k, __builtin_offsetof(R1TraceArgs, f));
    R1_G(scene.sweep) R1_G(scene.exact) R1_G(scene.shade) R1_G(scene.exact_g) R1_G(scene.members) R1_G(scene.mat) R1_G(scene.bvh_nodes) R1_G(scene.bvh_prims)
    R1_G(scene.bvh_ids) R1_G(queue) R1_G(samples) R1_G(num_rays) R1_G(gstack) R1_G(stats) R1_G(land_cnt) R1_G(land.out) R1_G(land.rays_dst)
    R1_G(land.frame_rays) R1_G(land.frame_left) R1_G(land.clear_heads) R1_G(land.owed_spill) R1_G(land.error) R1_G(bvh_wide) R1_G(bvh_entry)
#undef R1_G
}
#define R1_FRESH_ARGS(L)                                                                                                                  \
    R1ArgWords L##_words;                                                                                                                 \
    fresh_args(L##_words);                                                                                                                \
    const R1TraceArgs &L = L##_words.a;

// Correctly rounded sqrt / division.  NOT __fsqrt_rn/__fdiv_rn: without
// OCML_BASIC_ROUNDED_OPERATIONS hipcc maps __fsqrt_rn to the approximate v_sqrt_f32.  Plain
// sqrtf() and `/` are IEEE under -fhip-fp32-correctly-rounded-divide-sqrt (set in the Makefile).
__device__ __forceinline__ float ieee_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ieee_div(float a, float b) { return a / b; }

__device__ __forceinline__ V3 mk(float x, float y, float z)
{
    V3 r;
    r.x = x, r.y = y, r.z = z;
    return r;
}
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vscale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// mymath.h:205-207: sum(a*b) = (x + y) + z
__device__ __forceinline__ float vdot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// mymath.h:211: v * (1.0f / length(v)); Ray::Ray normalises every direction (rayweek1.cpp:107)
__device__ __forceinline__ V3 vunit(V3 v) { return vscale(v, ieee_div(1.0f, ieee_sqrt(vdot(v, v)))); }
__device__ __forceinline__ V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

// mymath.h:17-25
__device__ __forceinline__ uint32_t xorshift32(uint32_t &state)
{
    uint32_t x = state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 15;
    state = x;
    return x;
}
// mymath.h:27-35 and the x4 forms :41-73 — all exact: a 24-bit integer times a power of two
__device__ __forceinline__ float rand01(uint32_t &s) { return (float)(xorshift32(s) & 0xFFFFFFu) * (1.0f / 16777216.0f); }
// myrand02 - 1 (mymath.h:32-35, :58-73; used as `myrand02_x4(state) - Vec3(1,1,1)` :229 and in
// random_in_unit_disk): the product with 2^-23 is exact, so the fused form rounds once, exactly
// where the reference's subtraction rounds
__device__ __forceinline__ float rand02_minus1(uint32_t &s) { return __fmaf_rn((float)(xorshift32(s) & 0xFFFFFFu), 1.0f / 8388608.0f, -1.0f); }

template <bool BIG>
struct IdxType
{
    typedef uint16_t type;
};
template <>
struct IdxType<true>
{
    typedef uint32_t type;
};

struct Path
{
    V3 o, d;                        // Ray::_origin, Ray::_dir (unit)
    uint32_t s_scalar, s0, s1, s2;  // ThreadData::state, lanes 0..2 of ThreadData::state4
    uint32_t k;                     // local sample index (output slot)
    // (color() invocations of the sample = depth + 1 when the path ends — every level counts one ray, rayweek1.cpp:517, and the depth
    //  grows by one per level that goes on: path_rays)
    int depth;                      // color()'s depth argument
    int sp;                         // entries on the attenuation stack
};

// color() invocations of a path that has just ended (shade_level returned true)
__device__ __forceinline__ uint32_t path_rays(const Path &p) { return (uint32_t)p.depth + 1u; }

// mymath.h:224-235
__device__ __forceinline__ V3 random_in_unit_sphere(Path &p)
{
    V3 r;
    do
    {
        r = mk(rand02_minus1(p.s0), rand02_minus1(p.s1), rand02_minus1(p.s2));
    } while (vdot(r, r) >= 1);
    return r;
}

// exactly rounded x^5 (the reference calls powf(x, 5), rayweek1.cpp:458)
__device__ __forceinline__ float pow5(float x)
{
    double d = (double)x;
    double d2 = d * d;
    return (float)(d2 * d2 * d);
}

// ---- the exact ray/sphere test: pass 1 + pass 2 of Hitable::hit for ONE sphere -----------
// rayweek1.cpp:192-202 (co, nb, c, discr with the reference's two FMA chains), :204 (sign
// bit), :294-313 (roots, strict compares, t_max shrinks).  Candidates must be presented in
// increasing index order (ties keep the earlier sphere).
__device__ __forceinline__ void exact_test(const f4 e, uint32_t idx, const V3 o, const V3 d, float &t_max, int &hit_index)
{
    const float cox = e.x - o.x;
    const float coy = e.y - o.y;
    const float coz = e.z - o.z;
    const float nb = __fmaf_rn(coz, d.z, __fmaf_rn(coy, d.y, cox * d.x));
    const float c = __fmaf_rn(coz, coz, __fmaf_rn(coy, coy, cox * cox)) - e.w;
    const float discr = nb * nb - c;
    if (!(__float_as_uint(discr) >> 31))
    {
        const float discr_sq = ieee_sqrt(discr);
        float temp = nb - discr_sq;
        if (temp < t_max && temp > 0.001f)
        {
            t_max = temp;
            hit_index = (int)idx;
        }
        else
        {
            temp = nb + discr_sq;
            if (temp < t_max && temp > 0.001f)
            {
                t_max = temp;
                hit_index = (int)idx;
            }
        }
    }
}

// ---- sweep, reference form: every active sphere through exact_test ------------------------
__device__ __forceinline__ void sweep_reference(const R1DeviceScene &S, const V3 o, const V3 d, float &t_max, int &hit_index)
{
    const cf4_ptr tab = (cf4_ptr)S.exact;
    for (uint32_t i = 0; i < S.n_active; ++i)
        exact_test(tab[i], i, o, d, t_max, hit_index); // uniform index => scalar loads
}

// ---- sweep, prefilter form -----------------------------------------------------------------
// For unit d:  discr = (co.d)^2 - |co|^2 + r^2  with co = c - o
//            = (c.d - o.d)^2 - (|o|^2 - 2 c.o) - (|c|^2 - r^2)
// The test is applied to GROUPS of <= R1_GROUP_MAX nearby spheres through a bounding sphere
// (g, R) that contains every member (host, r1_capi.cpp build_groups):
// Per ray:    negod = -(o.d), m2o = -2 o, oo' = |o|^2 (1 - 2^-15)
// Per group:  Kp = (|g|^2 - R^2) - 2^-15 (C^2 + R^2), rounded down, C^2 >= |g|^2 and every |c_i|^2
// Test:       fma(nb', nb', -t') >= Kp   with two 3-FMA chains nb', t'        => 7 FMA + 1 compare
// If the REFERENCE flags member i (its fp32 discriminant has a clear sign bit), the ray's line
// passes within sqrt(r_i^2 + E1) of c_i (E1 = the reference form's fp32 error), hence within
// R + E1/(2 r_i) of g, i.e. the group's real-number discriminant is >= -R E1 / r_i; with
// R <= 3.5 r_i (R1_GROUP_RATIO) and the error of this form itself that is > -2^-15 (C^2 + R^2 +
// |o|^2), so the group is flagged here; cooperative_exact then applies the
// reference's own rule to each member.
// What ONE candidate sphere offers a ray, independent of the other candidates: pass 2 of
// Hitable::hit (rayweek1.cpp:294-313) accepts t1 = nb - s when t1 > t_min, otherwise
// t2 = nb + s when t2 > t_min, each only if it is below the running t_max; since t2 >= t1 a
// sphere whose t1 fails `< t_max` cannot pass with t2, so the sphere's offer is fixed and the
// loop result is the minimum offer, ties to the lowest index (strict `<` keeps the earlier one).
// Returns FLT_MAX for "no offer".
__device__ __forceinline__ float exact_offer(const f4 e, const V3 o, const V3 d)
{
    const float cox = e.x - o.x;
    const float coy = e.y - o.y;
    const float coz = e.z - o.z;
    const float nb = __fmaf_rn(coz, d.z, __fmaf_rn(coy, d.y, cox * d.x));
    const float c = __fmaf_rn(coz, coz, __fmaf_rn(coy, coy, cox * cox)) - e.w;
    const float discr = nb * nb - c;
    float offer = FLT_MAX;
    if (!(__float_as_uint(discr) >> 31))
    {
        const float discr_sq = ieee_sqrt(discr);
        const float t1 = nb - discr_sq;
        const float t = (t1 > 0.001f) ? t1 : nb + discr_sq;
        if (t > 0.001f && t < FLT_MAX)
            offer = t;
    }
    return offer;
}

// Wave-cooperative exact phase.  Flag counts per lane are very uneven (mean ~3 groups, max of
// 64 lanes ~10, a few rays skim a whole row of spheres), so the (ray, group) pairs of the whole
// wave are compacted into one dense LDS list and re-tested 64 member slots at a time by
// whichever lane comes next: full-width trips instead of mostly-empty ones.  Offers are
// combined per ray with a 64-bit LDS atomic min on {t bits, sphere index}: closest hit, ties to
// the lowest index — the reference's rule.  All of this must be called by ALL 64 lanes.
// IDX = uint16_t (<= 1023 groups: 6 lane bits + 10 group bits per pair) or uint32_t (6 + 26).
template <typename IDX>
struct PairBits
{
    static constexpr int value = sizeof(IDX) == 2 ? 10 : 26;
    static constexpr int cap = sizeof(IDX) == 2 ? R1_PAIR_CAP_SMALL : R1_PAIR_CAP; // pairs of one wave
};

// Inclusive prefix sum over the 64 lanes with data-parallel-primitive moves (no LDS round trips: the __shfl_up form was six dependent
// ds_bpermute_b32 plus compare / select / add each): Kogge-Stone inside every row of 16 lanes (row_shr 1, 2, 4, 8; lanes shifted in from
// outside the row read 0), then lane 15 of a row added to the next row (row_bcast:15 into rows 1 and 3) and lane 31 to the upper half
// (row_bcast:31 into rows 2 and 3).  All 64 lanes must be active.
__device__ __forceinline__ int wave_inclusive_scan(int v, const int /*lane*/)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}

// `total` (lane, group) pairs are in `pairs`.  SLOTS == R1_GROUP_MAX: each pair expands to its
// member slots — 16 pairs x 4 members fill a trip, a lane takes pair (base + lane) / 4, member
// lane % 4.  SLOTS == 1: the pairs are single-sphere groups, one lane each.
// `rays` (optional): this wave's ray table in LDS, component c of lane l at rays[c * R1_BLOCK + l] — a lane then fetches its
// owner's ray with six ds_read_b32 (8 issue cycles each) instead of six ds_bpermute_b32 (24).  The table overlays the flag
// words / candidate lists, which are dead once the pair list is built.
template <bool STATS, typename IDX, int SLOTS>
__device__ __forceinline__ void exact_trips(const R1DeviceScene &S, const V3 o, const V3 d, const int total, const IDX *pairs,
                                            unsigned long long *best /* this wave's [64] */, const int lane, unsigned long long *wstat,
                                            const float *rays = nullptr)
{
    constexpr int IDX_BITS = PairBits<IDX>::value;
    constexpr int SHIFT = SLOTS == 1 ? 0 : 2;
    static_assert(SLOTS == 1 || (SLOTS == 4 && R1_GROUP_MAX == 4), "member-slot mapping assumes 4 spheres per group");
    const int slots = total * SLOTS;
    if (STATS)
        wstat[2] += (unsigned long long)((slots + 63) >> 6);
    for (int base = 0; base < slots; base += 64) // wave-uniform trip count
    {
        const int j = (base + lane) >> SHIFT;
        const bool have_pair = j < total;
        const uint32_t pr = have_pair ? (uint32_t)pairs[j] : ((uint32_t)lane << IDX_BITS);
        const int owner = (int)(pr >> IDX_BITS);
        const uint32_t grp = pr & ((1u << IDX_BITS) - 1u);
        const uint32_t slot = grp * R1_GROUP_MAX + (SLOTS == 1 ? 0u : (uint32_t)(lane & (R1_GROUP_MAX - 1)));
        const uint32_t idx = have_pair ? S.members[slot] : 0xFFFFFFFFu;
#if R1_EXACT_G
        const f4 sphere = ((const f4 *)S.exact_g)[slot]; // (no pair: group 0's slots, fetched and not used)
#endif
        V3 ro, rd;
        if (rays)
        {
#if R1_RAYS_AOS
            const float *w = rays + (owner >> 3) * R1_BLOCK + (owner & 7) * 8;
            const f4 r0 = *(const f4 *)w, r1 = *(const f4 *)(w + 4);
            ro = mk(r0.x, r0.y, r0.z), rd = mk(r0.w, r1.x, r1.y);
#else
            ro = mk(rays[0 * R1_BLOCK + owner], rays[1 * R1_BLOCK + owner], rays[2 * R1_BLOCK + owner]);
            rd = mk(rays[3 * R1_BLOCK + owner], rays[4 * R1_BLOCK + owner], rays[5 * R1_BLOCK + owner]);
#endif
        }
        else
        {
            ro.x = __shfl(o.x, owner, 64), ro.y = __shfl(o.y, owner, 64), ro.z = __shfl(o.z, owner, 64);
            rd.x = __shfl(d.x, owner, 64), rd.y = __shfl(d.y, owner, 64), rd.z = __shfl(d.z, owner, 64);
        }
        if (idx != 0xFFFFFFFFu)
        {
#if R1_EXACT_G
            const float t = exact_offer(sphere, ro, rd);
#else
            const float t = exact_offer(((const f4 *)S.exact)[idx], ro, rd);
#endif
            if (t < FLT_MAX)
                atomicMin(&best[owner], ((unsigned long long)__float_as_uint(t) << 32) | idx);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// writes this lane's ray into the wave's LDS ray table (see exact_trips); returns the wave's base pointer
__device__ __forceinline__ const float *publish_rays(uint32_t *scratch /* [>= 6][R1_BLOCK] */, const V3 o, const V3 d, const int tid)
{
    float *t = (float *)scratch;
    __builtin_amdgcn_wave_barrier(); // every lane has read what it needed from the scratch rows
#if R1_RAYS_AOS
    // eight words per ray, side by side: the wave's 64 columns of rows 0..7 hold rays 8 r .. 8 r + 7 in row r (ray_at)
    float *w = t + (tid & ~63) + ((tid & 63) >> 3) * R1_BLOCK + (tid & 7) * 8;
    *(f4 *)w = f4{o.x, o.y, o.z, d.x};
    *(f4 *)(w + 4) = f4{d.y, d.z, 0.0f, 0.0f};
#else
    t[0 * R1_BLOCK + tid] = o.x, t[1 * R1_BLOCK + tid] = o.y, t[2 * R1_BLOCK + tid] = o.z;
    t[3 * R1_BLOCK + tid] = d.x, t[4 * R1_BLOCK + tid] = d.y, t[5 * R1_BLOCK + tid] = d.z;
#endif
    __builtin_amdgcn_wave_barrier();
    return t + (tid & ~63);
}

// Append path (big scenes): every lane holds `cnt` flagged group indices in cand[j][tid].
template <bool STATS, typename IDX>
__device__ __forceinline__ void cooperative_exact(const R1DeviceScene &S, const V3 o, const V3 d, int cnt, const uint32_t *cand,
                                                  IDX *pairs /* this wave's [R1_PAIR_CAP] */, unsigned long long *best /* this wave's [64] */,
                                                  const int tid, const int lane, unsigned long long *wstat)
{
    constexpr int IDX_BITS = PairBits<IDX>::value;
    const int incl = wave_inclusive_scan(cnt, lane);
    const int total = __builtin_amdgcn_readlane(incl, 63);
    const int excl = incl - cnt;
    for (int j = 0; j < cnt; ++j)
        pairs[excl + j] = (IDX)(((uint32_t)lane << IDX_BITS) | cand[j * R1_BLOCK + tid]);
    static_assert(R1_CAND_CAP >= 6, "the ray table needs six rows of the candidate list");
    const float *rays = publish_rays(const_cast<uint32_t *>(cand), o, d, tid);
    exact_trips<STATS, IDX, R1_GROUP_MAX>(S, o, d, total, pairs, best, lane, wstat, rays);
}

// Bit path (<= 1023 groups): every lane holds `nwords` 32-bit flag words in words[w][tid]; bit
// 31 - p of word w flags group gbase + 32 w + p.  More than R1_PAIR_CAP pairs in the wave (a
// wave of rays that all skim rows of spheres) are worked off half a word at a time: 16 flags per
// lane always fit.
template <bool STATS, typename IDX>
__device__ __forceinline__ void cooperative_bits(const R1DeviceScene &S, const V3 o, const V3 d, const uint32_t nwords, const uint32_t gbase,
                                                 const uint32_t *words, IDX *pairs, unsigned long long *best, const int tid, const int lane,
                                                 unsigned long long *wstat)
{
    constexpr int IDX_BITS = PairBits<IDX>::value;
    // flags of multi-member groups (ids < n_multi) and of single spheres are counted apart: the
    // former go to the front of `pairs`, the latter to its back, and are worked off with 4 and 1
    // member slot per pair.  Both counts ride one prefix sum (16 bits each).
    int cnt = 0;
    for (uint32_t w = 0; w < nwords; ++w)
    {
        const uint32_t word = words[w * R1_BLOCK + tid];
        const int first = (int)(gbase + 32u * w);      // group of bit 31
        const int nm = (int)S.n_multi - first;         // groups of this word below n_multi
        const uint32_t multi_mask = nm >= 32 ? 0xFFFFFFFFu : (nm <= 0 ? 0u : ~(0xFFFFFFFFu >> nm));
        cnt += __popc(word & multi_mask) + (__popc(word & ~multi_mask) << 16);
    }
    const int incl = wave_inclusive_scan(cnt, lane);
    const int totals = __builtin_amdgcn_readlane(incl, 63);
    const int total_m = totals & 0xFFFF, total_s = totals >> 16;
    if (STATS)
        wstat[9] += (unsigned long long)((cnt & 0xFFFF) + (cnt >> 16));
    if (totals == 0)
        return;
    constexpr int CAP = PairBits<IDX>::cap;
    if (total_m + total_s <= CAP)
    {
        int pos_m = (incl - cnt) & 0xFFFF;
        int pos_s = CAP - 1 - ((incl - cnt) >> 16); // singles grow down from the end
        for (uint32_t w = 0; w < nwords; ++w)
        {
            uint32_t word = words[w * R1_BLOCK + tid];
            while (word)
            {
                const int p = __clz((int)word);
                word &= ~(0x80000000u >> p);
                const uint32_t g = gbase + 32u * w + (uint32_t)p;
                const IDX v = (IDX)(((uint32_t)lane << IDX_BITS) | g);
                if (g < S.n_multi)
                    pairs[pos_m++] = v;
                else
                    pairs[pos_s--] = v;
            }
        }
        static_assert(R1_BIT_WORDS >= (R1_RAYS_AOS ? 8 : 6) && R1_CAND_CAP >= 8, "the ray table needs six (eight: side-by-side form) rows of the flag words");
        const float *rays = publish_rays(const_cast<uint32_t *>(words), o, d, tid); // the words of this batch are consumed
        exact_trips<STATS, IDX, R1_GROUP_MAX>(S, o, d, total_m, pairs, best, lane, wstat, rays);
        exact_trips<STATS, IDX, 1>(S, o, d, total_s, pairs + (CAP - total_s), best, lane, wstat, rays);
        return;
    }
    // more pairs than the list holds: work the flag words off in segments of CAP / 64 bits, whose
    // flags always fit (64 lanes x SEG bits)
    constexpr uint32_t SEG = (uint32_t)CAP / 64u, PER_WORD = 32u / SEG;
    for (uint32_t seg = 0; seg < PER_WORD * nwords; ++seg)
    {
        const uint32_t w = seg / PER_WORD, part = seg - w * PER_WORD;
        uint32_t word = words[w * R1_BLOCK + tid] & (((1u << SEG) - 1u) << (32u - SEG * (part + 1u)));
        const int c2 = __popc(word);
        const int incl2 = wave_inclusive_scan(c2, lane);
        const int total2 = __builtin_amdgcn_readlane(incl2, 63);
        int pos = incl2 - c2;
        while (word)
        {
            const int p = __clz((int)word);
            word &= ~(0x80000000u >> p);
            pairs[pos++] = (IDX)(((uint32_t)lane << IDX_BITS) | (gbase + 32u * w + (uint32_t)p));
        }
        __builtin_amdgcn_wave_barrier();
        exact_trips<STATS, IDX, R1_GROUP_MAX>(S, o, d, total2, pairs, best, lane, wstat);
    }
}

// ---- sweep, prefilter form -----------------------------------------------------------------
// (formula and slack: see the comment block above)  Called by all 64 lanes; lanes with
// alive == false never flag a candidate but help in the cooperative exact phase.
template <bool STATS, typename IDX, bool BLOCK_SYNC>
__device__ __forceinline__ void sweep_prefilter(const R1DeviceScene &S, const bool alive, const V3 o, const V3 d, float &t_max,
                                                int &hit_index, uint32_t *cand /* flag words [R1_BIT_WORDS][R1_BLOCK], or (big scenes) flagged groups [R1_CAND_CAP][R1_BLOCK] */,
                                                IDX *pairs /* [R1_BLOCK/64][R1_PAIR_CAP] */,
                                                unsigned long long *best /* [R1_BLOCK] */, f4 *tile /* BLOCK_SYNC: [2][R1_TILE_F4] */,
                                                const int tid, unsigned long long *wstat)
{
    const int lane = tid & 63;
    IDX *wpairs = pairs + (tid >> 6) * PairBits<IDX>::cap;
    unsigned long long *wbest = best + (tid & ~63);
    const unsigned long long NONE = ~0ull;

    const float negod = -__fmaf_rn(o.z, d.z, __fmaf_rn(o.y, d.y, o.x * d.x));
    const float oo = __fmaf_rn(o.z, o.z, __fmaf_rn(o.y, o.y, o.x * o.x));
    // dead lanes: t' = +inf => q = -inf (or NaN) => never >= Kp
    const float oo_adj = alive ? __fmaf_rn(oo, -0x1p-15f, oo) : __builtin_inff();
    const float mx = -2.0f * o.x, my = -2.0f * o.y, mz = -2.0f * o.z;

    wbest[lane] = NONE;
    int cnt = 0;

    // Pair layout (r1_capi.cpp): 8 floats per two spheres {cx0 cx1 cy0 cy1 cz0 cz1 Kp0 Kp1}, so
    // that every v_pk_fma_f32 takes an aligned SGPR pair straight from the scalar load: two
    // spheres per VALU instruction, 3.5 + 1 instructions per sphere.  A chunk = 8 spheres =
    // two s_load_dwordx16; the NEXT chunk is requested before the current one is evaluated
    // (the table carries one never-candidate chunk of padding behind the last real one).
    const v2f dxx = {d.x, d.x}, dyy = {d.y, d.y}, dzz = {d.z, d.z};
    const v2f mxx = {mx, mx}, myy = {my, my}, mzz = {mz, mz};
    const v2f nod = {negod, negod}, ooa = {oo_adj, oo_adj};
    const cf16_ptr tab = (cf16_ptr)S.sweep;
    const uint32_t chunks = S.n_sweep >> 3; // r1_capi.cpp pads to 8 spheres + one prefetch chunk

#define R1_PAIR(P, L, B)                                                                                               \
    {                                                                                                                  \
        const v2f cx = {L[B + 0], L[B + 1]}, cy = {L[B + 2], L[B + 3]}, cz = {L[B + 4], L[B + 5]};                     \
        const v2f nb = __builtin_elementwise_fma(cz, dzz, __builtin_elementwise_fma(cy, dyy, __builtin_elementwise_fma(cx, dxx, nod))); \
        const v2f t = __builtin_elementwise_fma(cz, mzz, __builtin_elementwise_fma(cy, myy, __builtin_elementwise_fma(cx, mxx, ooa))); \
        const v2f q = __builtin_elementwise_fma(nb, nb, -t);                                                           \
        c[2 * P] = q.x >= L[B + 6];                                                                                    \
        c[2 * P + 1] = q.y >= L[B + 7];                                                                                \
        any |= __ballot(c[2 * P]) | __ballot(c[2 * P + 1]);                                                            \
    }
#define R1_CHUNK_EVAL(L0, L1, CH)                                                                                      \
    {                                                                                                                  \
        bool c[8];                                                                                                     \
        unsigned long long any = 0;                                                                                    \
        R1_PAIR(0, L0, 0) R1_PAIR(1, L0, 8) R1_PAIR(2, L1, 0) R1_PAIR(3, L1, 8)                                        \
        if (any) /* wave-uniform: scalar branch */                                                                     \
        {                                                                                                              \
            /* a chunk adds at most 8 entries per lane: make room first, then append unchecked */                      \
            if (__ballot(cnt > R1_CAND_CAP - 8))                                                                       \
            {                                                                                                          \
                cooperative_exact<STATS, IDX>(S, o, d, cnt, cand, wpairs, wbest, tid, lane, wstat);                    \
                cnt = 0;                                                                                               \
            }                                                                                                          \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) if (c[u])                                                    \
            {                                                                                                          \
                cand[cnt * R1_BLOCK + tid] = (uint32_t)(8 * (CH) + u);                                                 \
                ++cnt;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
    }
    if (!BLOCK_SYNC)
    {
        // Flags are kept as BITS, one per group, shifted into a per-lane word: no branch, no
        // per-flag bookkeeping in the sweep; a word goes to LDS every 32 groups and the wave works a
        // batch of R1_BIT_WORDS words (256 groups) off at a time in cooperative_bits.  The word
        // collects the SIGN of q - Kp (one packed subtract per pair of groups + one v_alignbit_b32
        // per group: bits = 2 bits + sign; 13.4 issue cycles per pair against 17.2 for v_cmp +
        // v_addc, profiles/r01/isa_issue_costs.txt) and is inverted when stored: q >= Kp <=> the
        // difference is not negative (a float difference has the exact sign; equal gives +0).
#define R1_FLAG(RV) bits = __builtin_amdgcn_alignbit(bits, __float_as_uint(RV), 31);
#define R1_PAIR_BITS(L, B)                                                                                             \
    {                                                                                                                  \
        const v2f cx = {L[B + 0], L[B + 1]}, cy = {L[B + 2], L[B + 3]}, cz = {L[B + 4], L[B + 5]};                     \
        const v2f nb = __builtin_elementwise_fma(cz, dzz, __builtin_elementwise_fma(cy, dyy, __builtin_elementwise_fma(cx, dxx, nod))); \
        const v2f t = __builtin_elementwise_fma(cz, mzz, __builtin_elementwise_fma(cy, myy, __builtin_elementwise_fma(cx, mxx, ooa))); \
        const v2f q = __builtin_elementwise_fma(nb, nb, -t);                                                           \
        const v2f kp = {L[B + 6], L[B + 7]};                                                                           \
        const v2f r = q - kp;                                                                                          \
        R1_FLAG(r.x) R1_FLAG(r.y)                                                                                      \
    }
#if R1_SWEEP_PAIRS2
        // two pairs of groups side by side, statement by statement: the chains of one pair fill the wait states hipcc otherwise pads the
        // other's dependent v_pk_fma_f32 with (two s_nop per pair in the one-pair form)
#define R1_PAIR2_BITS(LA, BA, LB, BB)                                                                                  \
    {                                                                                                                  \
        const v2f cxa = {LA[BA + 0], LA[BA + 1]}, cya = {LA[BA + 2], LA[BA + 3]}, cza = {LA[BA + 4], LA[BA + 5]};      \
        const v2f cxb = {LB[BB + 0], LB[BB + 1]}, cyb = {LB[BB + 2], LB[BB + 3]}, czb = {LB[BB + 4], LB[BB + 5]};      \
        v2f nba = __builtin_elementwise_fma(cxa, dxx, nod), nbb = __builtin_elementwise_fma(cxb, dxx, nod);            \
        v2f ta = __builtin_elementwise_fma(cxa, mxx, ooa), tb = __builtin_elementwise_fma(cxb, mxx, ooa);              \
        nba = __builtin_elementwise_fma(cya, dyy, nba), nbb = __builtin_elementwise_fma(cyb, dyy, nbb);                \
        ta = __builtin_elementwise_fma(cya, myy, ta), tb = __builtin_elementwise_fma(cyb, myy, tb);                    \
        nba = __builtin_elementwise_fma(cza, dzz, nba), nbb = __builtin_elementwise_fma(czb, dzz, nbb);                \
        ta = __builtin_elementwise_fma(cza, mzz, ta), tb = __builtin_elementwise_fma(czb, mzz, tb);                    \
        const v2f qa = __builtin_elementwise_fma(nba, nba, -ta), qb = __builtin_elementwise_fma(nbb, nbb, -tb);        \
        const v2f kpa = {LA[BA + 6], LA[BA + 7]}, kpb = {LB[BB + 6], LB[BB + 7]};                                      \
        const v2f ra = qa - kpa, rb = qb - kpb;                                                                        \
        R1_FLAG(ra.x) R1_FLAG(ra.y) R1_FLAG(rb.x) R1_FLAG(rb.y)                                                        \
    }
#define R1_CHUNK_BITS(L0, L1) {R1_PAIR2_BITS(L0, 0, L0, 8) R1_PAIR2_BITS(L1, 0, L1, 8)}
#else
#define R1_CHUNK_BITS(L0, L1) {R1_PAIR_BITS(L0, 0) R1_PAIR_BITS(L0, 8) R1_PAIR_BITS(L1, 0) R1_PAIR_BITS(L1, 8)}
#endif
        uint32_t bits = 0;
        uint32_t nwords = 0, gbase = 0; // wave-uniform
        // two register sets (A, B) alternate: while one chunk is evaluated the next one loads
        f16 a0 = tab[0], a1 = tab[1];
        uint32_t ch = 0;
        for (; ch + 1 < chunks; ch += 2)
        {
            // Scalar loads return out of order, so lgkmcnt can only be waited to 0: make set A land
            // BEFORE set B is requested (the empty asm "uses" A, which places the wait here), then
            // evaluate A with B in flight and no further wait.
            asm volatile("" ::"s"(a0[0]), "s"(a1[0]));
            const f16 b0 = tab[2 * ch + 2], b1 = tab[2 * ch + 3];
            __builtin_amdgcn_sched_barrier(0);
            R1_CHUNK_BITS(a0, a1)
            asm volatile("" ::"s"(b0[0]), "s"(b1[0]));
            a0 = tab[2 * ch + 4], a1 = tab[2 * ch + 5];
            __builtin_amdgcn_sched_barrier(0);
            R1_CHUNK_BITS(b0, b1)
            if (ch & 2u) // 32 groups since the last word
            {
                cand[nwords * R1_BLOCK + tid] = ~bits;
                if (++nwords == R1_BIT_WORDS)
                {
                    cooperative_bits<STATS, IDX>(S, o, d, nwords, gbase, cand, wpairs, wbest, tid, lane, wstat);
                    nwords = 0;
                    gbase = (ch + 2) * 8;
                }
            }
        }
        if (ch < chunks) // odd number of chunks: the last one is already in set A
        {
            R1_CHUNK_BITS(a0, a1)
            ++ch;
        }
        if (ch & 3u) // a partial word: left-align it
        {
            cand[nwords * R1_BLOCK + tid] = ~bits << (32u - 8u * (ch & 3u));
            ++nwords;
        }
        if (STATS)
        {
            wstat[5] += __builtin_readcyclecounter() - wstat[15];
            wstat[15] = __builtin_readcyclecounter();
        }
        cooperative_bits<STATS, IDX>(S, o, d, nwords, gbase, cand, wpairs, wbest, tid, lane, wstat);
#undef R1_CHUNK_BITS
#undef R1_PAIR_BITS
#undef R1_PAIR2_BITS
#undef R1_FLAG
    }
    else
    {
        // Big scenes (BASELINE config 5: 100 k spheres = 3.2 MB of table).  The scalar cache
        // sustains under 1 B/clk/CU on misses (measured: 100 k spheres ran at a quarter of the
        // small-scene rate), so the table goes through the VECTOR memory path instead: the
        // workgroup stages R1_TILE_SPHERES-sphere tiles into LDS with coalesced 16-byte loads
        // (each byte fetched once per workgroup and shared by its 256 rays), double-buffered,
        // one barrier per tile, and every wave reads the tile back with broadcast ds_read_b128.
        // The whole workgroup runs the bounce loop in lock step (see the kernel).
        const f4 *gsrc = (const f4 *)S.sweep;
        const uint32_t n_tiles = S.n_sweep / R1_TILE_SPHERES;
        f4 r0 = gsrc[tid], r1 = gsrc[R1_BLOCK + tid];
        __syncthreads(); // every wave has left the previous sweep's last tile
        tile[tid] = r0, tile[R1_BLOCK + tid] = r1;
        __syncthreads();
        for (uint32_t ti = 0; ti < n_tiles; ++ti)
        {
            const f4 *cur = tile + (ti & 1u) * R1_TILE_F4;
            const f4 *nsrc = gsrc + (size_t)(ti + 1) * R1_TILE_F4; // the table carries one padding tile
            r0 = nsrc[tid], r1 = nsrc[R1_BLOCK + tid];
#pragma unroll 2
            for (uint32_t cc = 0; cc < R1_TILE_SPHERES / 8; ++cc)
            {
                const f4 *q4 = cur + cc * 8;
                const f4 x0 = q4[0], x1 = q4[1], x2 = q4[2], x3 = q4[3], x4 = q4[4], x5 = q4[5], x6 = q4[6], x7 = q4[7];
                const f16 l0 = __builtin_shufflevector(__builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7),
                                                       __builtin_shufflevector(x2, x3, 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8,
                                                       9, 10, 11, 12, 13, 14, 15);
                const f16 l1 = __builtin_shufflevector(__builtin_shufflevector(x4, x5, 0, 1, 2, 3, 4, 5, 6, 7),
                                                       __builtin_shufflevector(x6, x7, 0, 1, 2, 3, 4, 5, 6, 7), 0, 1, 2, 3, 4, 5, 6, 7, 8,
                                                       9, 10, 11, 12, 13, 14, 15);
                R1_CHUNK_EVAL(l0, l1, ti * (R1_TILE_SPHERES / 8) + cc)
            }
            f4 *nxt = tile + ((ti + 1u) & 1u) * R1_TILE_F4;
            nxt[tid] = r0, nxt[R1_BLOCK + tid] = r1;
            __syncthreads();
        }
    }
#undef R1_CHUNK_EVAL
#undef R1_PAIR
    if (BLOCK_SYNC)
        cooperative_exact<STATS, IDX>(S, o, d, cnt, cand, wpairs, wbest, tid, lane, wstat);
    const unsigned long long key = wbest[lane];
    if (key != NONE)
    {
        t_max = __uint_as_float((uint32_t)(key >> 32));
        hit_index = (int)(uint32_t)(key & 0xFFFFFFFFull);
    }
}

// ---- sweep through the optional spatial index (R1_VARIANT_BVH) -------------
// Per-lane ordered traversal of the binary box tree built by r1_bvh.cpp.  Only the choice of
// spheres presented to exact_offer differs from the exhaustive sweeps; the boxes are inflated
// by pad = A |o - C|^2 + K (conservative with respect to the reference's fp32 test, see
// r1_bvh.cpp) and a subtree is skipped only if its inflated box is missed or lies entirely
// beyond the closest offer so far.  Result = minimum offer, ties to the lowest sphere index —
// the reference's in-order rule (see exact_offer).  The traversal stack lives in LDS,
// [entry][thread]; the builder bounds the tree depth by R1_BVH_STACK.
#define R1_BVH_DONE 0xFFFFFFFFu

// STATS (diagnostic build): wstat[2] wave trips of the node loop, [3] wave trips of the leaf
// pair loop, [9] node visits summed over lanes, [5] sphere-pair tests summed over lanes, [16] leaf trips
// summed over lanes (published in stats slot 14).
// One child box {m, e} against the ray, inflated by the node's pad (pa = pad * |1/d| per axis, computed once
// per node visit for both children; pad = A |o - C|^2 + K, conservative with respect to the reference's fp32
// sphere test AND to this test's own rounding: r1_bvh.cpp).  Slab form a = m / d - o / d (one fused
// multiply-add per axis, oi = o * (1/d) rounded once per ray), b = e |1/d| + pa: 17 VALU instructions per
// box against 28 for the round-1 form that measured |m - o|^2 per box.  Scalar arithmetic on purpose: the
// packed form (both children per v_pk_* instruction) costs the same issue cycles on this chip
// (profiles/r01/isa_issue_costs.txt) but more VGPRs.
__device__ __forceinline__ bool bvh_box(const float mx, const float my, const float mz, const float ex, const float ey, const float ez,
                                        const V3 pa, const V3 oi, const V3 inv, const V3 ainv, const float best, float &t_near)
{
    const float ax = __fmaf_rn(mx, inv.x, -oi.x), ay = __fmaf_rn(my, inv.y, -oi.y), az = __fmaf_rn(mz, inv.z, -oi.z);
    const float bx = __fmaf_rn(ex, ainv.x, pa.x), by = __fmaf_rn(ey, ainv.y, pa.y), bz = __fmaf_rn(ez, ainv.z, pa.z);
    // NaN (an axis the ray does not move along: inf - inf, 0 x inf) drops out of fmaxf/fminf: no constraint
    const float tn = fmaxf(fmaxf(ax - bx, ay - by), az - bz);
    const float tf = fminf(fminf(ax + bx, ay + by), az + bz);
    t_near = tn;
    // (three compares instead of tn <= fminf(tf, best): fminf makes the compiler canonicalise `best` every trip; a NaN tn or tf
    //  — every axis unconstrained — fails them either way)
    return tn <= tf && tn <= best && tf >= 0.0f;
}

// Up to TWO sphere pairs (four spheres) of a leaf for this lane's ray: first pass 1 of Hitable::hit for all
// four (rayweek1.cpp:192-202: nb, discr), then pass 2 (:294-313: square root, roots, strict compares) only
// for the spheres whose discriminant has a clear sign bit (:204), one per trip of a wave-uniform loop.  A
// lane rarely holds more than one such sphere in a leaf, so the wave runs the expensive half ~2 times per
// four spheres instead of four times (it used to cost 155 VALU instructions per pair, two thirds of them
// pass 2).  Same offers as exact_offer, same update rule (minimum offer, ties to the lowest sphere index).
// `pairs` = 1 or 2; an odd sphere's partner has radius_sq = -inf (discriminant -inf: never flagged).
__device__ __forceinline__ void leaf_quad(const float4 *__restrict__ prims, const uint32_t *__restrict__ ids, const uint32_t first,
                                          const uint32_t pairs, const V3 o, const V3 d, float &best, uint32_t &best_id)
{
    const float4 a0 = prims[2 * (size_t)first], a1 = prims[2 * (size_t)first + 1];
    float4 b0 = a0, b1 = a1;
    if (pairs > 1u)
        b0 = prims[2 * (size_t)first + 2], b1 = prims[2 * (size_t)first + 3];
    float nb[4], ds[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
        const float4 p0 = q < 2 ? a0 : b0, p1 = q < 2 ? a1 : b1;
        const float cx = (q & 1) ? p0.y : p0.x, cy = (q & 1) ? p0.w : p0.z, cz = (q & 1) ? p1.y : p1.x, rsq = (q & 1) ? p1.w : p1.z;
        const float cox = cx - o.x, coy = cy - o.y, coz = cz - o.z;
        nb[q] = __fmaf_rn(coz, d.z, __fmaf_rn(coy, d.y, cox * d.x));
        const float c = __fmaf_rn(coz, coz, __fmaf_rn(coy, coy, cox * cox)) - rsq;
        ds[q] = nb[q] * nb[q] - c;
    }
    // bit q of `mask`: slot q's discriminant has a clear sign bit (rayweek1.cpp:204).  The four sign bits are shifted together with
    // v_alignbit_b32 ((hi:lo) >> 31 = hi << 1 | sign of lo): 4 + 3 instructions instead of 12 for compare + select + or per slot.
    uint32_t sgn = __float_as_uint(ds[3]) >> 31;
    sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(ds[2]), 31u);
    sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(ds[1]), 31u);
    sgn = __builtin_amdgcn_alignbit(sgn, __float_as_uint(ds[0]), 31u);
    uint32_t mask = ~sgn & (pairs < 2u ? 3u : 15u);
    // Straight-line body, selects instead of branches: every lane of the leaf runs every trip (a lane without a flagged sphere left
    // computes on slot 3's numbers and is kept from the update by `have`), and the sphere's index is fetched before its offer is
    // known to count.  The branched form (only lanes with a flagged sphere, the index only for an offer in range) cost seven register
    // moves per trip for the loop-carried best / best_id and two exec-mask regions.
    while (__ballot(mask != 0u)) // wave-uniform
    {
        const bool have = mask != 0u;
        const uint32_t q = (uint32_t)__builtin_ctz(mask | 8u); // the lowest flagged slot; 3 for a lane that has none left
        mask &= mask - 1u;
        const float n_ = q == 0u ? nb[0] : (q == 1u ? nb[1] : (q == 2u ? nb[2] : nb[3]));
        const float d_ = q == 0u ? ds[0] : (q == 1u ? ds[1] : (q == 2u ? ds[2] : ds[3]));
        const uint32_t id = ids[2 * (size_t)first + q];
        const float root = ieee_sqrt(d_);
        const float t1 = n_ - root;
        const float t = (t1 > 0.001f) ? t1 : n_ + root;
        // (bitwise on purpose: && / || become nested exec-mask regions)
        const bool upd = have & (t > 0.001f) & (t < FLT_MAX) & ((t < best) | ((t == best) & (id < best_id)));
        best = upd ? t : best;
        best_id = upd ? id : best_id;
    }
}

// Traversal-stack entries.  Big scenes: the 32-bit child reference as it is.  Small scenes (the kernels that keep the node table in LDS:
// <= 256 nodes, <= 1023 spheres = 512 pairs): 16 bits — bit 15 leaf, bits 12..14 the pair count, bits 0..11 the node or first pair —
// which halves the stack in LDS (4 KB instead of 8 for the large scene's tree: together with the 10-word attenuation stack and the
// node table 22 KB per workgroup = seven workgroups per CU with the whole attenuation stack of round 2 in LDS).
__device__ __forceinline__ void trav_put(uint32_t *trav, const int at, const uint32_t ref) { trav[at] = ref; }
__device__ __forceinline__ uint32_t trav_get(const uint32_t *trav, const int at) { return trav[at]; }
// (the small-scene kernels walk with the 16-bit form throughout: the child references are converted once, when the workgroup copies
// the node table into LDS — r1_ref16 — so a push is a 16-bit store and a pop a zero-extending load)
__device__ __forceinline__ void trav_put(uint16_t *trav, const int at, const uint32_t ref) { trav[at] = (uint16_t)ref; }
__device__ __forceinline__ uint32_t trav_get(const uint16_t *trav, const int at) { return trav[at]; }
__device__ __forceinline__ uint32_t r1_ref16(const uint32_t ref) { return ((ref >> 16) & 0xF000u) | (ref & 0x0FFFu); }

// Traversal state of one lane.  It lives in registers ACROSS the outer loop of the trace kernel
// (carry-over, below); the stack entries are in LDS, [entry][thread].
struct Trav
{
    uint32_t cur;     // node or leaf reference being visited, R1_BVH_DONE when the walk is complete
    int sp;           // entries on the traversal stack
    float best;       // closest offer so far (FLT_MAX: none)
    uint32_t best_id; // its sphere (0xFFFFFFFF: none)
};

__device__ __forceinline__ void trav_start(Trav &t)
{
    t.cur = 0u, t.sp = 0, t.best = FLT_MAX, t.best_id = 0xFFFFFFFFu;
}

#ifndef R1_CARRY_DIV
#define R1_CARRY_DIV 6u // a walk phase ends when at most 1 / R1_CARRY_DIV of the live lanes are still walking (2 / 3 / 4 / 6 / 8: 30.3 / 31.6 / 31.9 / 32.1 / 32.0 Grays/s)
#endif
// Advances every lane whose walk is not complete.  CARRY = false: until all walks are complete (the
// classic while-while loop).  CARRY = true: returns as soon as at most a quarter of the `n_alive` live
// lanes of the wave are still walking; those lanes keep their state and go on in the next call, next
// to the fresh rays of the lanes that have been shaded and refilled meanwhile.  The longest of 64
// walks no longer sets the trip count of the loops for everybody (a wave made 14.3 node trips for a mean
// of 7.0 visits per lane), at the price of shading ~3/4 of the wave at a time.  Called by all 64 lanes.
// The walk is while-while: all lanes on inner nodes descend until each has reached a leaf or finished, then the leaves
// are tested.  (Round 2 also carried a voting form — one step per trip, of the kind most walking lanes wait for: lane
// utilisation of the node / leaf steps 0.53 / 0.69 -> 0.74 / 0.75, and 5 % SLOWER once the node table sat in LDS, because
// the vote costs ~20 VALU instructions per trip; removed in round 3)
// LN: the node table is read from `lnodes`, the workgroup's copy in LDS (the trace kernel on small scenes), instead of S.bvh_nodes.
// ENTRY: `entry_at` (null for every ray but a primary one) points at the reference the walk goes on at after the root step instead of
// the root's inner child: the deepest node every primary ray of the tile stays under (r1_capi.cpp compute_entries).
template <bool STATS, bool CARRY, bool LN, typename TS, bool ENTRY = false>
__device__ __forceinline__ void bvh_advance(const R1DeviceScene &S, const V3 o, const V3 d, Trav &tv, TS *trav, const int tid,
                                            const uint32_t n_alive, unsigned long long *wstat, const float4 *lnodes /* LDS */, const uint32_t top = 0u /* !LN: nodes [0, top) are in lnodes */,
                                            const uint32_t *entry_tab = nullptr, const uint32_t entry_idx = 0xFFFFFFFFu, const uint16_t *entry_lds = nullptr /* LDS; null: entry_tab */)
{
    const float4 *__restrict__ nodes = S.bvh_nodes;
    const float4 *__restrict__ prims = S.bvh_prims;
    const uint32_t *__restrict__ ids = S.bvh_ids;
    // v_rcp_f32 (1 ulp) is enough here: the reciprocals only feed the conservative box test, whose
    // pad budgets for its own rounding (r1_bvh.cpp)
    const V3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const V3 ainv = mk(fabsf(inv.x), fabsf(inv.y), fabsf(inv.z));
    const V3 oi = mk(o.x * inv.x, o.y * inv.y, o.z * inv.z);
    const float rx = o.x - S.bvh_centre[0], ry = o.y - S.bvh_centre[1], rz = o.z - S.bvh_centre[2];
    const float r2 = __fmaf_rn(rz, rz, __fmaf_rn(ry, ry, rx * rx)); // |o - C|^2 of pad = A r2 + K
    V3 pa_ray = mk(0.0f, 0.0f, 0.0f);
    if (LN)
    {
        const float pad_ray = lnodes[3].x * r2; // the tree's A (every node carries it)
        pa_ray = mk(pad_ray * ainv.x, pad_ray * ainv.y, pad_ray * ainv.z);
    }
    float best = tv.best;
    uint32_t best_id = tv.best_id;
    uint32_t cur = tv.cur;
    int sp = tv.sp;
    // child references: bit 31 leaf / bits 28..30 pair count / 28 bits of index — or, in the LDS copy of a small scene's table,
    // bit 15 / bits 12..14 / 12 bits (see trav_put)
    constexpr uint32_t LEAF_BIT = LN ? 0x8000u : 0x80000000u, INDEX_MASK = LN ? 0x0FFFu : 0x0FFFFFFFu;
    constexpr int COUNT_SHIFT = LN ? 12 : 28;
    // one visit of the inner node `cur`: both children's boxes, nearer child next, farther one on the stack (or the stack's top if neither is hit)
    auto visit_node = [&]() {
        if (STATS)
        {
            wstat[9] += 1;
            if ((tid & 63) == __ffsll((long long)__ballot(1)) - 1)
                wstat[2] += 1;
        }
#if R1_BVH4
        if (LN)
        {
            // a 4-wide node (r1_capi.cpp build_wide): two child PAIRS in the binary node's row form — {m0x m1x m0y m1y} {m0z m1z e0x e1x}
            // {e0y e1y e0z e1z}, twice — and {ref[4]}.  Four box tests; every hit becomes ONE word — the upper half of its entry distance
            // (clamped at 0: the bits of a float >= 0 order as integers) over its 16-bit reference, all ones for a miss — the four words
            // are sorted (5 min / max pairs), the nearest is visited next and the others go on the stack far-to-near.
            // (the second pair is fetched after the first is tested — the index passes through the asm statement together with the first
            //  pair's words: 24 live node floats at once do not fit the 72 registers of the 7-wave build)
            uint32_t at = 7u * cur;
            const uint4 rf = *(const uint4 *)(lnodes + at + 6u);
            float t0, t1, t2, t3;
            float4 q0 = lnodes[at + 0u], q1 = lnodes[at + 1u], q2 = lnodes[at + 2u];
            const bool g0 = bvh_box(q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, pa_ray, oi, inv, ainv, best, t0);
            const bool g1 = bvh_box(q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, pa_ray, oi, inv, ainv, best, t1);
            uint32_t k0 = g0 ? ((__float_as_uint(fmaxf(t0, 0.0f)) & 0xFFFF0000u) | rf.x) : 0xFFFFFFFFu;
            uint32_t k1 = g1 ? ((__float_as_uint(fmaxf(t1, 0.0f)) & 0xFFFF0000u) | rf.y) : 0xFFFFFFFFu;
            asm volatile("" : "+v"(at), "+v"(k0), "+v"(k1));
            q0 = lnodes[at + 3u], q1 = lnodes[at + 4u], q2 = lnodes[at + 5u];
            const bool g2 = bvh_box(q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, pa_ray, oi, inv, ainv, best, t2);
            const bool g3 = bvh_box(q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, pa_ray, oi, inv, ainv, best, t3);
            uint32_t k2 = g2 ? ((__float_as_uint(fmaxf(t2, 0.0f)) & 0xFFFF0000u) | rf.z) : 0xFFFFFFFFu;
            uint32_t k3 = g3 ? ((__float_as_uint(fmaxf(t3, 0.0f)) & 0xFFFF0000u) | rf.w) : 0xFFFFFFFFu;
            uint32_t lo, hi;
            lo = min(k0, k1), hi = max(k0, k1), k0 = lo, k1 = hi;
            lo = min(k2, k3), hi = max(k2, k3), k2 = lo, k3 = hi;
            lo = min(k0, k2), hi = max(k0, k2), k0 = lo, k2 = hi;
            lo = min(k1, k3), hi = max(k1, k3), k1 = lo, k3 = hi;
            lo = min(k1, k2), hi = max(k1, k2), k1 = lo, k2 = hi;
            if (k3 != 0xFFFFFFFFu)
                trav_put(trav, sp * R1_BLOCK + tid, k3 & 0xFFFFu), ++sp;
            if (k2 != 0xFFFFFFFFu)
                trav_put(trav, sp * R1_BLOCK + tid, k2 & 0xFFFFu), ++sp;
            if (k1 != 0xFFFFFFFFu)
                trav_put(trav, sp * R1_BLOCK + tid, k1 & 0xFFFFu), ++sp;
            if (k0 != 0xFFFFFFFFu)
                cur = k0 & 0xFFFFu;
            else if (sp > 0)
                cur = trav_get(trav, --sp * R1_BLOCK + tid);
            else
                cur = R1_BVH_DONE;
            return;
        }
#endif
        // {m0x m1x m0y m1y} {m0z m1z e0x e1x} {e0y e1y e0z e1z} {A K child0 child1}
        float4 q0, q1, q2, q3;
        if (LN || cur < top)
        {
            q0 = lnodes[4 * cur + 0], q1 = lnodes[4 * cur + 1], q2 = lnodes[4 * cur + 2], q3 = lnodes[4 * cur + 3];
            if (!LN)
                asm volatile("" : "+v"(q0.x), "+v"(q1.x), "+v"(q2.x), "+v"(q3.x));
        }
        else
            q0 = nodes[4 * (size_t)cur + 0], q1 = nodes[4 * (size_t)cur + 1], q2 = nodes[4 * (size_t)cur + 2],
            q3 = nodes[4 * (size_t)cur + 3];
        float tn0, tn1;
        // The pad.  Table in LDS (LN): A R2 |1/d| of the whole tree, computed once per call (`pa_ray`; the nodes' K are part of
        // their half extents, r1_bvh.cpp).  Table in global memory: per node, pad = A dist2 + K with dist2 = R2 or — wave-uniform,
        // scenes of small spheres — |m0 + m1 - 2 o|^2; such trees always run through these kernels (r1_capi.cpp enqueue_frame).
        V3 pa = pa_ray;
        if (!LN)
        {
            float dist2 = r2;
            if (S.bvh_pad_local)
            {
                const float sx = __fmaf_rn(-2.0f, o.x, q0.x + q0.y), sy = __fmaf_rn(-2.0f, o.y, q0.z + q0.w), sz = __fmaf_rn(-2.0f, o.z, q1.x + q1.y);
                dist2 = __fmaf_rn(sz, sz, __fmaf_rn(sy, sy, sx * sx));
            }
            const float pad = __fmaf_rn(q3.x, dist2, q3.y);
            pa = mk(pad * ainv.x, pad * ainv.y, pad * ainv.z);
        }
        const bool h0 = bvh_box(q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, pa, oi, inv, ainv, best, tn0);
        const bool h1 = bvh_box(q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, pa, oi, inv, ainv, best, tn1);
        const uint32_t c0 = __float_as_uint(q3.z), c1 = __float_as_uint(q3.w);
        if (h0 && h1)
        {
            const bool swap = tn1 < tn0;
            trav_put(trav, sp * R1_BLOCK + tid, swap ? c0 : c1);
            ++sp;
            cur = swap ? c1 : c0;
        }
        else if (h0)
            cur = c0;
        else if (h1)
            cur = c1;
        else if (sp > 0)
            cur = trav_get(trav, --sp * R1_BLOCK + tid);
        else
            cur = R1_BVH_DONE;
    };
    // The root step outside the loops (r1_bvh.cpp: the root of the reference's scenes is [a leaf of <= 2 pairs that every ray tests | the
    // rest], S.bvh_root_leaf): the lanes that start a walk in this call (cur == 0: no child reference points at the root) test that leaf
    // and then the box of the other child, all of them together and in straight-line code, and go on at the other child.  Same offers, same
    // pruning rule as a visit of node 0 followed by the leaf; one node trip and one leaf trip fewer per ray in the divergent loops below.
    // (the small-scene kernels find the code in the K slot of their LDS copy of node 0 — the trace kernel puts it there, K itself is folded
    //  into the half extents for their trees — instead of holding a scalar register for it through the whole kernel)
    const uint32_t root_leaf = LN ? __float_as_uint(lnodes[3].y) : S.bvh_root_leaf;
    if (root_leaf != 0u && cur == 0u) // (the first condition is wave-uniform)
    {
        const int k = root_leaf == 1u ? 1 : 0; // column of the OTHER child in the node's rows
        const float *nf = (const float *)lnodes; // node 0: always in the workgroup's LDS copy (big scenes keep the top of the tree there: at least node 0, r1_capi.cpp)
        const uint32_t leaf = __float_as_uint(nf[14 + (1 - k)]);
        uint32_t other = __float_as_uint(nf[14 + k]);
        const uint32_t lp = (leaf >> COUNT_SHIFT) & 7u;
        if (STATS)
        {
            wstat[5] += (unsigned long long)lp, wstat[16] += 1ull, wstat[17] += 1ull; // ([17]: root steps = one box test each, stats slot 15)
            if ((tid & 63) == __ffsll((long long)__ballot(1)) - 1)
                wstat[3] += 1;
        }
        leaf_quad(prims, ids, leaf & INDEX_MASK, lp, o, d, best, best_id);
        // (the other child's box is fetched only now: six more live registers across the leaf test spill in the 7-wave builds)
        V3 pa = pa_ray;
        if (!LN)
        {
            float dist2 = r2;
            if (S.bvh_pad_local)
            {
                const float sx = __fmaf_rn(-2.0f, o.x, nf[0] + nf[1]), sy = __fmaf_rn(-2.0f, o.y, nf[2] + nf[3]), sz = __fmaf_rn(-2.0f, o.z, nf[4] + nf[5]);
                dist2 = __fmaf_rn(sz, sz, __fmaf_rn(sy, sy, sx * sx));
            }
            const float pad = __fmaf_rn(nf[12], dist2, nf[13]);
            pa = mk(pad * ainv.x, pad * ainv.y, pad * ainv.z);
        }
        float tn;
        const bool h = bvh_box(nf[0 + k], nf[2 + k], nf[4 + k], nf[6 + k], nf[8 + k], nf[10 + k], pa, oi, inv, ainv, best, tn);
        if (ENTRY && entry_idx != 0xFFFFFFFFu && h)
        {
            if (LN && entry_lds != nullptr) // (wave-uniform)
            {
                other = entry_lds[entry_idx];
                other = other == 0xFFFFu ? R1_BVH_DONE : other;
            }
            else
                other = ((const r1_gu32 *)entry_tab)[entry_idx];
        }
        cur = h ? other : R1_BVH_DONE;
        // (the sibling's own visit taken into this step as well — `if (cur < LEAF_BIT) visit_node();` here — measured 35.5 against 36.2
        //  Grays/s: a generic visit runs at 0.54 lane utilisation inside the loop and at the ~0.34 of the starting lanes out here)
    }
#if R1_BVH4
    if (LN && cur == 0u)
        cur = 1u; // a tree without the root step: slot 0 of the 4-wide table is the binary root for the root step's use, the collapsed root is node 1
#endif
    for (;;)
    {
        const unsigned long long walking = __ballot(cur != R1_BVH_DONE);
        if (walking == 0ull)
            break;
        if (CARRY && R1_CARRY_DIV * (uint32_t)__popcll(walking) <= n_alive)
            break; // (n_alive <= 3: never true while a lane walks, so the last walks of a wave run to their end)
        // inner nodes: descend to the nearer child, remember the farther one
        while (cur < LEAF_BIT) // (R1_BVH_DONE has the leaf bit set in either form: one compare)
            visit_node();
        if (cur != R1_BVH_DONE)
        {
            // leaf: `cnt` PAIRS of spheres {cx_a cx_b cy_a cy_b} {cz_a cz_b rsq_a rsq_b}; an odd
            // sphere's partner has radius_sq = -inf (discriminant -inf: never offers a hit)
            const uint32_t first = cur & INDEX_MASK, cnt = (cur >> COUNT_SHIFT) & 7u;
            for (uint32_t j = 0; j < cnt; j += 2u)
            {
                const uint32_t take = cnt - j < 2u ? 1u : 2u;
                if (STATS)
                {
                    wstat[5] += (unsigned long long)take; // sphere pairs tested by this lane
                    wstat[16] += 1ull;                    // leaf trips of this lane
                    if ((tid & 63) == __ffsll((long long)__ballot(1)) - 1)
                        wstat[3] += 1;
                }
                leaf_quad(prims, ids, first + j, take, o, d, best, best_id);
            }
            cur = sp > 0 ? trav_get(trav, --sp * R1_BLOCK + tid) : R1_BVH_DONE;
        }
    }
    tv.cur = cur, tv.sp = sp, tv.best = best, tv.best_id = best_id;
}

// one complete walk (the wavefront kernels)
template <bool STATS>
__device__ __forceinline__ void sweep_bvh(const R1DeviceScene &S, const bool alive, const V3 o, const V3 d, float &t_max, int &hit_index,
                                          uint32_t *trav, const int tid, unsigned long long *wstat)
{
    Trav tv;
    trav_start(tv);
    if (!alive)
        tv.cur = R1_BVH_DONE;
    bvh_advance<STATS, false, false, uint32_t>(S, o, d, tv, trav, tid, 64u, wstat, nullptr);
    if (tv.best_id != 0xFFFFFFFFu)
    {
        t_max = tv.best;
        hit_index = (int)tv.best_id;
    }
}

// ---- the tail of a frame: a few live paths per wave ------------------------------------------------
// Every live ray of the wave in turn against ALL active spheres, lane l testing spheres l, l + 64, ...
// with the reference's per-sphere arithmetic (exact_offer), then the minimum offer, ties to the lowest
// index (the reference's in-order rule, see exact_offer).  Same result as the other sweeps; meant for
// waves that have only a handful of paths left (R1TraceArgs::coop_lanes): one such step costs the wave
// ~n_active / 64 sphere tests per ray and a single round of coalesced loads instead of a walk down the
// tree made of dependent fetches.  Called by all 64 lanes.
__device__ __forceinline__ void cooperative_sweep(const R1DeviceScene &S, unsigned long long live, const V3 o, const V3 d, float &t_max,
                                                  int &hit_index, const int lane)
{
    const f4 *__restrict__ tab = (const f4 *)S.exact;
    while (live) // wave-uniform
    {
        const int src = __ffsll((long long)live) - 1;
        live &= live - 1ull;
        const V3 ro = mk(__shfl(o.x, src, 64), __shfl(o.y, src, 64), __shfl(o.z, src, 64));
        const V3 rd = mk(__shfl(d.x, src, 64), __shfl(d.y, src, 64), __shfl(d.z, src, 64));
        unsigned long long key = ~0ull;
        for (uint32_t i = (uint32_t)lane; i < S.n_active; i += 64u)
        {
            const float t = exact_offer(tab[i], ro, rd);
            if (t < FLT_MAX)
            {
                const unsigned long long k = ((unsigned long long)__float_as_uint(t) << 32) | i; // t > 0: bit order = value order
                key = k < key ? k : key;
            }
        }
        // the reference flags ~0.5 % of the spheres per ray: walk the few lanes that hold an offer
        unsigned long long best = ~0ull;
        unsigned long long offers = __ballot(key != ~0ull);
        while (offers)
        {
            const int l = __ffsll((long long)offers) - 1;
            offers &= offers - 1ull;
            const unsigned long long k = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(key >> 32), l) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, l);
            best = k < best ? k : best;
        }
        if (lane == src && best != ~0ull)
        {
            t_max = __uint_as_float((uint32_t)(best >> 32));
            hit_index = (int)(uint32_t)best;
        }
    }
}

// Attenuation stack.  Small scenes: packed in LDS, three 10-bit sphere indices per word.  Big
// scenes (> 1023 active spheres): one u32 per entry in a global workspace laid out
// [entry][global thread] (coalesced); its traffic is nothing next to a 100 k-sphere sweep.
template <bool BIG, int LW = R1_STACK_WORDS>
__device__ __forceinline__ void stack_push(uint32_t *stack, uint32_t *gstack, uint32_t gstride, uint32_t gtid, int tid, int sp, uint32_t idx)
{
    if (BIG)
    {
        gstack[(size_t)sp * gstride + gtid] = idx;
        return;
    }
    if (LW < R1_STACK_WORDS && sp >= 3 * LW) // deep entries (rare): global workspace, [entry - 3 LW][global thread]
    {
        // (the lane's part of the address is formed HERE: as a loop-invariant 64-bit value it is hoisted out of the tracing loop and spilled to scratch)
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const uint32_t g = (gtid & ~(uint32_t)(R1_BLOCK - 1)) | (uint32_t)t_; // (= gtid, rebuilt from the workgroup's wave-uniform part)
        ((r1_gu32 *)gstack)[(uint32_t)(sp - 3 * LW) * gstride + g] = idx;
        return;
    }
    const int w = sp / 3, sh = (sp - 3 * w) * 10;
    uint32_t v = stack[w * R1_BLOCK + tid];
    v = (v & ~(0x3FFu << sh)) | (idx << sh);
    stack[w * R1_BLOCK + tid] = v;
}
template <bool BIG>
__device__ __forceinline__ uint32_t stack_get(const uint32_t *stack, const uint32_t *gstack, uint32_t gstride, uint32_t gtid, int tid, int e)
{
    if (BIG)
        return gstack[(size_t)e * gstride + gtid];
    const int w = e / 3, sh = (e - 3 * w) * 10;
    return (stack[w * R1_BLOCK + tid] >> sh) & 0x3FFu;
}

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const R1FastDiv dv)
{
    return dv.pow2 ? (n >> dv.shift) : (__umulhi(n, dv.mul) >> dv.shift);
}

// seeds + primary ray of sample s of pixel (x, y) (rayweek1.cpp:759-760)
__device__ __forceinline__ void start_ray(const R1TraceArgs &A, Path &p, const int x, const int y, const uint32_t s, const uint32_t seed)
{
    const r1_sample_seed sd = r1_seed_sample(seed, (uint32_t)(y * A.width + x), s);
    p.s_scalar = sd.scalar;
    p.s0 = sd.lane0;
    p.s1 = sd.lane1;
    p.s2 = sd.lane2;
    p.depth = 0;
    p.sp = 0;

    // uv = (myrand01_x4(state4) + (x, y)) * (1/W, 1/H): lanes 0,1 used, lane 2 advances too
    const float j0 = rand01(p.s0), j1 = rand01(p.s1);
    (void)xorshift32(p.s2);
    const float u = (j0 + (float)x) * A.inv_w;
    const float v = (j1 + (float)y) * A.inv_h;

    // random_in_unit_disk (rayweek1.cpp:353-362): g++ argument order => y gets the first draw
    V3 dk;
    do
    {
        const float first = rand02_minus1(p.s_scalar);
        const float second = rand02_minus1(p.s_scalar);
        dk = mk(second, first, 0.0f); // 2 * Vec3(rand, rand, 0) - Vec3(1, 1, 0): z = 0 - 0
    } while (vdot(dk, dk) >= 1.0f);

    // Camera::getRay (rayweek1.cpp:381-386)
    const V3 rd = vscale(dk, A.cam.lens_radius);
    const V3 offset = vadd(vscale(ld3(A.cam.u), rd.x), vscale(ld3(A.cam.v), rd.y));
    const V3 org = ld3(A.cam.origin);
    p.o = vadd(org, offset);
    const V3 dir =
        vsub(vsub(vadd(vadd(ld3(A.cam.lower_left), vscale(ld3(A.cam.horizontal), u)), vscale(ld3(A.cam.vertical), v)), org), offset);
    p.d = vunit(dir);
}

// (local tile j, pixel `pix` of the padded tile) -> image coordinates; false outside the image (edge tiles)
__device__ __forceinline__ bool tile_pixel(const R1TraceArgs &A, const uint32_t j, const uint32_t pix, int &x, int &y)
{
    const uint32_t ly = fastdiv(pix, A.div_tw);
    const uint32_t lx = pix - ly * (uint32_t)A.tile_w;
    const uint32_t tile = (uint32_t)A.shard + j * (uint32_t)A.num_shards;
    const uint32_t ty = fastdiv(tile, A.div_tx);
    const uint32_t tx = tile - ty * (uint32_t)A.tiles_x;
    x = (int)(tx * (uint32_t)A.tile_w + lx);
    y = (int)(ty * (uint32_t)A.tile_h + ly);
    return x < A.width && y < A.height;
}

// sample slot k -> (tile, pixel, sample); then seeds + primary ray.
// Returns false for a void slot (pixel of an edge tile that lies outside the image).
template <bool BATCH = false>
__device__ __forceinline__ bool start_sample(const R1TraceArgs &A, Path &p, uint32_t k)
{
    const uint32_t j = fastdiv(k, A.div_full); // padded tile, frame-major over the frames of the launch
    const uint32_t r = k - j * A.full;
    const uint32_t pix = fastdiv(r, A.div_spp);
    const uint32_t s = r - pix * (uint32_t)A.spp;
    uint32_t jl = j, seed = A.seed;
    if (BATCH) // (its own build of the kernel, MODE 3: the single-frame kernels stay as they were, to the register)
    {
        typedef const uint32_t __attribute__((address_space(4))) *cu32_ptr;
        const cu32_ptr b = (cu32_ptr)A.batch; // {n_frames, seed_stride, div_tiles{mul, shift, pow2}, n_local_tiles}
        R1FastDiv dv;
        dv.mul = b[2], dv.shift = b[3], dv.pow2 = b[4];
        const uint32_t f = fastdiv(j, dv);
        jl = j - f * b[5];
        seed += f * b[1];
    }
    int x, y;
    if (!tile_pixel(A, jl, pix, x, y))
        return false;
    // output slot: samples are stored [tile][sample][pixel] so that the resolve pass reads
    // coalesced (consecutive pixels of one sample index)
    p.k = (j * (uint32_t)A.spp + s) * (uint32_t)(A.tile_w * A.tile_h) + pix;
    start_ray(A, p, x, y, s, seed);
    return true;
}

// PIXEL mode (frames in flight): a lane owns a PIXEL and runs its spp samples one after the other, so
// `col += color(...)` (rayweek1.cpp:762) happens in a register in the reference's order and the resolved
// pixel (rayweek1.cpp:765-775) is the only thing the frame writes: no per-sample records, no second kernel.
struct Pixel
{
    V3 acc;       // col of render_tile
    uint32_t xy;  // x | y << 16
    uint32_t off; // output pixel index (dense tile block: the queue slot itself; row-major image: y * width + x)
    uint32_t s;   // next sample; == spp: no pixel in hand
};

// pixel slot kp of the frame's queue = (local tile, pixel of the padded tile) -> Pixel; false for a void slot
__device__ __forceinline__ bool pixel_take(const R1TraceArgs &A, Pixel &px, const uint32_t kp)
{
    const uint32_t j = fastdiv(kp, A.div_full); // PIXEL mode: `full` = tile_w * tile_h
    int x, y;
    if (!tile_pixel(A, j, kp - j * A.full, x, y))
        return false;
    px.acc = mk(0, 0, 0);
    px.xy = (uint32_t)x | ((uint32_t)y << 16);
    px.off = A.block_layout ? kp : (uint32_t)(y * A.width + x);
    px.s = 0;
    return true;
}

// rayweek1.cpp:765-775: col *= 1 / spp; sqrtf per channel; (uint8)(int)(c * 255.99f)
__device__ __forceinline__ void pixel_write(const R1TraceArgs &A, const Pixel &px)
{
    uint8_t *o = (uint8_t *)A.samples + 3 * (size_t)px.off; // PIXEL mode: `samples` is the output image / tile block
    o[0] = (uint8_t)(int)(ieee_sqrt(px.acc.x * A.inv_spp) * 255.99f);
    o[1] = (uint8_t)(int)(ieee_sqrt(px.acc.y * A.inv_spp) * 255.99f);
    o[2] = (uint8_t)(int)(ieee_sqrt(px.acc.z * A.inv_spp) * 255.99f);
}

// Path state in memory, [3][n] float4: {ox oy oz dx} {dy dz s_scalar s0} {s1 s2 k rays|depth<<8|sp<<16}
// (0xFFFFFFFF in the last word marks a void sample slot).  Used by the wavefront variant.
#define R1_PATH_VOID 0xFFFFFFFFu
__device__ __forceinline__ void path_store(float4 *paths, const uint32_t n, const uint32_t slot, const Path &p)
{
    paths[slot] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
    paths[(size_t)n + slot] = make_float4(p.d.y, p.d.z, __uint_as_float(p.s_scalar), __uint_as_float(p.s0));
    paths[2 * (size_t)n + slot] = make_float4(__uint_as_float(p.s1), __uint_as_float(p.s2), __uint_as_float(p.k),
                                              __uint_as_float(((uint32_t)p.depth << 8) | ((uint32_t)p.sp << 16)));
}

// returns false for a void slot
__device__ __forceinline__ bool path_load(const float4 *paths, const uint32_t n, const uint32_t slot, Path &p)
{
    const float4 a = paths[slot], b = paths[(size_t)n + slot], c = paths[2 * (size_t)n + slot];
    p.o = mk(a.x, a.y, a.z);
    p.d = mk(a.w, b.x, b.y);
    p.s_scalar = __float_as_uint(b.z), p.s0 = __float_as_uint(b.w);
    p.s1 = __float_as_uint(c.x), p.s2 = __float_as_uint(c.y);
    p.k = __float_as_uint(c.z);
    const uint32_t packed = __float_as_uint(c.w);
    p.depth = (int)((packed >> 8) & 255u), p.sp = (int)((packed >> 16) & 255u);
    return packed != R1_PATH_VOID;
}

// ---- one color() level after the hit test (rayweek1.cpp:517-534): count the ray, then either
// scatter (new ray in p, hit index pushed on the attenuation stack) or finish the path: sky
// colour times the stacked attenuations, or black.  Returns true when the path has ended, with
// its radiance in `col`.  Shared by the megakernel and the wavefront kernels.
template <bool BIG, int LW = R1_STACK_WORDS>
__device__ __forceinline__ bool shade_level(const R1TraceArgs &A, Path &p, const int hit, const float t_hit, uint32_t *s_stack,
                                            const uint32_t gstride, const uint32_t gtid, const int tid, V3 &col)
{
    bool done = false; // (this level's ray, rayweek1.cpp:517, is counted through the depth: path_rays)
    col = mk(0, 0, 0);
    if (hit >= 0)
    {
        if (p.depth < A.max_bounces)
        {
            // hit record (rayweek1.cpp:316-322)
            const f4 e = ((const f4 *)A.scene.exact)[hit];
            const f4 sh = ((const f4 *)A.scene.shade)[hit];
            const f4 mt = ((const f4 *)A.scene.mat)[hit];
            const V3 hp = vadd(p.o, vscale(p.d, t_hit));
            const V3 n = vscale(vsub(hp, mk(e.x, e.y, e.z)), sh.x);
            const uint32_t type = __float_as_uint(mt.x);
            // Lambertian and Metal both draw random_in_unit_sphere from the x4 stream
            // (rayweek1.cpp:405, :430; Metal even with fuzz == 0): one shared loop
            V3 rius = mk(0, 0, 0);
            if (type != 2u)
                rius = random_in_unit_sphere(p);
            V3 dir; // un-normalised scattered direction; Ray::Ray normalises (rayweek1.cpp:107)
            if (type == 0u)
            {
                // Lambertian::scatter rayweek1.cpp:403-409
                const V3 target = vadd(vadd(hp, n), rius);
                dir = vsub(target, hp);
            }
            else if (type == 1u)
            {
                // Metal::scatter rayweek1.cpp:427-433; reflect :414-417
                const V3 refl = vsub(p.d, vscale(n, 2.0f * vdot(p.d, n)));
                dir = vadd(refl, vscale(rius, mt.y));
            }
            else
            {
                // Dielectric::scatter rayweek1.cpp:470-511 (attenuation 1: nothing to push)
                const float ref_idx = mt.y;
                const float ddn = vdot(p.d, n);
                const V3 reflected = vsub(p.d, vscale(n, 2.0f * ddn));
                V3 outward;
                float ni_over_nt, cosine;
                if (ddn > 0)
                {
                    outward = vneg(n);
                    ni_over_nt = ref_idx;
                    cosine = ref_idx * ddn;
                }
                else
                {
                    outward = n;
                    ni_over_nt = mt.z; // 1.0f / _refIdx, divided on the host (same IEEE division)
                    cosine = -ddn;
                }
                // refract rayweek1.cpp:439-452
                const float dt = vdot(p.d, outward);
                const float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
                float reflect_prob = 1.0f;
                V3 refracted = mk(0, 0, 0);
                if (discriminant > 0)
                {
                    refracted = vsub(vscale(vsub(p.d, vscale(outward, dt)), ni_over_nt), vscale(outward, ieee_sqrt(discriminant)));
                    // schlick rayweek1.cpp:454-459; r0 = ((1 - ref)/(1 + ref))^2 comes from the host
                    const float r0 = mt.w;
                    reflect_prob = r0 + (1.0f - r0) * pow5(1.0f - cosine);
                }
                dir = (rand01(p.s_scalar) < reflect_prob) ? reflected : refracted;
            }
            const V3 nd = vunit(dir);
            p.o = hp;
            p.d = nd;
            if (type == 2u || type == 0u || vdot(nd, n) > 0)
            {
                if (type != 2u)
                {
                    stack_push<BIG, LW>(s_stack, A.gstack, gstride, gtid, tid, p.sp, (uint32_t)hit);
                    ++p.sp;
                }
            }
            else
                done = true; // Metal::scatter() == false -> Vec3(0,0,0), rayweek1.cpp:432, :528
            if (!done)
                ++p.depth; // (a path that ends here keeps its depth: its rays = depth + 1 whichever way it ends)
        }
        else
            done = true; // depth == MAX_BOUNCES: no scatter, no draws, black (rayweek1.cpp:523-528)
    }
    else
    {
        // miss: sky (rayweek1.cpp:532-534), lerp = (1 - t) * a + t * b (mymath.h:212-216)
        const float t = 0.5f * (p.d.y + 1.0f);
        const float omt = 1.0f - t;
        col = mk(omt * 1.0f + t * 0.5f, omt * 1.0f + t * 0.7f, omt * 1.0f + t * 1.0f);
        // unwind: attenuation * color(...) innermost first (rayweek1.cpp:525)
        if (BIG)
        {
            for (int e = p.sp - 1; e >= 0; --e)
            {
                const float4 sh = A.scene.shade[stack_get<BIG>(s_stack, A.gstack, gstride, gtid, tid, e)];
                col = mk(sh.y * col.x, sh.z * col.y, sh.w * col.z);
            }
        }
        else if (p.sp > 0)
        {
            int top = p.sp;
            if (LW < R1_STACK_WORDS)
            {
                int t2 = tid;
                asm volatile("" : "+v"(t2)); // (as in stack_push)
                const uint32_t g = (gtid & ~(uint32_t)(R1_BLOCK - 1)) | (uint32_t)t2;
                for (int e = top - 1; e >= 3 * LW; --e) // the deep entries first (innermost attenuation first)
                {
                    const float4 sh = A.scene.shade[((const r1_gu32 *)A.gstack)[(uint32_t)(e - 3 * LW) * gstride + g]];
                    col = mk(sh.y * col.x, sh.z * col.y, sh.w * col.z);
                }
                top = top < 3 * LW ? top : 3 * LW;
            }
            // packed stack: one LDS word holds entries 3w, 3w+1, 3w+2 (10 bits each); walk it word by word from the top
            // entry down.  The three albedos of a word are fetched together, whether the word is full or not (any 10-bit
            // index is inside the table, r1_capi.cpp), and the slots above the top entry multiply by 1.0f, which changes
            // no bit: one round trip to the table per word instead of one per entry.
            int w = (top - 1) / 3, j = (top - 1) - 3 * w;
            int t_ = tid;
            asm volatile("" : "+v"(t_)); // (the lane's LDS offset is formed here: hoisted out of the tracing loop, tid * 4 was a spilled register reloaded at every path's end)
            for (; w >= 0; --w, j = 2)
            {
                const uint32_t v = s_stack[w * R1_BLOCK + t_];
                const float4 s2 = A.scene.shade[(v >> 20) & 0x3FFu], s1 = A.scene.shade[(v >> 10) & 0x3FFu], s0 = A.scene.shade[v & 0x3FFu];
                const bool u2 = j >= 2, u1 = j >= 1;
                col = mk((u2 ? s2.y : 1.0f) * col.x, (u2 ? s2.z : 1.0f) * col.y, (u2 ? s2.w : 1.0f) * col.z);
                col = mk((u1 ? s1.y : 1.0f) * col.x, (u1 ? s1.z : 1.0f) * col.y, (u1 ? s1.w : 1.0f) * col.z);
                col = mk(s0.y * col.x, s0.z * col.y, s0.w * col.z);
            }
        }
        done = true;
    }
    return done;
}

// ---- tiles resolved inside the trace kernel (R1_LAND) ------------------------------------------------------
// The reference resolves a pixel where it traced it (rayweek1.cpp:762-775).  Here the samples of a pixel end on different lanes and
// waves, and their fp32 sum must run in sample order, so every sample still leaves a 16-byte record; but the records of a 32 x 32
// tile are summed by the waves of the trace launch itself, and the pixels go straight to where the frame is wanted (device memory or
// the caller's page-locked host buffer): no second launch waiting for workgroup slots behind persistent waves, no copy.
// What makes it cheap is that a tile never leaves its XCD.  The eight XCDs' L2s are not coherent with each other: records traced on
// one XCD and summed on another must be written through to memory (sc1 stores: +7 % on the tracing loop, measured) and waited for
// by workgroups that do nothing else (+14 %): profiles/r04/land_cross_xcd_cost_breakdown.txt, the first form of this code.  So:
//   * a tile belongs to the XCD that CLAIMS it: every XCD has a cursor {tile (or group of 8 tiles), next sample slot} its waves take their
//     chunks from; the wave that takes the last chunk (or finds nothing installed yet) claims the launch's next unclaimed tile(s) for its
//     XCD and installs it; waves that meet a used-up tile meanwhile wait for that (microseconds).  Nothing
//     is assumed about which XCD a workgroup runs on — the dispatcher deals them round-robin, starting wherever the last launch
//     stopped — and XCDs that run faster simply claim more tiles;
//   * the waves of an XCD store the records with plain stores and count the samples they finish per tile — in LDS, for the tiles of the
//     wave's current and previous chunk — and subtract the count from the tile's countdown when the wave moves on to another tile: one
//     fire-and-forget atomic per chunk (an atomic per finished sample group and iteration cost 7 %: the vector-memory counter of this
//     chip is in order, so every iteration's first load waited for the memory-side atomic in front of it; a returning one per chunk 3 %);
//   * a wave that has run out of work looks at the tiles it ever took a chunk from (a short list in its LDS row): a countdown at zero is
//     swapped for a mark, and the wave that wins the swap sums the tile — its records sit in this XCD's L2 or in memory, nowhere else —
//     adds the tile's rays to the frame's count and re-arms the countdown for the next launch.  Of the waves that worked on a tile the
//     last to get there finds it complete, so every tile is summed.  The wave that takes the frame's last tile off publishes the ray
//     count.  (The list: 24 entries in LDS, the rest in a row of global memory long enough for every tile of the launch.)
//   * every record carries the launch's tag in its ray-count word: a countdown is not ordered after the stores it counts, so a wave
//     that finds a record of an earlier launch in a tile it owes simply reads the tile again (past the vector L1: sc0 loads).
// Per launch, behind the queue pointer (one of two sets, zeroed by the launch after): line x = XCD x's cursor (uint64); line 16 = the
// launch's next unclaimed tile.
#ifndef R1_LAND_EXP
#define R1_LAND_EXP 0 // measurements only (make tuning EXTRA=-DR1_LAND_EXP=n; frames are NOT valid): 1 no tile is resolved, 2 and no countdown atomics
#endif
#define R1_LAND_OWED 24u        // tiles of a wave's list kept in LDS (the rest: its row of the spill area in global memory)
#define R1_LAND_LOADS 8      // records of one pixel a lane keeps in flight while it sums a tile (4 registers each; 10 spill in the 72-register builds)
#define R1_LAND_MAX_XCD 8u

__device__ __forceinline__ uint32_t xcc_id() { return (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; } // HW_REG_XCC_ID[3:0]

// What a wave knows about the tiles it works on — kept in LDS (a row of R1_LAND_ROW words per wave), NOT in registers: the tracing loop
// is at its register budget, and every wave-uniform value carried around it cost spilled scalar registers, whose lanes take a
// vector register from the loop (five such values: +6.5 % per frame, profiles/r04/land_xcd_local_steps.txt).
//   [0] t0, [1] n0: tile of the wave's current chunk, samples of it the wave has finished since it last subtracted
//   [2] t1, [3] n1: the same for the previous chunk's tile
//   [4] touched: tiles this wave has taken chunks from; [8 ...]: their indices (the first R1_LAND_OWED of them)
#define R1_LAND_ROW 32u
#define R1_LAND_CLAIMED 0xFFFFFFFFu

// the wave takes a chunk from tile t for the first time (or again): t goes on the list of tiles it looks at before it exits.  ONE lane.
__device__ __forceinline__ void land_note(const R1TraceArgs &A, uint32_t *row, const uint32_t t)
{
    const uint32_t at = row[4];
    row[4] = at + 1u;
    // (past R1_LAND_OWED the wave's row of the spill area, which has room for every tile of the launch: the XCD's cursor only moves
    //  forward, so a wave meets a tile at most once — however unevenly the launch's workgroups get their slots, the list cannot overflow)
    if (at < R1_LAND_OWED)
        row[8u + at] = t;
    else
    {
        // (32-bit index on a scalar base: as a 64-bit per-lane address the row's part of it is hoisted out of the tracing loop and spilled to scratch)
        uint32_t tx = threadIdx.x;
        asm volatile("" : "+v"(tx)); // (formed here, not hoisted)
        const uint32_t wave = blockIdx.x * (R1_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(tx >> 6));
        ((r1_gu32 *)A.land.owed_spill)[(size_t)wave * A.land.spill_stride + at] = t;
    }
}

// subtract n finished samples from tile t's countdown: fire and forget (a RETURNING atomic here — "who reaches zero owes the tile" —
// stalled the wave for a memory-side round trip at every chunk: +3 % per frame).  Who sums the tile is settled when waves exit.
__device__ __forceinline__ void land_flush(const R1TraceArgs &A, const uint32_t t, const uint32_t n)
{
    if (n == 0u || R1_LAND_EXP >= 2)
        return;
    (void)__hip_atomic_fetch_sub(A.land_cnt + t * R1_LAND_CNT_STRIDE, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// this lane has just stored the record of sample slot k: one more finished sample of its tile.  Per lane, no wave-uniform state.
__device__ __forceinline__ void land_count(const R1TraceArgs &A, uint32_t *row, const uint32_t k)
{
    const uint32_t j = fastdiv(k, A.div_full); // tile of the launch: k = (j spp + s) tile_px + pix
    if (j == row[0])
        atomicAdd(&row[1], 1u); // (LDS)
    else if (j == row[2])
        atomicAdd(&row[3], 1u);
    else
        land_flush(A, j, 1u); // (rare: a path that outlived two chunks; its tile is on the wave's list since the chunk was taken)
}

// the wave's next chunk comes from tile t.  ONE lane.
__device__ __forceinline__ void land_chunk(const R1TraceArgs &A, uint32_t *row, const uint32_t t)
{
    const uint32_t t0 = row[0];
    if (t == t0)
        return;
    land_flush(A, row[2], row[3]);
    row[2] = t0, row[3] = row[1];
    row[0] = t, row[1] = 0u;
    land_note(A, row, t);
}

// One tile of the launch (t = frame * n_local_tiles + local tile) by ONE wave, lane l taking pixels l, l + 64, ...: false if a record
// did not carry the launch's tag (nothing is accounted then; pixels written from such a pass are overwritten by the pass that
// succeeds — the host sees the buffer only after the kernel).  Same arithmetic and order as r1_resolve_kernel = rayweek1.cpp:762-775.
template <int LOADS /* records of one pixel a lane keeps in flight: 4 registers each */>
__device__ __forceinline__ bool land_resolve_tile(const R1TraceArgs &A, const uint32_t t, const int lane)
{
    const R1LandArgs &L = A.land;
    const uint32_t f = t / A.n_local_tiles, lt = t - f * A.n_local_tiles;
    const uint32_t tile = (uint32_t)A.shard + lt * (uint32_t)A.num_shards;
    const uint32_t ty = tile / (uint32_t)A.tiles_x, tx = tile - ty * (uint32_t)A.tiles_x;
    const int x0 = (int)tx * A.tile_w, y0 = (int)ty * A.tile_h;
    const int tw = min(A.tile_w, A.width - x0), th = min(A.tile_h, A.height - y0);
    const uint32_t tile_px = (uint32_t)(A.tile_w * A.tile_h);
    // the tile's records [s][pixel], read with sc0 loads (buffer form: 16 bytes at once): they must come from the L2, not from this CU's
    // vector L1 — a pass that meets a record whose store is still on its way would otherwise find the same stale line again
    const __amdgpu_buffer_rsrc_t rec = __builtin_amdgcn_make_buffer_rsrc((void *)(A.samples + (size_t)t * A.full), 0, (int)(A.full * 16u), 0x00020000);
    constexpr int SC0 = 1;
    typedef uint32_t land_u4 __attribute__((ext_vector_type(4)));
    uint8_t *const out = L.out + (size_t)f * L.out_stride;
    const uint32_t spp = (uint32_t)A.spp;
    uint32_t bad = 0;
    unsigned long long rays = 0;
    for (uint32_t pix = (uint32_t)lane; pix < tile_px; pix += 64u)
    {
        const int ly = (int)(pix / (uint32_t)A.tile_w), lx = (int)(pix - (uint32_t)ly * (uint32_t)A.tile_w);
        if (lx >= tw || ly >= th)
            continue; // void slots of an edge tile
        float cr = 0, cg = 0, cb = 0;
        for (uint32_t s0 = 0; s0 < spp; s0 += LOADS)
        {
            const uint32_t n = min((uint32_t)LOADS, spp - s0); // (wave-uniform)
            land_u4 v[LOADS];
#pragma unroll
            for (uint32_t u = 0; u < LOADS; ++u)
                if (u < n)
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rec, (int)(((s0 + u) * tile_px + pix) * 16u), 0, SC0);
#pragma unroll
            for (uint32_t u = 0; u < LOADS; ++u)
                if (u < n)
                {
                    cr += __uint_as_float(v[u].x), cg += __uint_as_float(v[u].y), cb += __uint_as_float(v[u].z); // col += color(...) rayweek1.cpp:762
                    bad |= (v[u].w & ~255u) ^ A.land_tag;
                    rays += v[u].w & 255u;
                }
        }
        cr *= L.inv_spp, cg *= L.inv_spp, cb *= L.inv_spp;
        cr = ieee_sqrt(cr), cg = ieee_sqrt(cg), cb = ieee_sqrt(cb);
        const size_t o = L.block_layout ? ((size_t)lt * tile_px + pix) * 3 : ((size_t)(y0 + ly) * A.width + (x0 + lx)) * 3;
        out[o + 0] = (uint8_t)(int)(cr * 255.99f);
        out[o + 1] = (uint8_t)(int)(cg * 255.99f);
        out[o + 2] = (uint8_t)(int)(cb * 255.99f);
    }
    if (__ballot(bad != 0u))
        return false;
    for (int off = 32; off > 0; off >>= 1)
        rays += __shfl_down(rays, off, 64);
    if (lane == 0)
    {
        __hip_atomic_store(A.land_cnt + t * R1_LAND_CNT_STRIDE, (uint32_t)(tw * th) * spp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed for the next launch
        // the tile's rays, then one tile less to go: the second atomic is issued only when the first has been performed, so whoever
        // takes the frame's last tile off finds every tile's rays added — and publishes the count (rayweek1.cpp:809-813)
        const unsigned long long before = __hip_atomic_fetch_add(L.frame_rays + f, rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(before));
        const uint32_t left = __hip_atomic_fetch_sub(L.frame_left + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == 1u)
        {
            const unsigned long long total = __hip_atomic_exchange(L.frame_rays + f, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *(L.rays_in_out ? (unsigned long long *)(out + L.rays_offset) : L.rays_dst) = total;
            __hip_atomic_store(L.frame_left + f, A.n_local_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // as the next launch expects it
        }
    }
    return true;
}

// A tracing wave has run out of work: it subtracts what it still holds, then looks at every tile it ever took a chunk from.  A countdown
// at zero means every sample of the tile has been counted; the wave that swaps the zero for the CLAIMED mark sums the tile.  Every tile
// finds its wave: of the waves that worked on a tile, the last to get here has seen all the others subtract (they waited for their
// atomics before they looked), so it reads zero unless another wave already has the tile.
template <int LOADS>
__device__ __forceinline__ void land_exit(const R1TraceArgs &A, uint32_t *row, const int lane)
{
    if (R1_LAND_EXP)
        return;
#ifdef R1_LAND_EXIT_PRIO
    __builtin_amdgcn_s_setprio(R1_LAND_EXIT_PRIO); // (a wave that sums tiles holds its workgroup's slot: let it finish first)
#endif
    if (lane == 0)
    {
        land_flush(A, row[2], row[3]);
        land_flush(A, row[0], row[1]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this wave's record stores, subtractions and spilled list entries have been performed
    __builtin_amdgcn_wave_barrier();
    const uint32_t touched = (uint32_t)__builtin_amdgcn_readfirstlane((int)row[4]);
    for (uint32_t base = 0; base < touched; base += 64u) // (wave-uniform; one trip unless the list spilled)
    {
        const uint32_t i = base + (uint32_t)lane;
        uint32_t t = 0u, c = 1u;
        if (i < touched)
        {
            if (i < R1_LAND_OWED)
            {
                t = row[8u + i];
                asm volatile("" : "+v"(t)); // (two loads kept apart: a select between an LDS and a global address becomes a generic load)
            }
            else
            {
                const uint32_t wave = blockIdx.x * (R1_BLOCK / 64) + (threadIdx.x >> 6);
                t = ((const r1_gu32 *)A.land.owed_spill)[(size_t)wave * A.land.spill_stride + i];
            }
            // (device scope: the countdowns are only ever touched by atomics, which are performed behind the L2s)
            c = __hip_atomic_load(A.land_cnt + t * R1_LAND_CNT_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c == 0u) // every lane that reads zero tries for its tile; one wave of the XCD wins it
            {
                uint32_t seen = 0u;
                if (!__hip_atomic_compare_exchange_strong(A.land_cnt + t * R1_LAND_CNT_STRIDE, &seen, R1_LAND_CLAIMED, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT))
                    c = 1u;
            }
        }
        unsigned long long mine = __ballot(c == 0u);
        while (mine)
        {
            const int b = __ffsll((long long)mine) - 1;
            mine &= mine - 1ull;
            const uint32_t tt = (uint32_t)__builtin_amdgcn_readlane((int)t, b);
            // a record whose store is still on its way (the countdown is not ordered after the stores) shows an old tag: read again
            uint32_t tries = 0;
            while (!land_resolve_tile<LOADS>(A, tt, lane))
                if (++tries == R1_LAND_MAX_WAIT)
                {
                    if (lane == 0 && A.land.error)
                        *A.land.error = 1u; // never in a correct run; the host reports the launch as failed
                    break;
                }
        }
    }
}

} // namespace

// ============================================================================================
// The trace kernel.  Persistent: grid = CUs x blocks/CU, every wave loops until the global
// sample queue is empty and its own lanes have finished their paths.
// ============================================================================================
// STATS = diagnostic build (variant R1_VARIANT_STATS): same results, plus per-phase cycle and
// utilisation counters in A.stats; never used by the product path.
// Waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument): the
// product kernels are sized by their LDS (tree: 6 workgroups per CU with a 128-node table — 10 KB attenuation stack + 8 KB
// traversal stack + 8 KB nodes; exhaustive sweep: 5), so their VGPR count has to stay under 512 / 6 -> 80 and 512 / 5 -> 96.
// The big-scene tree kernel waits for node fetches from L2, not for the VALU: it is built for 8 waves per SIMD (<= 64 VGPRs,
// which it meets without the spare sample, and <= 96 SGPRs — at its natural 106 the 800 SGPRs of a SIMD hold seven waves):
// 13.1 -> 14.4 Grays/s on 100 004 spheres.
// (round 3: the small-scene tree kernels for frames in flight, MODE 0 / 3, run SEVEN workgroups per CU — their LDS footprint is 22 KB with
// R1_STACK_LDS_WORDS_TP — so they are built for 7 waves per SIMD: <= 73 VGPRs, which the kernel meets at 69, and <= 96 SGPRs)
template <int VARIANT, bool STATS, bool BIG, int MODE>
struct TraceWaves
{
#ifndef R1_TREE_WAVES_TP
#define R1_TREE_WAVES_TP 7
#endif
#ifndef R1_TREE_WAVES_LAT
#define R1_TREE_WAVES_LAT 6 // (the synchronous build: 67 VGPRs, seven workgroups per CU at run time; built for eight — 64 VGPRs, two words of the
                            //  LDS stack less — it spills and is no faster: profiles/r04/retune_after_fresh_args.txt)
#endif
    static constexpr int value = STATS ? 1 : (VARIANT == 4 ? (BIG ? 8 : ((MODE == 0 || MODE == 3) ? R1_TREE_WAVES_TP : R1_TREE_WAVES_LAT)) : (VARIANT == 2 && !BIG ? 5 : 1));
};

// MODE 1 = LAT = latency-mode build (the synchronous entry points: one frame, full grid): sub-queues and
// the cooperative tail are compiled in.  The throughput-mode build (frames in flight, few
// long-lived waves per frame) leaves them out: they cost it registers and bring it nothing.
// MODE 0 = frames in flight (the throughput entry point): samples in one guided queue, few long-lived waves per frame.
// MODE 2 = PIXEL mode (the throughput entry point after r1_set_pixel_mode; see struct Pixel): the queue holds pixels.
template <int VARIANT, bool STATS, bool BIG, int MODE>
__global__ void __launch_bounds__(R1_BLOCK, (TraceWaves<VARIANT, STATS, BIG, MODE>::value)) r1_trace_kernel(const R1TraceArgs A)
{
    constexpr bool LAT = MODE == 1, PIX = MODE == 2, BATCH = MODE == 3; // MODE 3 = MODE 0 whose queue spans the frames of a batch
    // tiles resolved inside the kernel: the throughput builds of the tree kernels (frames in flight, MODE 0 / 3); a
    // launch through them is a landing launch (r1_launch_trace checks).  The synchronous frame keeps the resolve launch (measured
    // slower with its tiles summed at wave exit, R1_LAND_SYNC), and so do the exhaustive sweep's kernels.
    constexpr bool LAND = R1_LAND_MODE(MODE) && !STATS && VARIANT == 4; // (the exhaustive sweep keeps the resolve launch: its loop pays 14 % for the bookkeeping, 16.6 against 19.2 Grays/s)
    const uint32_t tb = blockIdx.x, n_tb = gridDim.x;
    // LAND: this workgroup's XCD (HW_REG_XCC_ID), the tiles of the launch
    const uint32_t xcd = LAND ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(xcc_id() & 7u)) : 0u;
    const uint32_t land_tiles = LAND ? A.total_samples / A.full : 0u;
#ifdef R1_LAND_DEBUG
    if (LAND && threadIdx.x == 0 && blockIdx.x < 8)
        A.queue[32u * (17u + blockIdx.x % 7u)] = 0x1000u | (xcc_id() << 4) | xcd; // (tools/land_debug.py)
#endif
    if (LAND && blockIdx.x == 0) // the cursors / wave counts the launch BEFORE this one used: nobody touches them during this launch
        for (uint32_t i = threadIdx.x; i < 32u * A.land.clear_count; i += R1_BLOCK)
            A.land.clear_heads[i] = 0u;
    typedef typename IdxType<BIG>::type IDX;
    unsigned long long wstat[18];
    if (STATS)
    {
        for (int i = 0; i < 18; ++i)
            wstat[i] = 0;
        wstat[14] = __builtin_readcyclecounter();
    }
    // STATS builds: optional per-wave log {start, queue-empty, end (100 MHz real-time clock), iterations};
    // its address is handed over in stats[15] (tools/wave_timeline.py)
    unsigned long long *wave_log = nullptr;
    unsigned long long log_start = 0, log_exhausted = 0;
    if (STATS && A.stats)
    {
        wave_log = (unsigned long long *)A.stats[16]; // (the word behind the sixteen published slots)
        log_start = __builtin_amdgcn_s_memrealtime();
    }
    // words of the attenuation stack in LDS (the rest of a deep path's entries live in the global workspace)
    constexpr int LW = VARIANT == 4 ? (((MODE == 0 || MODE == 3) && !STATS) ? R1_STACK_LDS_WORDS_TP : R1_STACK_LDS_WORDS) : R1_STACK_WORDS;
    __shared__ uint32_t s_stack[BIG ? 1 : LW * R1_BLOCK];
    __shared__ uint32_t s_cand[VARIANT == 2 ? (BIG ? R1_CAND_CAP : R1_BIT_WORDS) * R1_BLOCK : 1];
    __shared__ IDX s_pairs[VARIANT == 2 ? (R1_BLOCK / 64) * PairBits<IDX>::cap : 1];
    // R1_VARIANT_BVH: traversal stack [tree depth][thread], sized at launch (dynamic LDS)
    extern __shared__ uint32_t s_trav[];
    const uint32_t gstride = n_tb * R1_BLOCK, gtid = tb * R1_BLOCK + threadIdx.x;
    __shared__ unsigned long long s_best[VARIANT == 2 ? R1_BLOCK : 1];
    __shared__ f4 s_tile[BIG && VARIANT == 2 ? 2 * R1_TILE_F4 : 1];

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;

    // Tree kernels, small scenes (!BIG: at most R1_NODES_LDS_MAX nodes, r1_capi.cpp): the workgroup keeps its own copy of
    // the node table in LDS, behind the traversal stack.  A node visit is four dependent 16-byte loads per lane; LDS
    // answers sooner than the vector L1, and the table no longer competes with the sphere tables for its 32 KB:
    // 26.9 -> 29.3 Grays/s, one synchronous frame 1.34 -> 1.17 ms (large scene, 128 nodes = 8 KB).
    constexpr bool LN = VARIANT == 4 && !BIG;
    typedef typename IdxType<!LN>::type TS; // traversal-stack entry: uint16_t for the small-scene tree kernels, else uint32_t
    const size_t trav_words = VARIANT == 4 ? (size_t)A.bvh_depth * R1_BLOCK * sizeof(TS) / 4 : 0; // (bvh_depth x 256 entries: a multiple of 16 bytes either way)
    const float4 *lnodes = (const float4 *)(s_trav + trav_words);
    // (big scenes: A.bvh_lds_f4 covers the first nodes of the breadth-first top of the tree only)
    const uint32_t top = LN ? 0u : A.bvh_lds_f4 >> 2;
    if (VARIANT == 4 && A.bvh_lds_f4)
    {
        float4 *dst = (float4 *)(s_trav + trav_words);
        for (uint32_t i = (uint32_t)tid; i < A.bvh_lds_f4; i += R1_BLOCK)
        {
            float4 q = (R1_BVH4 && LN) ? A.bvh_wide[i] : A.scene.bvh_nodes[i]; // (4-wide table: references already in the 16-bit form)
            if (LN && !R1_BVH4 && (i & 3u) == 3u) // {A K child0 child1}: the small-scene kernels walk with 16-bit child references
                q.z = __uint_as_float(r1_ref16(__float_as_uint(q.z))), q.w = __uint_as_float(r1_ref16(__float_as_uint(q.w)));
            dst[i] = q;
        }
        // node 0's K slot (0 for these trees: K is part of the half extents) carries the root step's code, see bvh_advance (written by
        // the thread that copied that row: program order; outside the loop, where the test made the compiler peel and unroll the copy)
        if (LN && tid == 3)
            ((float *)dst)[13] = __uint_as_float(A.scene.bvh_root_leaf);
        if (LN && !BATCH && R1_ENTRY_MODE(MODE) && A.entry_lds) // the primary rays' entry nodes, 16 bits each (all ones: the walk is over after the root step)
            for (uint32_t i = (uint32_t)tid; i < A.entry_lds; i += R1_BLOCK)
                ((uint16_t *)(dst + A.bvh_lds_f4))[i] = (uint16_t)A.bvh_entry[i];
        __syncthreads();
    }

    Path p;
    p.o = mk(0, 0, 0), p.d = mk(0, 0, 0);
    p.s_scalar = p.s0 = p.s1 = p.s2 = 1;
    p.k = 0, p.depth = 0, p.sp = 0;
    bool alive = false;
    Pixel px; // PIXEL mode: the pixel this lane owns
    px.acc = mk(0, 0, 0), px.xy = 0, px.off = 0, px.s = (uint32_t)A.spp;
    Trav tv; // tree kernels: this lane's walk (carried over outer iterations, see bvh_advance)
    trav_start(tv);
    tv.cur = R1_BVH_DONE;
    unsigned long long lane_rays = 0;
    // Frames in flight (MODE 0; tree kernels and the small-scene exhaustive sweep, 18.25 -> 19.0 Grays/s at 96 VGPRs, its
    // limit for five waves per SIMD): every lane keeps ONE prepared sample (its primary ray and stream states,
    // 11 registers) next to the path it is tracing.  A lane whose path ends takes its own spare, and spares are generated
    // for all lanes that lack one at once — when a lane has died without one, or R1_SPARE_MIN lanes lack one — instead of
    // for the ~36 % of the lanes that died in this iteration: the ~200 instructions of hashing, lens disk and camera ray
    // run every second iteration at ~60 % of the lanes.  29.4 -> 30.5 Grays/s (thresholds 20 / 32 / 44 / never by count:
    // 30.2 / 30.5 / 30.6 / 30.4).  Not for a synchronous frame: what the lanes hold when the queue runs dry is the
    // frame's tail (1.17 -> 1.23 ms with spares).
#ifndef R1_SPARE
#define R1_SPARE 1
#endif
#ifndef R1_SPARE_MIN
#define R1_SPARE_MIN 40u
#endif
    constexpr bool SPARE = R1_SPARE && !BIG && (VARIANT == 4 || VARIANT == 2) && (MODE == 0 || MODE == 3); // (big scenes: the registers buy an eighth wave instead)
    Path spare = p;
    bool has_spare = false;

    // wave-uniform queue state.  Chunks shrink as the queue drains (guided self-scheduling):
    // a wave asks for ~1/(2*waves) of what it last saw remaining, between R1_CHUNK_MIN and
    // R1_CHUNK samples, so that the last waves to finish hold little work.
    // LAND: chunks come from the cursor of this XCD (A.queue + 32 xcd, see above); q_next / q_end are sample slots of the launch either way
    uint32_t q_next = 0, q_end = 0;
    const uint32_t q_total = A.total_samples;
    uint32_t q_remaining = q_total;
    const uint32_t n_waves2 = 2u * n_tb * (R1_BLOCK / 64);
    uint32_t *const q_head = LAND ? A.queue + 32u * xcd : A.queue;
    // tiles an XCD claims at a time.  A synchronous frame has the whole chip on 64-sample chunks: an XCD is through a tile in ~8 us, and
    // the ~5 us its cursor is closed while the next claim is installed would be most of that (1.65 ms per frame with one tile per claim)
    constexpr uint32_t LAND_GROUP = LAT ? 8u : 1u;
    __shared__ uint32_t s_land[LAND ? (R1_BLOCK / 64) * R1_LAND_ROW : 1];
#define row (s_land + (LAND ? (threadIdx.x >> 6) * R1_LAND_ROW : 0u)) /* this wave's row (see land_count); recomputed where it is used: one register less around the loop */
    if (LAND && lane < 8)
        row[lane] = (lane == 0 || lane == 2) ? 0xFFFFFFFFu : 0u;
    bool exhausted = false;
    // sub-queue this wave pulls from (A.nq > 1), wave-uniform
    // (workgroups b .. b + 7 sit on the eight XCDs and share their sub-queues, so every sub-queue is served from
    // every XCD: the XCDs of one chip ran this kernel up to 20 % apart in speed, tools/wave_timeline.py)
    const uint32_t home = LAT && A.nq > 1 ? __builtin_amdgcn_readfirstlane(((tb >> 3) * (R1_BLOCK / 64) + (threadIdx.x >> 6)) % A.nq) : 0u;

    for (;;)
    {
        // ---- refill finished lanes from the wave's chunk of the global sample queue ----
        if (STATS)
            wstat[15] = __builtin_readcyclecounter();
        if (PIX && !alive && px.s < (uint32_t)A.spp)
        {
            // the next sample of the pixel in hand: no queue, no index arithmetic
            start_ray(A, p, (int)(px.xy & 0xFFFFu), (int)(px.xy >> 16), px.s, A.seed);
            alive = true;
            if (VARIANT == 4)
                trav_start(tv);
        }
        if (SPARE && !alive && has_spare)
        {
            p.o = spare.o, p.d = spare.d, p.s_scalar = spare.s_scalar, p.s0 = spare.s0, p.s1 = spare.s1, p.s2 = spare.s2, p.k = spare.k;
            p.depth = 0, p.sp = 0;
            alive = true, has_spare = false;
            trav_start(tv);
        }
        unsigned long long need = __ballot(!alive);
        if (SPARE)
        {
            const unsigned long long lack = __ballot(!has_spare);
            need = (need != 0ull || (uint32_t)__popcll(lack) >= R1_SPARE_MIN) ? lack : 0ull;
        }
        while (need)
        {
            if (q_next == q_end)
            {
                if (exhausted)
                    break;
                uint32_t want, base = 0;
                if (LAT && A.nq > 1)
                {
                    // fixed chunks dealt round-robin to the sub-queues: chunk j of sub-queue `home` is chunk j nq + home
                    want = A.chunk_max;
                    if (lane == 0)
                        base = atomicAdd(A.queue + 32u * home, 1u);
                    base = (__builtin_amdgcn_readfirstlane(base) * A.nq + home) * want;
                }
                else if (LAND)
                {
                    // guided as the single queue is: ~ what is left of the launch / (2 waves) — here: the tiles nobody has claimed yet
                    want = min(A.chunk_max, max(A.chunk_min, q_remaining / n_waves2));
                    base = q_total; // (exhausted unless a chunk is found)
                    for (;;)
                    {
                        unsigned long long old = 0;
                        if (lane == 0)
                            old = __hip_atomic_fetch_add((unsigned long long *)q_head, (unsigned long long)want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(old >> 32)); // group + 1; 0: none installed yet; ~0: none left
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old);          // slots of the group handed out before this call
                        if (hi == 0xFFFFFFFFu)
                            break;
                        // a group = LAND_GROUP consecutive tiles claimed at once (the last one of a launch may be shorter)
                        const uint32_t g0 = (hi - 1u) * LAND_GROUP;
                        const uint32_t gs = hi != 0u ? min(LAND_GROUP, land_tiles - g0) * A.full : 0u;
                        const bool have = hi != 0u && lo < gs;
                        const bool advance = (hi == 0u && lo == 0u) || (have && lo + want >= gs); // exactly one wave per installed group (and per XCD at the start)
                        if (advance && lane == 0)
                        {
                            const uint32_t g = __hip_atomic_fetch_add(A.queue + 32u * 16u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            unsigned long long nv = 0xFFFFFFFFull << 32;
                            if (g * LAND_GROUP < land_tiles)
                                nv = (unsigned long long)(g + 1u) << 32;
                            (void)__hip_atomic_exchange((unsigned long long *)q_head, nv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (have)
                        {
                            base = g0 * A.full + lo;
                            want = min(want, gs - lo);
                            q_remaining = (land_tiles - min(land_tiles, g0 + LAND_GROUP)) * A.full; // (the tiles behind this group: what the next call is guided by)
                            const uint32_t tl = fastdiv(lo, A.div_full); // the chunk's (first) tile within the group
                            if (lane == 0)
                            {
                                land_chunk(A, row, g0 + tl);
                                if (lo - tl * A.full + want > A.full) // (it runs on into the next tile: a chunk is at most one tile long)
                                    land_chunk(A, row, g0 + tl + 1u);
                            }
                            break;
                        }
                        if (advance)
                            continue; // installed the XCD's first tile myself: take from it
                        // a used-up tile (or none yet): the wave that installs the next one is a few atomics away
                        for (;;)
                        {
                            __builtin_amdgcn_s_sleep(16);
                            unsigned long long now = 0;
                            if (lane == 0)
                                now = __hip_atomic_load((unsigned long long *)q_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(now >> 32)) != hi)
                                break;
                        }
                    }
                }
                else
                {
                    want = min(A.chunk_max, max(A.chunk_min, q_remaining / n_waves2));
                    if (lane == 0)
                        base = atomicAdd(q_head, want);
                    base = __builtin_amdgcn_readfirstlane(base);
                }
                if (base >= q_total)
                {
                    // No stealing between sub-queues: chunks are dealt round-robin, so the sub-queues run dry together, the
                    // host gives every sub-queue home waves on every XCD, and looking for leftovers elsewhere costs more than
                    // it brings (nq failed atomics of ~1 us each at the end of every wave; a one-load scan of the heads
                    // spilled 50 SGPRs in this loop).
                    exhausted = true;
                    if (STATS)
                        log_exhausted = __builtin_amdgcn_s_memrealtime();
                    break;
                }
                q_next = base;
                q_end = min(base + want, q_total);
                if (!LAND)
                    q_remaining = q_total - q_end;
            }
            const uint32_t avail = q_end - q_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
#if R1_FRESH
            // (what start_sample reads: tiling, fast divisions, camera — fetched here, not carried around the loop)
            R1ArgWords FA_words;
            if (!PIX)
                fresh_args(FA_words);
            const R1TraceArgs &FA = PIX ? A : FA_words.a;
#else
            const R1TraceArgs &FA = A;
#endif
            // (Generating the primary rays in a separate full-width kernel and loading them here was
            // measured: 1.52 ms per frame against 1.28 — the extra launch per frame costs more overlap
            // between frames than the refill saves.)
            if (SPARE)
            {
                if (!has_spare && rank < avail)
                    has_spare = start_sample<BATCH>(FA, spare, q_next + rank); // false: void slot, ask again
            }
            else if (!alive && rank < avail)
            {
                if (PIX)
                {
                    alive = pixel_take(A, px, q_next + rank); // false: void slot, ask again
                    if (alive)
                        start_ray(A, p, (int)(px.xy & 0xFFFFu), (int)(px.xy >> 16), 0u, A.seed);
                }
                else
                    alive = start_sample<BATCH>(FA, p, q_next + rank); // false: void slot, ask again
                if (VARIANT == 4 && alive)
                    trav_start(tv);
            }
            q_next += min((uint32_t)__popcll(need), avail);
            need = SPARE ? __ballot(!has_spare) : __ballot(!alive);
        }
        if (SPARE && !alive && has_spare) // the lanes that had died without a spare start on the one they just got
        {
            p.o = spare.o, p.d = spare.d, p.s_scalar = spare.s_scalar, p.s0 = spare.s0, p.s1 = spare.s1, p.s2 = spare.s2, p.k = spare.k;
            p.depth = 0, p.sp = 0;
            alive = true, has_spare = false;
            trav_start(tv);
        }
        if (BIG && VARIANT == 2)
        {
            // lock-step workgroup: leave together (every wave must reach the sweep's barriers)
            if (!__syncthreads_or(alive ? 1 : 0))
                break;
        }
        else if (__ballot(alive) == 0ull)
            break;
        // A frame's critical path is its longest bounce chain (up to 51 dependent sweeps, ~1/3 of
        // a 10-spp frame when the wave shares its SIMD with three others).  Waves that carry a
        // deep path ask for issue priority so that the chain finishes before the queue runs dry.
        {
            const int prio = __ballot(alive && p.depth >= 36) ? 3 : __ballot(alive && p.depth >= 24) ? 2 : __ballot(alive && p.depth >= 12) ? 1 : 0;
            if (prio == 3)
                __builtin_amdgcn_s_setprio(3);
            else if (prio == 2)
                __builtin_amdgcn_s_setprio(2);
            else if (prio == 1)
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
        }
        if (STATS)
        {
            wstat[4] += __builtin_readcyclecounter() - wstat[15];
            wstat[15] = __builtin_readcyclecounter();
            wstat[0] += 1;
            wstat[1] += (unsigned long long)__popcll(__ballot(alive));
        }

        // ---- one color() level: hit test for every live lane (rayweek1.cpp:519) ----
        float t_hit = FLT_MAX;
        int hit = -1;
        bool ready = alive; // lanes whose hit test is complete after this step (tree kernels: the others carry their walk over)
        const unsigned long long live_now = __ballot(alive);
        if (LAT && !BIG && VARIANT != 1 && exhausted && (uint32_t)__popcll(live_now) <= A.coop_lanes)
        {
            cooperative_sweep(A.scene, live_now, p.o, p.d, t_hit, hit, lane); // the frame's tail: few paths left in this wave
            tv.cur = R1_BVH_DONE;                                            // (a walk in progress is simply dropped: this sweep is complete by itself)
        }
        else if (VARIANT == 1)
        {
            if (alive)
                sweep_reference(A.scene, p.o, p.d, t_hit, hit);
        }
        else if (VARIANT == 4)
        {
            // while-while with carry-over
            constexpr bool ENTRY = R1_ENTRY_MODE(MODE);
#if R1_FRESH
            R1_FRESH_ARGS(HA) // (table pointers, the tree's centre: fetched for the walk, free again after it)
#else
            const R1TraceArgs &HA = A;
#endif
            // (a primary ray that starts its walk: depth 0 and at the root)
            const uint32_t entry_idx = ENTRY && p.depth == 0 && tv.cur == 0u ? fastdiv(p.k, HA.div_full) : 0xFFFFFFFFu;
            bvh_advance<STATS, true, LN, TS, ENTRY>(HA.scene, p.o, p.d, tv, (TS *)s_trav, tid, (uint32_t)__popcll(live_now), wstat, lnodes, top, HA.bvh_entry, entry_idx,
                                                    (LN && !BATCH && HA.entry_lds) ? (const uint16_t *)(lnodes + HA.bvh_lds_f4) : nullptr);
            ready = alive && tv.cur == R1_BVH_DONE;
            if (tv.best_id != 0xFFFFFFFFu)
                t_hit = tv.best, hit = (int)tv.best_id;
        }
        else
        {
#if R1_FRESH
            R1ArgWords HA_words;
            if (!PIX)
                fresh_args(HA_words);
            const R1TraceArgs &HA = PIX ? A : HA_words.a;
#else
            const R1TraceArgs &HA = A;
#endif
            sweep_prefilter<STATS, IDX, BIG>(HA.scene, alive, p.o, p.d, t_hit, hit, s_cand, s_pairs, s_best, s_tile, tid, wstat);
        }
        if (STATS)
        {
            wstat[6] += __builtin_readcyclecounter() - wstat[15];
            wstat[15] = __builtin_readcyclecounter();
        }

        bool fin = false; // this lane's sample ended in this iteration
#if R1_FRESH
        // (material tables, the stack workspace, where the records go; PIXEL mode keeps the prologue's copy: its pixel_write indexes the
        //  arguments in a way that leaves the fresh copy in scratch)
        R1ArgWords SA_words;
        if (!PIX)
            fresh_args(SA_words);
        const R1TraceArgs &SA = PIX ? A : SA_words.a;
#else
        const R1TraceArgs &SA = A;
#endif
        if (ready)
        {
            V3 col;
            if (shade_level<BIG, LW>(SA, p, hit, t_hit, s_stack, gstride, gtid, tid, col))
            {
                if (PIX)
                {
                    px.acc = vadd(px.acc, col); // col += color(...), samples in order (rayweek1.cpp:762)
                    if (++px.s == (uint32_t)SA.spp)
                        pixel_write(SA, px);
                }
                else if (LAND)
                    SA.samples[p.k] = make_float4(col.x, col.y, col.z, __uint_as_float(path_rays(p) | SA.land_tag));
                else
                    SA.samples[p.k] = make_float4(col.x, col.y, col.z, __uint_as_float(path_rays(p)));
                if (!LAND)
                    lane_rays += path_rays(p); // (LAND: the waves that sum the tiles add up the records' counts)
                alive = false, fin = true;
            }
            else if (VARIANT == 4)
                trav_start(tv); // the scattered ray starts its walk at the root
        }
        if (LAND && fin)
            land_count(SA, row, p.k);
        if (STATS)
            wstat[7] += __builtin_readcyclecounter() - wstat[15];
    }
    if (STATS)
    {
        // [0] wave iterations [1] alive lanes [2] candidate-loop trips [3] overflow lanes
        // [4] refill cycles [5] pass-1 cycles [6] candidate cycles [7] shade cycles [8] wave cycles
        // [9] candidates (all lanes)
        wstat[8] = __builtin_readcyclecounter() - wstat[14];
        unsigned long long c9 = wstat[9];
        for (int off = 32; off > 0; off >>= 1)
            c9 += __shfl_down(c9, off, 64);
        wstat[9] = c9;
        if (VARIANT == 4)
        {
            // per-lane counters of sweep_bvh
            const int slots[5] = {2, 3, 5, 16, 17};
            for (int q = 0; q < 5; ++q)
            {
                unsigned long long c = wstat[slots[q]];
                for (int off = 32; off > 0; off >>= 1)
                    c += __shfl_down(c, off, 64);
                wstat[slots[q]] = c;
            }
        }
        if (lane == 0 && wave_log)
        {
            unsigned long long *rec = wave_log + 4 * (size_t)(blockIdx.x * (R1_BLOCK / 64) + (threadIdx.x >> 6));
            rec[0] = log_start, rec[1] = log_exhausted, rec[2] = __builtin_amdgcn_s_memrealtime(), rec[3] = wstat[0];
        }
        if (lane == 0 && A.stats)
        {
            for (int i = 0; i < 10; ++i)
                atomicAdd(&A.stats[i], wstat[i]);
            atomicMax(&A.stats[10], wstat[8]);                      // longest wave (cycles)
            atomicMax(&A.stats[11], ~wstat[8]);                     // ~shortest wave
            atomicMax(&A.stats[12], (unsigned long long)__builtin_readcyclecounter());  // last wave end (this XCD's counter)
            atomicMax(&A.stats[13], ~wstat[14]);                    // ~first wave start
            if (VARIANT == 4)
            {
                atomicAdd(&A.stats[14], wstat[16]);                 // tree: leaf trips summed over lanes
                atomicAdd(&A.stats[15], wstat[17]);                 // tree: root steps (bvh_advance) summed over lanes
            }
        }
    }

    if (LAND)
        land_exit<(BIG && VARIANT == 4) ? 6 : R1_LAND_LOADS>(A, row, lane); // (the big-scene tree kernels are built for 64 registers)
#undef row
    // ray count: wave reduction, one atomic per wave (rayweek1.cpp:809-813)
    if (!LAND)
    {
        for (int off = 32; off > 0; off >>= 1)
            lane_rays += __shfl_down(lane_rays, off, 64);
        if (lane == 0 && lane_rays)
            atomicAdd(A.num_rays, lane_rays);
    }
}

#endif // R1_TRACE_HPP
