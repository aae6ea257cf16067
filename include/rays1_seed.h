/* rays1_seed.h — the per-sample seeding contract (build-defined; the reference has none).
 *
 * The reference (src/step13/rayweek1.cpp:798-802) gives every OS thread ONE pair of
 * sequential xorshift32 streams (`state`, `state4`) that run on across samples, pixels
 * and tiles, and hands tiles to threads through a race (rayweek1.cpp:830-838), so its
 * multi-threaded output is not reproducible and cannot be followed by a GPU.  This
 * build instead derives the stream states of every (pixel, sample) from a hash of
 * (seed, pixel index, sample index) and then follows the reference's exact draw order
 * INSIDE the sample (rayweek1.cpp:759-762).  Anything that renders with this contract
 * (the HIP kernel, oracle/r1_oracle.c, oracle/ref_harness.cpp driving the reference's
 * own color()/getRay()) produces the same image for any tiling / device count.
 *
 * Streams per sample:
 *   scalar  — `ThreadData::state`   (random_in_unit_disk, Dielectric::scatter)
 *   lane0..2 — lanes 0..2 of `ThreadData::state4` (pixel jitter uses lanes 0,1 of one
 *             draw; random_in_unit_sphere uses lanes 0,1,2).  Lane 3 of state4 never
 *             reaches any result in the reference (Vec3 dot() only sums lanes 0..2,
 *             mymath.h:205-207) so it has no stream here; harnesses that need a value
 *             for it use R1_SEED_LANE3.
 */
#ifndef RAYS1_SEED_H
#define RAYS1_SEED_H

#include <stdint.h>

#if defined(__HIPCC__)
#define R1_HD __host__ __device__ static inline
#else
#define R1_HD static inline
#endif

#define R1_SEED_LANE3 0x9E3779B9u

/* 32-bit finaliser (public-domain "lowbias32" constants). Bijective on u32. */
R1_HD uint32_t r1_mix32(uint32_t v)
{
    v ^= v >> 16;
    v *= 0x7FEB352Du;
    v ^= v >> 15;
    v *= 0x846CA68Bu;
    v ^= v >> 16;
    return v;
}

/* xorshift32 has the single fixed point 0; never hand it out. */
R1_HD uint32_t r1_nonzero(uint32_t v)
{
    return v ? v : 0x6C078965u;
}

typedef struct r1_sample_seed
{
    uint32_t scalar; /* ThreadData::state            */
    uint32_t lane0;  /* lane 0 of ThreadData::state4 */
    uint32_t lane1;  /* lane 1                        */
    uint32_t lane2;  /* lane 2                        */
} r1_sample_seed;

/* pixel = y * width + x  (row y = 0 is the bottom row, as in the reference's image
 * buffer, rayweek1.cpp:750); sample = s in [0, spp). */
R1_HD r1_sample_seed r1_seed_sample(uint32_t seed, uint32_t pixel, uint32_t sample)
{
    uint32_t h = r1_mix32(seed ^ 0xA511E9B3u);
    h = r1_mix32(h + pixel * 0x9E3779B9u);
    h = r1_mix32(h ^ (sample * 0x85EBCA6Bu + 0xC2B2AE35u));
    r1_sample_seed s;
    s.scalar = r1_nonzero(r1_mix32(h + 0x01234567u));
    s.lane0  = r1_nonzero(r1_mix32(h + 0x3C6EF372u));
    s.lane1  = r1_nonzero(r1_mix32(h + 0xDAA66D2Bu));
    s.lane2  = r1_nonzero(r1_mix32(h + 0x78DDE6E4u));
    return s;
}

#endif /* RAYS1_SEED_H */
