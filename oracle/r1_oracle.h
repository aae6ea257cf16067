/* oracle/r1_oracle.h — TEST INFRASTRUCTURE: CPU restatement of the reference's hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  Nothing under rays1bench_amd/ links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit in
 * tests/test_oracle_golden.py against fixtures produced by the reference's own code
 * (oracle/ref_harness.cpp / ref_harness_step1.cpp compiled from /root/reference with
 * g++ 11.4 `-O2 -mavx2 -mfma -ffp-contract=off`; generator: oracle/gen_golden.py).
 */
#ifndef R1_ORACLE_H
#define R1_ORACLE_H

#include <stdint.h>
#include "../include/rays1.h"
#include "../include/rays1_seed.h"

#ifdef __cplusplus
extern "C" {
#endif

/* mymath.h:17-35 */
uint32_t r1o_xorshift32(uint32_t *state);
float r1o_rand01(uint32_t *state);
float r1o_rand02(uint32_t *state);
/* mymath.h:41-73, lanes 0..3 */
void r1o_rand01_x4(uint32_t state4[4], float out[4]);
void r1o_rand02_x4(uint32_t state4[4], float out[4]);

/* One pixel-sample under the seeding contract: rayweek1.cpp:757-763 with the stream
 * states of include/rays1_seed.h.  rgb = color(), *rays = color() invocations. */
void r1o_trace_sample(const r1_scene *scene, const r1_camera *cam, int32_t width, int32_t height, int32_t max_bounces,
                      uint32_t seed, int32_t x, int32_t y, int32_t s, float rgb[3], uint32_t *rays);

/* Whole frame under the seeding contract: per-sample results summed in sample order and
 * resolved as rayweek1.cpp:765-775.  samples_out (optional) gets width*height*spp records
 * {r,g,b,bit_cast<float>(rays)}.  nthreads <= 0 uses all hardware threads. */
int r1o_render_frame(const r1_scene *scene, const r1_camera *cam, const r1_params *params, uint8_t *rgb_out,
                     uint64_t *num_rays_out, float *samples_out, int32_t nthreads);

/* The reference's deterministic single-thread path (rayweek1.cpp:879-888): tiles 0..T-1
 * through render_tile (rayweek1.cpp:722-782) with the sequential streams
 * state = 10001, state4 lanes (l0..l3) = (1007, 1005, 1003, 1001). */
int r1o_render_sequential(const r1_scene *scene, const r1_camera *cam, int32_t width, int32_t height, int32_t spp,
                          int32_t max_bounces, uint8_t *rgb_out, uint64_t *num_rays_out);

/* The reference's multi-threaded scheduler semantics (rayweek1.cpp:785-842): one
 * sequential stream pair per thread (200*i + 10001 ...), atomic tile counter.  Not
 * reproducible run to run, like the reference; used only as the "port" CPU baseline. */
int r1o_render_threads(const r1_scene *scene, const r1_camera *cam, int32_t width, int32_t height, int32_t spp,
                       int32_t max_bounces, int32_t nthreads, uint8_t *rgb_out, uint64_t *num_rays_out);

/* BASELINE config 1: step1 semantics (src/step1/rayweek1.cpp), small scene
 * (step1/rayweek1.cpp:537-558), serial loop with the global stream 1236787
 * (step1/rayweek1.cpp:32, :689-727). */
int r1o_step1_small(int32_t width, int32_t height, int32_t spp, uint8_t *rgb_out, uint32_t *num_rays_out);

#ifdef __cplusplus
}
#endif
#endif
