/*
 * vrc_oracle.h -- CPU restatement ("oracle") of the CpuVoxelRaycaster voxel
 * ray-traversal hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product (cpuvoxelraycaster_amd/, include/vrc.h) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" in the strict sense.  The reference holds no
 * tests, golden vectors or fixtures for this path, and
 * its traversal / shading headers need GLM and SFML, which this image lacks,
 * so they cannot be compiled here without stand-in headers (not allowed).
 * What IS pinned:
 *   - terrain heights: the restated SimplexFractal noise (orc_terrain_heights)
 *     equals, for all 1024 x 1024 columns, the output of the real vendored
 *     FastNoise compiled from /root/reference (oracle/_ref, oracle/Makefile,
 *     tests/tools/make_golden.py -> tests/golden/terrain_height_1024.npz) -- this row
 *     is pinned by reference execution;
 *   - the camera-path replay reader (include/replay.hpp, standard library only) is compiled where it lies into
 *     oracle/_ref/libref_replay.so; what it parses from a set of replay texts is the fixture
 *     tests/golden/replay_cases.json (tests/tools/make_golden_replay.py) -- pinned by reference execution too;
 *   - the restated builder + traversal reproduce the statistics SURVEY.md
 *     App. B measured on the unmodified reference (node count 10 528 393 at
 *     512^3, first nodes, level histogram, per-pose hit counts, mean / max
 *     loop iterations, normal-axis split, a sample hit), and its instrumented
 *     random-ray / GI-ray run replayed draw for draw with the restated xorshf96
 *     stream (5 025 random-ray hits; 116 671 GI hits = 77 971 LOD cut-offs with
 *     the exact per-scale histogram + 38 700 leaf hits) -- tests/test_oracle_pins.py.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Build with -O2 -ffp-contract=off (FMA contraction changes
 * traversal results).
 */
#ifndef VRC_ORACLE_H
#define VRC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/lsvo_utils.hpp:5-18 -- 8-byte node; byte 3 is struct padding. */
typedef struct orc_lnode {
    uint8_t color;        /* always 1, never read (lsvo_utils.hpp:11) */
    uint8_t child_mask;
    uint8_t leaf_mask;
    uint8_t pad;          /* indeterminate in the reference; 0 here */
    uint32_t child_offset;
} orc_lnode;

/* include/volumetric.hpp:7-22 -- HitPoint, with the borrowed Cell* replaced
 * by `hit` (0 miss, 1 leaf hit lsvo.hpp:92-95, 2 LOD cut lsvo.hpp:82-85) and
 * the node identity made explicit.  On a miss every field except `hit`,
 * `complexity` is 0 (uninitialised in the reference). 48 bytes. */
typedef struct orc_hit {
    float position[3];
    float normal[3];
    float voxel_coord[2];
    uint32_t hit;         /* kind | child_shift<<8 | scale<<16 */
    uint32_t node;        /* parent_id of the node whose child was hit */
    float distance;
    uint32_t complexity;
} orc_hit;

/* ---- T2: pointer octree + compile (svo.hpp:72-114, lsvo_utils.cpp:4-49) ---- */
typedef struct orc_svo orc_svo;
orc_svo *orc_svo_new(uint32_t depth);
void orc_svo_free(orc_svo *s);
void orc_svo_set_cell(orc_svo *s, uint32_t x, uint32_t y, uint32_t z);
/* main.cpp:63-76 with the noise already evaluated: height[x*size+z] is the
 * int32 `height` of main.cpp:69; solid for y in [1, max(16,min(size,height)))
 * stored at y + size/2. */
void orc_svo_fill_terrain(orc_svo *s, const int32_t *height);
/* returns node count; *out is malloc'd (free with orc_free). */
uint64_t orc_svo_compile(const orc_svo *s, orc_lnode **out);
void orc_free(void *p);

/* ---- N4: terrain heights (main.cpp:61-69 over lib/fastnoise SimplexFractal);
 * height[x*size + z]; pinned against the real FastNoise build (oracle/_ref). ---- */
void orc_terrain_heights(int32_t size, int32_t seed, int32_t *height);

/* ---- T1: LSVO<N>::castRay (lsvo.hpp:33-172) ---- */
void orc_cast_ray(const orc_lnode *nodes, uint32_t depth,
                  const float org[3], const float dir[3],
                  float ray_size_coef, float ray_size_bias, orc_hit *out);
/* batch form, optional per-ray coef/bias arrays (NULL = 0); threads<=1 serial */
void orc_cast_rays(const orc_lnode *nodes, uint32_t depth, uint64_t n,
                   const float *org_xyz, const float *dir_xyz,
                   const float *coef, const float *bias,
                   orc_hit *out, int threads);

/* ---- G1: Grid3D::castRay (grid_3d.hpp:36-132), cells = type bytes [x][y][z] ---- */
void orc_grid_cast_ray(const uint8_t *cells, int32_t X, int32_t Y, int32_t Z,
                       const float org[3], const float dir[3], orc_hit *out);
void orc_grid_cast_rays(const uint8_t *cells, int32_t X, int32_t Y, int32_t Z,
                        uint64_t n, const float *org_xyz, const float *dir_xyz,
                        orc_hit *out, int threads);

/* ---- U1/U2: scalar utilities (utils.cpp) ---- */
float orc_frac(float f);                                  /* utils.cpp:60-64 */
uint32_t orc_counter_rand(uint32_t seed, uint32_t frame, uint32_t pixel, uint32_t draw);
float orc_get_rand_counter(uint32_t k, float mn, float mx); /* utils.cpp:77-81 quantisation */
void orc_xorshf96_reset(void);                            /* utils.cpp:11 */
uint64_t orc_xorshf96(void);                              /* utils.cpp:13-25 (LP64) */
void orc_xorshf96_fill(uint64_t n, uint64_t *out);
void orc_make_rotation(float angle_x, float angle_y, float rot[9]); /* utils.cpp:94-100 */

/* ---- R0 + S1..S4: per-frame render (main.cpp:131-152, raycaster.hpp:67-207) ---- */
typedef struct orc_camera {
    float position[3];    /* world voxel units (main.cpp:51) */
    float rot[9];         /* glm::mat3 columns m[0],m[1],m[2] (camera_controller.hpp:21) */
    float fov;
    float aperture;
    float focal_length;
} orc_camera;

typedef struct orc_frame_params {
    uint32_t width, height;
    float light_position[3];   /* SVO space, as setLightPosition gets it (main.cpp:126) */
    uint32_t use_gi;           /* raycaster.hpp:274 */
    uint32_t use_samples;      /* raycaster.hpp:275 */
    uint32_t shadow_samples;   /* 0 = reference default (use_samples ? 4 : 1), raycaster.hpp:147 */
    uint32_t gi_bounces;       /* 1 = reference; 2 = extension */
    int32_t checker_parity;    /* -1 = every pixel; 0/1 = main.cpp:143 checkerboard offset */
    uint32_t seed;
    uint32_t frame_index;      /* sample index: RNG key */
    uint32_t rng_mode;         /* 0 counter-based, 1 reference xorshf96 serial stream */
    uint32_t row_block;        /* rows per shard block (0 = no sharding) */
    uint32_t shard_index, shard_count; /* row-block b belongs to shard b % shard_count */
} orc_frame_params;

typedef struct orc_frame_stats {
    uint64_t rays;             /* LSVO::castRay calls executed */
    uint64_t sum_complexity;   /* sum of HitPoint::complexity over those calls */
    uint64_t primary_hits;
} orc_frame_stats;

/* accum: 4 x uint32 per pixel (r,g,b,count) -- integer restatement of the
 * double-precision Sample (raycaster.hpp:18-24), exact while sums < 2^32.
 * image: RGBA8 row-major, width*height*4; read-modify-written in non-sample
 * mode (raycaster.hpp:77-85).  prim (optional): per-pixel primary orc_hit. */
void orc_render_frame(const orc_lnode *nodes, uint32_t depth,
                      const uint8_t tex_top[768], const uint8_t tex_side[768],
                      const orc_camera *cam, const orc_frame_params *p,
                      uint8_t *image, uint32_t *accum, orc_hit *prim,
                      orc_frame_stats *stats, int threads);
/* orc_render_frame with the reference's thread decomposition: 16 threads x static 4x4 areas, x-outer / y-inner
 * (main.cpp:90-92,139-152); the CPU-baseline variant SURVEY 8d asks for. */
void orc_render_frame_areas16(const orc_lnode *nodes, uint32_t depth,
                              const uint8_t tex_top[768], const uint8_t tex_side[768],
                              const orc_camera *cam, const orc_frame_params *p,
                              uint8_t *image, uint32_t *accum, orc_frame_stats *stats);
/* The swarm lambda's body as a thread-pool job (main.cpp:139-152): worker `id` of 16 renders its area.  `user` points
 * to an orc_area_frame. */
typedef struct orc_area_frame {
    const orc_lnode *nodes;
    uint32_t depth;
    const uint8_t *tex_top, *tex_side;
    const orc_camera *cam;
    const orc_frame_params *params;
    uint8_t *image;
    uint32_t *accum;
    uint64_t rays[16], sum_complexity[16];
} orc_area_frame;
void orc_area_job(uint32_t id, uint32_t worker_count, void *user);
/* Analysis hook (tools/): the same frame, and per pixel the loop-iteration counts of its first ORC_TRACE_RAYS
 * castRay calls in cast order (primary, shadow..., GI, GI shadow, ...) into trace[pixel * ORC_TRACE_RAYS + k]
 * (untouched entries stay as the caller set them).  trace == NULL is orc_render_frame. */
#define ORC_TRACE_RAYS 8
void orc_render_frame_trace(const orc_lnode *nodes, uint32_t depth,
                            const uint8_t tex_top[768], const uint8_t tex_side[768],
                            const orc_camera *cam, const orc_frame_params *p,
                            uint8_t *image, uint32_t *accum, orc_hit *prim,
                            orc_frame_stats *stats, int threads, uint16_t *trace);
/* ... and one op code per loop iteration of each of those rays (0 descend, 1 advance, 2 advance + pop, 3 the iteration
 * ending in a hit) into ops[(pixel * ORC_TRACE_RAYS + ray) * ops_len + iteration] (tests/tools/path_schedule.py). */
void orc_render_frame_optrace(const orc_lnode *nodes, uint32_t depth,
                              const uint8_t tex_top[768], const uint8_t tex_side[768],
                              const orc_camera *cam, const orc_frame_params *p,
                              int threads, uint16_t *trace, uint8_t *ops, uint32_t ops_len);
/* raycaster.hpp:94-103 / :105-116 */
void orc_samples_to_image(const uint32_t *accum, uint32_t width, uint32_t height, uint8_t *image);

/* ---- N2: post-process / present chain (main.cpp:160-182; res/median_3.frag, res/median.frag) ----
 * image: the RayCaster's render_image (RGBA8, width x height).  denoised: the persistent denoised_tex, updated in
 * place: denoised = sat(round(denoised * c1 / 255) + round(image * c2 / 255)), c1 = uint8(255 * old), c2 = uint8(255 *
 * (1 - old)) (main.cpp:161-166; old = use_samples ? 0 : 0.1).  window (optional): out_width x out_height RGBA8 = the
 * denoised image scaled with nearest sampling (main.cpp:179-181, scale 1 / 0.75), through an optional per-channel
 * median (0 = none, 3 = 3x3, 5 = 5x5: the shaders' min/max networks, taps one texel apart, clamped to the edge). */
void orc_present(const uint8_t *image, uint8_t *denoised, uint32_t width, uint32_t height,
                 float old_value_conservation, uint32_t median, uint8_t *window, uint32_t out_width, uint32_t out_height);

#ifdef __cplusplus
}
#endif
#endif
