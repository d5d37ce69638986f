/*
 * vrc_walk_model.c -- CPU model of the FORMULATION the HIP walk uses (cpuvoxelraycaster_amd/csrc/vrc_walk.h), to check that
 * formulation against the oracle (oracle/vrc_oracle.c, which follows the reference line by line) ray by ray, here, without a
 * GPU.  TEST INFRASTRUCTURE ONLY (tests/test_walk_model.py); the product never links it.
 *
 * The reference is LSVO<N>::castRay, /root/reference/include/lsvo.hpp:33-172.  What the HIP walk does differently, and what
 * this model therefore restates:
 *
 * 1. No t_max on the stack.  The reference keeps {parent_index, t_max} per level (:97-100, :134-136).  t_max is always
 *    min(1, exit(parent cell)): :57-58 set it so for the root, :110 keeps it (t_max' = min(t_max, tc_max), and tc_max, the exit
 *    of the child cell, is never larger than the exit of the cell around it -- x -> x * t_coef - t_offset is monotone in
 *    float arithmetic too and a child's corner is never below its parent's), :134-136 restore it.  Hence
 *      tv_max (:86)            = min(t_max, tc_max) = min(1, tc_max)
 *      t_min <= t_max (:81)    = t_min <= T1 with the per-ray constant T1 = min(1, exit(root)): inside the cube
 *                                t_min (the exit of a cell inside the parent) never exceeds the parent's exit.
 *    The stack holds parent indices only, pushed on every descend.
 *
 * 2. Start below the root.  A secondary ray starts next to the hit of the ray before it (raycaster.hpp:139,153; :171-176,
 *    :195-198).  Its first iterations are pure descends from the root to the cell that holds its origin: child present and
 *    not a leaf (the previous walk went through the same nodes), t_min = 0 <= exits, the child chosen at :66-68 / :107-109 is
 *    the one that holds the origin.  The model (and the kernel) starts the ray at scale s0 with exactly the state those
 *    22 - s0 iterations produce -- parent from the previous walk's stack, pos = origin bits above s0, complexity 22 - s0 --
 *    provided every comparison of the skipped iterations is decided with a margin:
 *      - both origin and previous hit position are in [1, 2) and agree in all bits above s0 (same cells down to the parent),
 *      - s0 >= scale of the previous hit (those parents are on the stack),
 *      - no multiple of 2^s0 ulps lies within 8 ulps of an origin coordinate (the centre planes the skipped comparisons
 *        test; their rounding error is below 4.5 ulps of the coordinate),
 *      - the LOD test (:82) cannot fire in a cell that holds the origin: tc_max <= scale_f / max|d_i| there, so with bias 0
 *        it needs coef < max|d_i| -- the frame kernels cast with coef 0.5 (or 0) and check max|d_i| > 0.55, which every
 *        unit vector satisfies (>= 0.577) and a direction that collapsed to (+-EPS, +-EPS, +-EPS) does not (the light at
 *        1e30: normalize() returns 0, :44-46 make it EPS, and the reference's first iteration is an LOD hit).
 *    Otherwise the ray starts at the root as in the reference.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "vrc_oracle.h"

#ifndef ORC_MARGIN
#define ORC_MARGIN 8u     /* ulps; the comparisons of the skipped iterations are exact beyond 4.5 */
#endif

int orc_model_descend_by_position = 0;     /* study switch: see model_cast */
/* The margin as a run-time switch (tests/test_walk_model.py::test_constructed_margin_adversaries runs the model at 0..8 ulps to
 * record where it first holds against the oracle); the kernel's is the constant 8 of vrc_device.h: start_scale_next_to. */
uint32_t orc_model_margin = ORC_MARGIN;

typedef struct orc_walk_state {
    uint32_t rows[24];      /* parent index by scale (the LDS stack column of a lane) */
    uint32_t scale;         /* scale of the hit child; rows[scale] = the hit's parent */
    uint32_t valid;         /* the walk ended in a hit */
} orc_walk_state;

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float smin(float a, float b) { return b < a ? b : a; }
static inline float smax(float a, float b) { return a < b ? b : a; }
static inline float gsign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : x); }
static inline float frac_(float x) { return x - floorf(x); }
static inline int highbit(uint32_t x) { return x ? 31 - __builtin_clz(x) : -1; }

/* The paths from the root to the cell around a camera position (vrc_device.h: camera_path), one per combination of ray
 * direction signs: variant bit i = the ray travels towards +axis i.  That only matters for an axis on which the position is
 * EXACTLY 1.5, the centre plane of the cube -- the reference's default camera sits at size / 2 on x and z (main.cpp:51: (256, 200, 256) at 512^3).
 * There the first child choice (:66-68) is `1.5 * t_coef - t_offset > t_min` with t_offset = fl(1.5 * t_coef) (mirrored:
 * fl(3 t_coef) - fl(1.5 t_coef) = 2 fl(1.5 t_coef) - fl(1.5 t_coef), the same number): exactly 0 > 0, false -- in the walk's
 * mirrored frame every such ray takes the half BELOW 1.5, the half it travels into, and from there on its origin is the
 * upper face of its cell at every level: the centre planes are half a cell away (robust), the exits through the lower
 * corner too.  So such an axis takes no margin test; its coordinate counts as 1.5 - 1 ulp in the mirrored frame, i.e. the
 * cells just below 1.5 for a ray towards -axis and just above for a ray towards +axis. */
typedef struct orc_camera_paths {
    uint32_t rows[8][24];
    uint32_t scale[8];      /* the scale a camera ray of that variant starts at (22: at the root) */
    uint32_t onplane;       /* axes whose coordinate is exactly 1.5 */
} orc_camera_paths;

void orc_model_camera_paths(const orc_lnode *raw, uint32_t depth, const float cam[3], orc_camera_paths *out)
{
    memset(out, 0, sizeof(*out));
    uint32_t a[3] = { f2u(cam[0]), f2u(cam[1]), f2u(cam[2]) };
    for (int i = 0; i < 3; i++) if (cam[i] == 1.5f) out->onplane |= 1u << i;
    uint32_t mb = 0u;
    int inside = 1;
    for (int i = 0; i < 3; i++) {
        if (!((out->onplane >> i) & 1u)) mb |= (a[i] - orc_model_margin) ^ (a[i] + orc_model_margin);
        inside = inside && (a[i] >> 23) == 127u;
    }
    const int depth_off = 23 - (int)depth;
    int lowest = highbit(mb + mb);
    if (lowest < depth_off) lowest = depth_off;
    if (lowest < 23 - 3 * ((int)depth / 3)) lowest = 23 - 3 * ((int)depth / 3);    /* the kernel copies rows in groups of three */
    for (uint32_t v = 0; v < 8u; v++) {
        uint32_t b[3];
        for (int i = 0; i < 3; i++) b[i] = (((out->onplane >> i) & 1u) && !((v >> i) & 1u)) ? a[i] - 1u : a[i];
        int s = 22;
        uint32_t parent = 0u;
        if (inside) {
            while (s > lowest) {
                const orc_lnode nd = raw[parent];
                const uint32_t cs = 7u ^ (((b[0] >> s) & 1u) | (((b[1] >> s) & 1u) << 1) | (((b[2] >> s) & 1u) << 2));
                if (!((nd.child_mask >> cs) & 1u) || ((nd.leaf_mask >> cs) & 1u)) break;
                out->rows[v][s] = parent;
                parent += nd.child_offset + cs;
                --s;
            }
        }
        out->rows[v][s] = parent;
        out->scale[v] = (uint32_t)s;
    }
}

/* One ray.  prev / prev_pos: state and hit position of the ray this one starts next to (NULL: start at the root); or cam: the
 * ray starts AT the camera position these paths were made for (org == that position); or cam + lens_of: the ray starts at a
 * lens point NEXT TO the camera position lens_of (camera_controller.hpp:41,46: origin = position + rand_vec * rot).
 * *skipped = iterations not executed (already part of out->complexity). */
static void model_cast(const orc_lnode *raw, uint32_t depth, const float org[3], const float dir[3], float coef, float bias,
                       const orc_walk_state *prev, const float prev_pos[3], const orc_camera_paths *cam, const float *lens_of,
                       orc_hit *out, orc_walk_state *st, uint32_t *skipped)
{
    memset(out, 0, sizeof(*out));
    memset(st, 0, sizeof(*st));
    *skipped = 0;
    const int SVO_MAX = 23;
    const float EPS = 1.0f / (float)(1 << 23);
    const float px = org[0], py = org[1], pz = org[2];
    float dx = dir[0], dy = dir[1], dz = dir[2];
    if (!isfinite(px) || !isfinite(py) || !isfinite(pz) || !isfinite(dx) || !isfinite(dy) || !isfinite(dz)) return;
    if (fabsf(dx) < EPS) dx = copysignf(EPS, dx);
    if (fabsf(dy) < EPS) dy = copysignf(EPS, dy);
    if (fabsf(dz) < EPS) dz = copysignf(EPS, dz);
    const float tcx = -1.0f / fabsf(dx), tcy = -1.0f / fabsf(dy), tcz = -1.0f / fabsf(dz);
    float tox = px * tcx, toy = py * tcy, toz = pz * tcz;
    uint32_t mirror = 7u;
    if (dx > 0.0f) { mirror ^= 1u; tox = 3.0f * tcx - tox; }
    if (dy > 0.0f) { mirror ^= 2u; toy = 3.0f * tcy - toy; }
    if (dz > 0.0f) { mirror ^= 4u; toz = 3.0f * tcz - toz; }
    float t_min = smax(2.0f * tcx - tox, smax(2.0f * tcy - toy, 2.0f * tcz - toz));
    t_min = smax(0.0f, t_min);
    const float T1 = smin(1.0f, smin(tcx - tox, smin(tcy - toy, tcz - toz)));
    uint32_t parent = 0u, co = 0u, scale = 22u, complexity = 0u, normal = 0u;
    float posx = 1.0f, posy = 1.0f, posz = 1.0f, sf = 0.5f;
    if (1.5f * tcx - tox > t_min) { co ^= 1u; posx = 1.5f; }
    if (1.5f * tcy - toy > t_min) { co ^= 2u; posy = 1.5f; }
    if (1.5f * tcz - toz > t_min) { co ^= 4u; posz = 1.5f; }
    uint32_t rows[24];
    memset(rows, 0, sizeof(rows));

    float start_m[3] = { 0.0f, 0.0f, 0.0f };
    uint32_t start_mb = 0xffffffffu;
    orc_walk_state lens_prev;
    float lens_q[3];
    if (cam && lens_of) {
        /* the lens point is on one side of a centre plane the camera sits on by POSITION: it starts next to 1.5 - 1 ulp or
         * next to 1.5 on that side's path, like a ray next to a hit there (a point on the plane, or within the margin of it,
         * fails the margin test below and starts at the root) */
        const uint32_t side = (px > 1.5f ? 1u : 0u) | (py > 1.5f ? 2u : 0u) | (pz > 1.5f ? 4u : 0u);
        memcpy(lens_prev.rows, cam->rows[side], sizeof(lens_prev.rows));
        lens_prev.scale = cam->scale[side];
        lens_prev.valid = cam->scale[side] < 22u;
        for (int i = 0; i < 3; i++)
            lens_q[i] = ((cam->onplane >> i) & 1u) ? (((side >> i) & 1u) ? 1.5f : u2f(f2u(1.5f) - 1u)) : lens_of[i];
        prev = &lens_prev; prev_pos = lens_q; cam = 0;
    }
    if (cam) {
        const uint32_t v = mirror ^ 7u;
        const int s0 = (int)cam->scale[v];
        if (s0 < 22) {
            float m[3] = { (mirror & 1u) ? px : 3.0f - px, (mirror & 2u) ? py : 3.0f - py, (mirror & 4u) ? pz : 3.0f - pz };
            for (int i = 0; i < 3; i++) if ((cam->onplane >> i) & 1u) m[i] = u2f(f2u(1.5f) - 1u);
            const uint32_t keep = 0xffffffffu << s0;
            posx = u2f(f2u(m[0]) & keep); posy = u2f(f2u(m[1]) & keep); posz = u2f(f2u(m[2]) & keep);
            co = ((f2u(m[0]) >> s0) & 1u) | (((f2u(m[1]) >> s0) & 1u) << 1) | (((f2u(m[2]) >> s0) & 1u) << 2);
            scale = (uint32_t)s0;
            sf = u2f((uint32_t)(s0 - SVO_MAX + 127) << 23);
            memcpy(rows, cam->rows[v], sizeof(rows));
            parent = rows[s0];
            complexity = 22u - (uint32_t)s0;
            *skipped = complexity;
            t_min = 0.0f;
            /* (a camera path already ends where the descends by position end: nothing below it) */
        }
    }
    int may_start_below = !cam && prev && prev->valid;
#ifndef ORC_MODEL_NO_DIRECTION_GUARD   /* (tests of the test: without the guard the model must differ from the oracle) */
    if (coef != 0.0f || bias != 0.0f)
        may_start_below = may_start_below && bias == 0.0f && coef <= 0.5f && smax(fabsf(dx), smax(fabsf(dy), fabsf(dz))) > 0.55f;
#endif
    if (may_start_below) {
        const uint32_t ob[3] = { f2u(px), f2u(py), f2u(pz) };
        const uint32_t hb[3] = { f2u(prev_pos[0]), f2u(prev_pos[1]), f2u(prev_pos[2]) };
        uint32_t db = 0u, mb = 0u;
        for (int i = 0; i < 3; i++) { db |= ob[i] ^ hb[i]; mb |= (ob[i] - orc_model_margin) ^ (ob[i] + orc_model_margin); }
        int s0 = highbit(db | (mb + mb));
        if (s0 < (int)prev->scale) s0 = (int)prev->scale;
        if (s0 < 22) {
            /* the state of a pop to scale s0 with pos = the mirrored origin (lsvo.hpp:133-143) */
            const float mx = (mirror & 1u) ? px : 3.0f - px, my = (mirror & 2u) ? py : 3.0f - py, mz = (mirror & 4u) ? pz : 3.0f - pz;
            const uint32_t keep = 0xffffffffu << s0;
            posx = u2f(f2u(mx) & keep); posy = u2f(f2u(my) & keep); posz = u2f(f2u(mz) & keep);
            co = ((f2u(mx) >> s0) & 1u) | (((f2u(my) >> s0) & 1u) << 1) | (((f2u(mz) >> s0) & 1u) << 2);
            scale = (uint32_t)s0;
            sf = u2f((uint32_t)(s0 - SVO_MAX + 127) << 23);
            memcpy(rows, prev->rows, sizeof(rows));
            parent = rows[s0];
            complexity = 22u - (uint32_t)s0;
            *skipped = complexity;
            start_m[0] = mx; start_m[1] = my; start_m[2] = mz; start_mb = mb;
            t_min = 0.0f;      /* as the kernel sets it (ray_setup_below): the entry into the cube lies behind such an origin */
        }
    }

    /* ... and from there on down by position while the next iterations are descends as well: the child around the origin is
     * present and no leaf (node data), and the origin keeps the margin from the planes of the next scale.  (The common parent
     * of origin and hit is only where their PATHS part; the cell around the origin usually exists a level or two deeper.) */
    if (orc_model_descend_by_position && *skipped && start_mb != 0xffffffffu) {
        int lowest = highbit(start_mb + start_mb);
        while ((int)scale > lowest && (int)scale > 23 - (int)depth) {
            const orc_lnode nd = raw[parent];
            const uint32_t cs = co ^ mirror;
            if (!((nd.child_mask >> cs) & 1u) || ((nd.leaf_mask >> cs) & 1u)) break;
            rows[scale] = parent;
            parent += nd.child_offset + cs;
            --scale;
            sf *= 0.5f;
            const uint32_t keep = 0xffffffffu << scale;
            posx = u2f(f2u(start_m[0]) & keep); posy = u2f(f2u(start_m[1]) & keep); posz = u2f(f2u(start_m[2]) & keep);
            co = ((f2u(start_m[0]) >> scale) & 1u) | (((f2u(start_m[1]) >> scale) & 1u) << 1) | (((f2u(start_m[2]) >> scale) & 1u) << 2);
            ++complexity;
            ++*skipped;
        }
    }

    uint32_t kind = 0u, shift = 0u;
    while (scale < 23u) {
        if (complexity >= 65536u) break;
        ++complexity;
        const orc_lnode nd = raw[parent];
        const float cx = posx * tcx - tox, cy = posy * tcy - toy, cz = posz * tcz - toz;
        const float tc = smin(cx, smin(cy, cz));
        const uint32_t cs = co ^ mirror;
        const int child = (nd.child_mask >> cs) & 1;
        int inside;
        if (coef != 0.0f || bias != 0.0f) {
            inside = child && t_min <= T1;
            if (inside && tc * coef + bias >= sf) { kind = 2u; shift = cs; break; }
            inside = inside && t_min <= tc;
        } else {
            inside = child && t_min <= smin(1.0f, tc);
        }
        if (inside) {
            if ((nd.leaf_mask >> cs) & 1) { kind = 1u; shift = cs; break; }
            const float half = sf * 0.5f;
            const float thx = half * tcx + cx, thy = half * tcy + cy, thz = half * tcz + cz;
            rows[scale] = parent;
            parent += nd.child_offset + cs;
            co = 0u; --scale; sf = half;
            if (thx > t_min) { co ^= 1u; posx += sf; }
            if (thy > t_min) { co ^= 2u; posy += sf; }
            if (thz > t_min) { co ^= 4u; posz += sf; }
            continue;
        }
        uint32_t step = 0u;
        const uint32_t opx = f2u(posx), opy = f2u(posy), opz = f2u(posz);
        if (cx <= tc) { step ^= 1u; posx -= sf; }
        if (cy <= tc) { step ^= 2u; posy -= sf; }
        if (cz <= tc) { step ^= 4u; posz -= sf; }
        t_min = tc;
        co ^= step;
        normal = step;
        if (co & step) {
            const uint32_t ipx = f2u(posx), ipy = f2u(posy), ipz = f2u(posz);
            const uint32_t diff = (ipx ^ opx) | (ipy ^ opy) | (ipz ^ opz);
            const int s = highbit(diff);
            if (s < 0 || s >= SVO_MAX) { scale = 23u; break; }
            scale = (uint32_t)s;
            sf = u2f((uint32_t)(s - SVO_MAX + 127) << 23);
            parent = rows[scale];
            const uint32_t keep = 0xffffffffu << scale;
            posx = u2f(ipx & keep); posy = u2f(ipy & keep); posz = u2f(ipz & keep);
            co = ((ipx >> scale) & 1u) | (((ipy >> scale) & 1u) << 1) | (((ipz >> scale) & 1u) << 2);
        }
    }
    out->complexity = complexity;
    if (kind) {
        out->normal[0] = -gsign(dx) * (float)(normal & 1u);
        out->normal[1] = -gsign(dy) * (float)(normal & 2u);
        out->normal[2] = -gsign(dz) * (float)(normal & 4u);
        if ((mirror & 1u) == 0u) posx = 3.0f - sf - posx;
        if ((mirror & 2u) == 0u) posy = 3.0f - sf - posy;
        if ((mirror & 4u) == 0u) posz = 3.0f - sf - posz;
        out->distance = t_min;
        out->position[0] = smin(smax(px + t_min * dx, posx + EPS), posx + sf - EPS);
        out->position[1] = smin(smax(py + t_min * dy, posy + EPS), posy + sf - EPS);
        out->position[2] = smin(smax(pz + t_min * dz, posz + EPS), posz + sf - EPS);
        const float S = (float)(1u << depth);
        if (out->normal[0]) { out->voxel_coord[0] = frac_(out->position[2] * S); out->voxel_coord[1] = frac_(out->position[1] * S); }
        else if (out->normal[1]) { out->voxel_coord[0] = frac_(out->position[0] * S); out->voxel_coord[1] = frac_(out->position[2] * S); }
        else if (out->normal[2]) { out->voxel_coord[0] = frac_(out->position[0] * S); out->voxel_coord[1] = frac_(out->position[1] * S); }
        out->hit = kind | (shift << 8) | ((scale & 0xffu) << 16);
        out->node = parent;
        rows[scale] = parent;
        memcpy(st->rows, rows, sizeof(rows));
        st->scale = scale;
        st->valid = 1u;
    }
}

void orc_model_cast(const orc_lnode *raw, uint32_t depth, const float org[3], const float dir[3], float coef, float bias,
                    const orc_walk_state *prev, const float prev_pos[3], orc_hit *out, orc_walk_state *st, uint32_t *skipped)
{
    model_cast(raw, depth, org, dir, coef, bias, prev, prev_pos, 0, 0, out, st, skipped);
}

/* n camera rays of a camera with a lens: origins next to the camera position `cam` (camera_controller.hpp:41,46) */
void orc_model_cast_lens_rays(const orc_lnode *raw, uint32_t depth, uint64_t n, const float cam[3], const float *org, const float *dir,
                              orc_hit *out, orc_walk_state *st, uint32_t *skipped, uint32_t *scales, int threads)
{
    orc_camera_paths paths;
    orc_model_camera_paths(raw, depth, cam, &paths);
    if (scales) memcpy(scales, paths.scale, sizeof(paths.scale));
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++)
        model_cast(raw, depth, org + 3 * i, dir + 3 * i, 0.0f, 0.0f, 0, 0, &paths, cam, out + i, st + i, skipped + i);
}

/* n camera rays from one position (a pinhole camera's primary rays, main.cpp:149) */
void orc_model_cast_camera_rays(const orc_lnode *raw, uint32_t depth, uint64_t n, const float cam[3], const float *dir,
                                orc_hit *out, orc_walk_state *st, uint32_t *skipped, uint32_t *scales, int threads)
{
    orc_camera_paths paths;
    orc_model_camera_paths(raw, depth, cam, &paths);
    if (scales) memcpy(scales, paths.scale, sizeof(paths.scale));
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++)
        model_cast(raw, depth, cam, dir + 3 * i, 0.0f, 0.0f, 0, 0, &paths, 0, out + i, st + i, skipped + i);
}

/* n independent chains: ray i starts next to hit i of the previous call (prev / prev_pos NULL: from the root) */
void orc_model_cast_rays(const orc_lnode *raw, uint32_t depth, uint64_t n, const float *org, const float *dir,
                         float coef, float bias, const orc_walk_state *prev, const float *prev_pos,
                         orc_hit *out, orc_walk_state *st, uint32_t *skipped, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++)
        orc_model_cast(raw, depth, org + 3 * i, dir + 3 * i, coef, bias, prev ? prev + i : 0, prev_pos ? prev_pos + 3 * i : 0,
                       out + i, st + i, skipped + i);
}
