/*
 * vrc_oracle.c -- CPU restatement ("oracle") of the CpuVoxelRaycaster voxel
 * ray-traversal hot path.  TEST INFRASTRUCTURE ONLY -- see vrc_oracle.h for the
 * usage rule and the parity status ("parity unpinned" by reference execution;
 * pinned against SURVEY.md App. B statistics of the unmodified reference and
 * against terrain heights from the real FastNoise build in oracle/_ref).
 *
 * Plain C99, scalar, written from the reference's behaviour, not its text.
 * All citations are file:line under /root/reference.
 *
 * Arithmetic rules that parity depends on:
 *   - build with -ffp-contract=off: every a*b+c below is a rounded multiply
 *     followed by a rounded add;
 *   - std::max(a,b) is (a<b)?b:a and std::min(a,b) is (b<a)?b:a (argument
 *     order matters for NaN and for +-0): smax/smin below;
 *   - glm::dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;
 *     glm::normalize(v) = v * (1.0f / sqrtf(dot(v,v)))  (stock scalar GLM).
 */
#include "vrc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* scalar helpers                                                            */
/* ------------------------------------------------------------------------- */

static inline float smax(float a, float b) { return (a < b) ? b : a; } /* std::max */
static inline float smin(float a, float b) { return (b < a) ? b : a; } /* std::min */

/* utils.cpp:109-118 */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* utils.cpp:60-64: std::modf fractional part (sign-preserving). */
float orc_frac(float f)
{
    float whole;
    return modff(f, &whole);
}

/* glm::sign: -1, 0, +1; sign(+-0) = 0. */
static inline float gsign(float x) { return (float)((0.0f < x) - (x < 0.0f)); }

typedef struct { float x, y, z; } v3;

static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 vnormalize(v3 v)
{
    const float inv = 1.0f / sqrtf(vdot(v, v));
    v3 r = { v.x * inv, v.y * inv, v.z * inv };
    return r;
}

/* ------------------------------------------------------------------------- */
/* T2: pointer octree (svo.hpp:7-25,72-114) and compile (lsvo_utils.cpp:4-49) */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint32_t sub[8]; /* index = x*4 + y*2 + z, i.e. Node::sub[x][y][z] (svo.hpp:22); 0 = null */
    uint8_t leaf;
} onode;

struct orc_svo {
    uint32_t depth;
    onode *pool;
    uint64_t count, cap;
};

static uint32_t svo_alloc(orc_svo *s)
{
    if (s->count == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 1024;
        s->pool = (onode *)realloc(s->pool, s->cap * sizeof(onode));
        if (!s->pool) abort();
    }
    memset(&s->pool[s->count], 0, sizeof(onode));
    return (uint32_t)(s->count++);
}

orc_svo *orc_svo_new(uint32_t depth)
{
    orc_svo *s = (orc_svo *)calloc(1, sizeof(orc_svo));
    s->depth = depth;
    svo_alloc(s); /* slot 0 reserved as "null" */
    svo_alloc(s); /* slot 1 = m_root (svo.hpp:37) */
    return s;
}

void orc_svo_free(orc_svo *s)
{
    if (!s) return;
    free(s->pool);
    free(s);
}

void orc_free(void *p) { free(p); }

/* svo.hpp:72-76 + rec_setCell :91-114 (iterative form of the same descent:
 * always down to size 1, creating missing children, marking the last one leaf). */
void orc_svo_set_cell(orc_svo *s, uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t size = 1u << s->depth;
    uint32_t node = 1;
    /* out-of-volume coordinates index Node::sub out of bounds in the reference
     * (UB; happens for the terrain generator at S <= 128 where y + S/2 >= S):
     * dropped here. */
    if (x >= size || y >= size || z >= size) return;
    while (size != 1) {
        const uint32_t sub_size = size / 2;
        const uint32_t cx = x / sub_size, cy = y / sub_size, cz = z / sub_size;
        const uint32_t k = cx * 4 + cy * 2 + cz;
        if (!s->pool[node].sub[k]) {
            const uint32_t fresh = svo_alloc(s); /* may move the pool */
            s->pool[node].sub[k] = fresh;
        }
        node = s->pool[node].sub[k];
        x -= cx * sub_size; y -= cy * sub_size; z -= cz * sub_size;
        size = sub_size;
    }
    s->pool[node].leaf = 1;
}

/* main.cpp:63-76 */
void orc_svo_fill_terrain(orc_svo *s, const int32_t *height)
{
    const int32_t size = 1 << s->depth;
    for (int32_t x = 0; x < size; x++) {
        for (int32_t z = 0; z < size; z++) {
            const int32_t max_height = size;
            const int32_t h = height[(int64_t)x * size + z];
            const int32_t ground_level = 16;
            int32_t lim = h < max_height ? h : max_height; /* std::min(max_height, height) */
            lim = ground_level < lim ? lim : ground_level;  /* std::max(ground_level, ...) */
            for (int32_t y = 1; y < lim; ++y)
                orc_svo_set_cell(s, (uint32_t)x, (uint32_t)(y + size / 2), (uint32_t)z);
        }
    }
}

typedef struct {
    orc_lnode *data;
    uint64_t count, cap;
} lvec;

static void lvec_push(lvec *v)
{
    if (v->count == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->data = (orc_lnode *)realloc(v->data, v->cap * sizeof(orc_lnode));
        if (!v->data) abort();
    }
    /* LNode() : lsvo_utils.hpp:7-12 */
    orc_lnode n = { 1u, 0u, 0u, 0u, 0u };
    v->data[v->count++] = n;
}

/* lsvo_utils.cpp:4-49 */
static void compile_rec(const orc_svo *s, uint32_t node, lvec *data, uint32_t node_index)
{
    if (!node) return;
    const uint32_t child_pos = (uint32_t)data->count;
    data->data[node_index].child_offset = child_pos - node_index; /* :8-10 */
    int empty = 1;
    for (int k = 0; k < 8; k++) if (s->pool[node].sub[k]) empty = 0; /* :12-22 */
    if (empty) return;
    for (int i = 0; i < 8; i++) lvec_push(data); /* :25-27 */
    /* :29-47 -- x outer, y middle, z inner; slot = z*4 + y*2 + x */
    for (uint32_t x = 0; x < 2; ++x)
        for (uint32_t y = 0; y < 2; ++y)
            for (uint32_t z = 0; z < 2; ++z) {
                const uint32_t sub = s->pool[node].sub[x * 4 + y * 2 + z];
                if (!sub) continue;
                const uint32_t sub_index = z * 4 + y * 2 + x;
                data->data[node_index].child_mask |= (uint8_t)(1u << sub_index);
                if (!s->pool[sub].leaf)
                    compile_rec(s, sub, data, child_pos + sub_index);
                else
                    data->data[node_index].leaf_mask |= (uint8_t)(1u << sub_index);
            }
}

/* lsvo_utils.hpp:45-55 */
uint64_t orc_svo_compile(const orc_svo *s, orc_lnode **out)
{
    lvec v = { 0, 0, 0 };
    lvec_push(&v);
    compile_rec(s, 1, &v, 0);
    *out = v.data;
    return v.count;
}

/* ------------------------------------------------------------------------- */
/* N4: terrain heights = FastNoise SimplexFractal, as src/main.cpp:61-69 calls it */
/* (lib/fastnoise/FastNoise.cpp, v0.4.1, FN_DECIMAL = float).  This part of the  */
/* oracle IS pinned by reference execution: tests compare it with heights from   */
/* the real FastNoise compiled from /root/reference (oracle/_ref).               */
/* ------------------------------------------------------------------------- */

/* std::mt19937_64 (ISO C++ [rand.predef]): FastNoise::SetSeed, FastNoise.cpp:197-215 */
typedef struct { uint64_t mt[312]; int idx; } mt64;
static void mt64_seed(mt64 *g, uint64_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 312; i++) g->mt[i] = 6364136223846793005ULL * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (uint64_t)i;
    g->idx = 312;
}
static uint64_t mt64_next(mt64 *g)
{
    if (g->idx >= 312) {
        for (int i = 0; i < 312; i++) {
            const uint64_t x = (g->mt[i] & 0xFFFFFFFF80000000ULL) | (g->mt[(i + 1) % 312] & 0x7FFFFFFFULL);
            g->mt[i] = g->mt[(i + 156) % 312] ^ (x >> 1) ^ ((x & 1ULL) ? 0xB5026F5AA96619E9ULL : 0ULL);
        }
        g->idx = 0;
    }
    uint64_t y = g->mt[g->idx++];
    y ^= (y >> 29) & 0x5555555555555555ULL;
    y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
    y ^= (y << 37) & 0xFFF7EEE000000000ULL;
    y ^= (y >> 43);
    return y;
}

typedef struct { uint8_t perm[512], perm12[512]; float F2, G2, bounding; } fnoise;

static void fnoise_init(fnoise *n, int seed)
{
    mt64 gen;
    mt64_seed(&gen, (uint64_t)(int64_t)seed);               /* std::mt19937_64 gen(seed), :201 */
    for (int i = 0; i < 256; i++) n->perm[i] = (uint8_t)i;
    for (int j = 0; j < 256; j++) {                         /* :206-214 */
        const int rng = (int)(mt64_next(&gen) % (uint64_t)(256 - j));
        const int k = rng + j;
        const int l = n->perm[j];
        n->perm[j] = n->perm[j + 256] = n->perm[k];
        n->perm[k] = (uint8_t)l;
        n->perm12[j] = n->perm12[j + 256] = (uint8_t)(n->perm[j] % 12);
    }
    const float SQRT3 = 1.7320508075688772935274463415059f; /* :1270-1272 */
    n->F2 = 0.5f * (SQRT3 - 1.0f);
    n->G2 = (3.0f - SQRT3) / 6.0f;
    float amp = 0.5f, ampFractal = 1.0f;                    /* CalculateFractalBounding :217-227, gain .5, 3 octaves */
    for (int i = 1; i < 3; i++) { ampFractal += amp; amp *= 0.5f; }
    n->bounding = 1.0f / ampFractal;
}

static const float FN_GRAD_X[12] = { 1, -1, 1, -1, 1, -1, 1, -1, 0, 0, 0, 0 };   /* FastNoise.cpp:37-48 */
static const float FN_GRAD_Y[12] = { 1, 1, -1, -1, 0, 0, 0, 0, 1, -1, 1, -1 };

static inline int fn_fast_floor(float f) { return f >= 0 ? (int)f : (int)f - 1; } /* :184 */

static float fn_grad2(const fnoise *n, uint8_t offset, int x, int y, float xd, float yd)  /* :244-247,:312-317 */
{
    const uint8_t lut = n->perm12[(x & 0xff) + n->perm[(y & 0xff) + offset]];
    return xd * FN_GRAD_X[lut] + yd * FN_GRAD_Y[lut];
}

static float fn_simplex2(const fnoise *n, uint8_t offset, float x, float y)      /* :1274-1335 */
{
    float t = (x + y) * n->F2;
    const int i = fn_fast_floor(x + t), j = fn_fast_floor(y + t);
    t = (float)(i + j) * n->G2;
    const float X0 = (float)i - t, Y0 = (float)j - t;
    const float x0 = x - X0, y0 = y - Y0;
    int i1, j1;
    if (x0 > y0) { i1 = 1; j1 = 0; } else { i1 = 0; j1 = 1; }
    const float x1 = x0 - (float)i1 + n->G2, y1 = y0 - (float)j1 + n->G2;
    const float two_g2 = 2.0f * n->G2;
    const float x2 = x0 - 1.0f + two_g2, y2 = y0 - 1.0f + two_g2;
    float n0, n1, n2;
    t = 0.5f - x0 * x0 - y0 * y0;
    if (t < 0) n0 = 0; else { t *= t; n0 = t * t * fn_grad2(n, offset, i, j, x0, y0); }
    t = 0.5f - x1 * x1 - y1 * y1;
    if (t < 0) n1 = 0; else { t *= t; n1 = t * t * fn_grad2(n, offset, i + i1, j + j1, x1, y1); }
    t = 0.5f - x2 * x2 - y2 * y2;
    if (t < 0) n2 = 0; else { t *= t; n2 = t * t * fn_grad2(n, offset, i + 1, j + 1, x2, y2); }
    return 70.0f * (n0 + n1 + n2);
}

/* GetNoise -> GetSimplexFractal -> SingleSimplexFractalFBM (:410-447,:1175-1207): frequency .01, lacunarity 2 */
static float fn_get_noise(const fnoise *n, float x, float y)
{
    x *= 0.01f; y *= 0.01f;
    float sum = fn_simplex2(n, n->perm[0], x, y);
    float amp = 1.0f;
    for (int i = 1; i < 3; i++) {
        x *= 2.0f; y *= 2.0f;
        amp *= 0.5f;
        sum += fn_simplex2(n, n->perm[i], x, y) * amp;
    }
    return sum * n->bounding;
}

/* main.cpp:61-69: FastNoise myNoise (seed 1337); height = int32(64 * GetNoise(.75 x, .75 z) + 32) */
void orc_terrain_heights(int32_t size, int32_t seed, int32_t *height)
{
    fnoise n;
    fnoise_init(&n, seed);
    for (uint32_t x = 0; x < (uint32_t)size; x++)
        for (uint32_t z = 0; z < (uint32_t)size; z++)
            height[(int64_t)x * size + z] = (int32_t)(64.0f * fn_get_noise(&n, (float)(0.75f * (float)x), (float)(0.75f * (float)z)) + 32.0f);
}

/* ------------------------------------------------------------------------- */
/* T1: LSVO<N>::castRay (lsvo.hpp:33-172)                                    */
/* ------------------------------------------------------------------------- */

#define ORC_MAX_DEPTH 16
#define ORC_MAX_ITERATIONS 65536u

/* Analysis hook (tests/tools/path_schedule.py): when set, the walk appends one byte per loop iteration -- 0 descend,
 * 1 advance, 2 advance + pop, 3 the iteration that ends in a leaf / LOD hit -- up to orc_optrace_cap bytes. */
static __thread uint8_t *orc_optrace;
static __thread uint32_t orc_optrace_cap;
#define ORC_OP(code) do { if (orc_optrace && result->complexity <= orc_optrace_cap) orc_optrace[result->complexity - 1u] = (code); } while (0)

void orc_cast_ray(const orc_lnode *raw_data, uint32_t depth,
                  const float org[3], const float dir[3],
                  float ray_size_coef, float ray_size_bias, orc_hit *result)
{
    memset(result, 0, sizeof(*result));
    /* :37-40 */
    const int SVO_MAX_DEPTH = 23;
    const int MAX_DEPTH = (int)depth;
    const int DEPTH_OFFSET = SVO_MAX_DEPTH - MAX_DEPTH;
    const float SVO_SIZE = (float)(1 << MAX_DEPTH);
    const float EPS = 1.0f / (float)(1 << 23);
    /* :42 -- the reference leaves the stack uninitialised; slot MAX_DEPTH is
     * read (never written) when a ray leaves the root. */
    struct { uint32_t parent_index; float t_max; } stack[ORC_MAX_DEPTH + 1];
    memset(stack, 0, sizeof(stack));

    const float px = org[0], py = org[1], pz = org[2];
    float dx = dir[0], dy = dir[1], dz = dir[2];
    /* Non-finite input: with a NaN component no comparison at :116-118 holds, the
     * walk makes no progress and the reference never returns.  Defined here (and
     * in the HIP kernels) as an immediate miss with complexity 0.  This is what
     * RayCaster reaches when a ray starts inside a solid voxel: the hit normal
     * is all zero (App. A.10) and normalize() of the GI direction is NaN. */
    if (!isfinite(px) || !isfinite(py) || !isfinite(pz) || !isfinite(dx) || !isfinite(dy) || !isfinite(dz))
        return;
    /* :44-46 */
    if (fabsf(dx) < EPS) dx = copysignf(EPS, dx);
    if (fabsf(dy) < EPS) dy = copysignf(EPS, dy);
    if (fabsf(dz) < EPS) dz = copysignf(EPS, dz);
    /* :47-48 */
    const float tcx = -1.0f / fabsf(dx), tcy = -1.0f / fabsf(dy), tcz = -1.0f / fabsf(dz);
    float tox = px * tcx, toy = py * tcy, toz = pz * tcz;
    /* :49-52 */
    uint8_t mirror_mask = 7u;
    if (dx > 0.0f) { mirror_mask ^= 1u; tox = 3.0f * tcx - tox; }
    if (dy > 0.0f) { mirror_mask ^= 2u; toy = 3.0f * tcy - toy; }
    if (dz > 0.0f) { mirror_mask ^= 4u; toz = 3.0f * tcz - toz; }
    /* :54-58 */
    float t_min = smax(2.0f * tcx - tox, smax(2.0f * tcy - toy, 2.0f * tcz - toz));
    float t_max = smin(tcx - tox, smin(tcy - toy, tcz - toz));
    float h = t_max;
    t_min = smax(0.0f, t_min);
    t_max = smin(1.0f, t_max);
    /* :60-64 */
    uint32_t parent_id = 0u;
    uint8_t child_offset = 0u;
    int8_t scale = (int8_t)(SVO_MAX_DEPTH - 1);
    float posx = 1.0f, posy = 1.0f, posz = 1.0f;
    float scale_f = 0.5f;
    /* :66-68 */
    if (1.5f * tcx - tox > t_min) { child_offset ^= 1u; posx = 1.5f; }
    if (1.5f * tcy - toy > t_min) { child_offset ^= 2u; posy = 1.5f; }
    if (1.5f * tcz - toz > t_min) { child_offset ^= 4u; posz = 1.5f; }
    uint8_t normal = 0u;
    uint32_t hit_kind = 0u, hit_shift = 0u;

    /* :72 */
    while (scale < SVO_MAX_DEPTH && scale > MAX_DEPTH) {
        /* safety net shared with the HIP kernels: no finite ray needs anywhere
         * near this many iterations (max observed ~150 at 512^3) */
        if (result->complexity >= ORC_MAX_ITERATIONS) break;
        ++result->complexity;                                 /* :73 */
        const orc_lnode parent_ref = raw_data[parent_id];     /* :74 */
        /* :76-77 */
        const float tcornx = posx * tcx - tox, tcorny = posy * tcy - toy, tcornz = posz * tcz - toz;
        const float tc_max = smin(tcornx, smin(tcorny, tcornz));
        /* :79-81 */
        const uint8_t child_shift = child_offset ^ mirror_mask;
        const uint8_t child_mask = (uint8_t)(parent_ref.child_mask >> child_shift);
        if ((child_mask & 1u) && t_min <= t_max) {
            /* :82-85 LOD cut */
            if (tc_max * ray_size_coef + ray_size_bias >= scale_f) {
                hit_kind = 2u; hit_shift = child_shift;
                ORC_OP(3);
                break;
            }
            /* :86-88 */
            const float tv_max = smin(t_max, tc_max);
            const float half = scale_f * 0.5f;
            const float thx = half * tcx + tcornx, thy = half * tcy + tcorny, thz = half * tcz + tcornz;
            if (t_min <= tv_max) {                            /* :89 */
                const uint8_t leaf_mask = (uint8_t)(parent_ref.leaf_mask >> child_shift);
                if (leaf_mask & 1u) {                         /* :92-95 */
                    hit_kind = 1u; hit_shift = child_shift;
                    ORC_OP(3);
                    break;
                }
                if (tc_max < h) {                             /* :97-100 */
                    stack[scale - DEPTH_OFFSET].parent_index = parent_id;
                    stack[scale - DEPTH_OFFSET].t_max = t_max;
                }
                h = tc_max;                                   /* :101 */
                parent_id += parent_ref.child_offset + child_shift; /* :103 */
                child_offset = 0u;
                --scale;
                scale_f = half;
                if (thx > t_min) { child_offset ^= 1u; posx += scale_f; } /* :107-109 */
                if (thy > t_min) { child_offset ^= 2u; posy += scale_f; }
                if (thz > t_min) { child_offset ^= 4u; posz += scale_f; }
                t_max = tv_max;                               /* :110 */
                ORC_OP(0);
                continue;
            }
        }
        /* :115-118 advance */
        uint32_t step_mask = 0u;
        if (tcornx <= tc_max) { step_mask ^= 1u; posx -= scale_f; }
        if (tcorny <= tc_max) { step_mask ^= 2u; posy -= scale_f; }
        if (tcornz <= tc_max) { step_mask ^= 4u; posz -= scale_f; }
        t_min = tc_max;                                       /* :120-122 */
        child_offset ^= (uint8_t)step_mask;
        normal = (uint8_t)step_mask;
        ORC_OP((child_offset & step_mask) ? 2 : 1);
        if (child_offset & step_mask) {                       /* :124 pop */
            uint32_t differing_bits = 0u;
            const int32_t ipx = (int32_t)f2u(posx), ipy = (int32_t)f2u(posy), ipz = (int32_t)f2u(posz);
            if (step_mask & 1u) differing_bits |= (uint32_t)ipx ^ f2u(posx + scale_f);
            if (step_mask & 2u) differing_bits |= (uint32_t)ipy ^ f2u(posy + scale_f);
            if (step_mask & 4u) differing_bits |= (uint32_t)ipz ^ f2u(posz + scale_f);
            /* :132 -- u32 arithmetic, truncated into int8_t */
            scale = (int8_t)(uint8_t)((f2u((float)differing_bits) >> 23) - 127u);
            if (differing_bits == 0u) break; /* unreachable; UB in the reference */
            /* :133 */
            scale_f = u2f(((uint32_t)((int)scale - SVO_MAX_DEPTH) + 127u) << 23);
            /* :134-136 */
            parent_id = stack[scale - DEPTH_OFFSET].parent_index;
            t_max = stack[scale - DEPTH_OFFSET].t_max;
            /* :137-143 */
            const uint32_t shx = (uint32_t)(ipx >> scale), shy = (uint32_t)(ipy >> scale), shz = (uint32_t)(ipz >> scale);
            posx = u2f(shx << scale); posy = u2f(shy << scale); posz = u2f(shz << scale);
            child_offset = (uint8_t)((shx & 1u) | ((shy & 1u) << 1u) | ((shz & 1u) << 2u));
            h = 0.0f;                                         /* :144 */
        }
    }

    if (hit_kind) {                                           /* :148 */
        /* :149 -- non-unit normal (1,2,4), signed zeros from -sign(d) * 0 */
        result->normal[0] = -gsign(dx) * (float)(normal & 1u);
        result->normal[1] = -gsign(dy) * (float)(normal & 2u);
        result->normal[2] = -gsign(dz) * (float)(normal & 4u);
        /* :151-153 */
        if ((mirror_mask & 1) == 0) posx = 3.0f - scale_f - posx;
        if ((mirror_mask & 2) == 0) posy = 3.0f - scale_f - posy;
        if ((mirror_mask & 4) == 0) posz = 3.0f - scale_f - posz;
        result->distance = t_min;                             /* :155 */
        /* :156-158 */
        result->position[0] = smin(smax(px + t_min * dx, posx + EPS), posx + scale_f - EPS);
        result->position[1] = smin(smax(py + t_min * dy, posy + EPS), posy + scale_f - EPS);
        result->position[2] = smin(smax(pz + t_min * dz, posz + EPS), posz + scale_f - EPS);
        /* :160-168 (voxel_coord stays unset -- 0 here -- when the normal is all zero) */
        if (result->normal[0]) {
            result->voxel_coord[0] = orc_frac(result->position[2] * SVO_SIZE);
            result->voxel_coord[1] = orc_frac(result->position[1] * SVO_SIZE);
        } else if (result->normal[1]) {
            result->voxel_coord[0] = orc_frac(result->position[0] * SVO_SIZE);
            result->voxel_coord[1] = orc_frac(result->position[2] * SVO_SIZE);
        } else if (result->normal[2]) {
            result->voxel_coord[0] = orc_frac(result->position[0] * SVO_SIZE);
            result->voxel_coord[1] = orc_frac(result->position[1] * SVO_SIZE);
        }
        result->hit = hit_kind | (hit_shift << 8) | ((uint32_t)(uint8_t)scale << 16);
        result->node = parent_id;
    }
}

void orc_cast_rays(const orc_lnode *nodes, uint32_t depth, uint64_t n,
                   const float *org_xyz, const float *dir_xyz,
                   const float *coef, const float *bias, orc_hit *out, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++)
        orc_cast_ray(nodes, depth, org_xyz + 3 * i, dir_xyz + 3 * i,
                     coef ? coef[i] : 0.0f, bias ? bias[i] : 0.0f, out + i);
}

/* ------------------------------------------------------------------------- */
/* G1: Grid3D<X,Y,Z>::castRay (grid_3d.hpp:36-132)                           */
/* ------------------------------------------------------------------------- */

void orc_grid_cast_ray(const uint8_t *cells, int32_t X, int32_t Y, int32_t Z,
                       const float org[3], const float dir[3], orc_hit *point)
{
    memset(point, 0, sizeof(*point));
    /* :42-44 */
    const float t_dx = fabsf(1.0f / dir[0]), t_dy = fabsf(1.0f / dir[1]), t_dz = fabsf(1.0f / dir[2]);
    /* :48-55 */
    const int32_t step_x = dir[0] < 0 ? -1 : 1, step_y = dir[1] < 0 ? -1 : 1, step_z = dir[2] < 0 ? -1 : 1;
    const int32_t dir_x = step_x > 0 ? 1 : 0, dir_y = step_y > 0 ? 1 : 0, dir_z = step_z > 0 ? 1 : 0;
    /* :58-60 float -> int32 truncation */
    int32_t cell_x = (int32_t)org[0], cell_y = (int32_t)org[1], cell_z = (int32_t)org[2];
    /* :62-64 (int + int) converted to float, minus, divide */
    float t_max_x = ((float)(cell_x + dir_x) - org[0]) / dir[0];
    float t_max_y = ((float)(cell_y + dir_y) - org[1]) / dir[1];
    float t_max_z = ((float)(cell_z + dir_z) - org[2]) / dir[2];
    uint8_t hit_side = 0;
    const uint32_t max_iter = 2048;
    uint32_t iter = 0u;
    /* :70 */
    while (cell_x >= 0 && cell_y >= 0 && cell_z >= 0 && cell_x < X && cell_y < Y && cell_z < Z && iter < max_iter) {
        float t_max_min;
        ++iter;
        if (t_max_x < t_max_y) {                              /* :73-100 */
            if (t_max_x < t_max_z) { t_max_min = t_max_x; t_max_x += t_dx; cell_x += step_x; hit_side = 0; }
            else                   { t_max_min = t_max_z; t_max_z += t_dz; cell_z += step_z; hit_side = 2; }
        } else {
            if (t_max_y < t_max_z) { t_max_min = t_max_y; t_max_y += t_dy; cell_y += step_y; hit_side = 1; }
            else                   { t_max_min = t_max_z; t_max_z += t_dz; cell_z += step_z; hit_side = 2; }
        }
        /* :102-104 -- only the NEW cell is tested, never the start cell */
        if (cell_x >= 0 && cell_y >= 0 && cell_z >= 0 && cell_x < X && cell_y < Y && cell_z < Z) {
            const int64_t idx = ((int64_t)cell_x * Y + cell_y) * Z + cell_z;
            if (cells[idx] != 0) {
                const float hit_x = org[0] + t_max_min * dir[0];
                const float hit_y = org[1] + t_max_min * dir[1];
                const float hit_z = org[2] + t_max_min * dir[2];
                point->hit = 1u | ((uint32_t)hit_side << 8);
                point->node = (uint32_t)idx;
                point->position[0] = hit_x; point->position[1] = hit_y; point->position[2] = hit_z;
                if (hit_side == 0) {                          /* :112-121 */
                    point->normal[0] = (float)(-step_x);
                    point->voxel_coord[0] = 1.0f - orc_frac(hit_z);
                    point->voxel_coord[1] = orc_frac(hit_y);
                } else if (hit_side == 1) {
                    point->normal[1] = (float)(-step_y);
                    point->voxel_coord[0] = orc_frac(hit_x);
                    point->voxel_coord[1] = orc_frac(hit_z);
                } else {
                    point->normal[2] = (float)(-step_z);
                    point->voxel_coord[0] = orc_frac(hit_x);
                    point->voxel_coord[1] = orc_frac(hit_y);
                }
                point->distance = t_max_min;                  /* :123-124 */
                point->complexity = iter;
                break;
            }
        }
    }
}

void orc_grid_cast_rays(const uint8_t *cells, int32_t X, int32_t Y, int32_t Z,
                        uint64_t n, const float *org_xyz, const float *dir_xyz,
                        orc_hit *out, int threads)
{
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads)
    for (int64_t i = 0; i < (int64_t)n; i++)
        orc_grid_cast_ray(cells, X, Y, Z, org_xyz + 3 * i, dir_xyz + 3 * i, out + i);
}

/* ------------------------------------------------------------------------- */
/* U1: RNG                                                                   */
/* ------------------------------------------------------------------------- */

/* utils.cpp:11-25 with `unsigned long` = 64 bit (LP64, what a Linux build of
 * the reference runs).  Global, serial -- rng_mode 1 only, single-threaded. */
static uint64_t xs_x = 123456789u, xs_y = 362436069u, xs_z = 521288629u;
void orc_xorshf96_reset(void) { xs_x = 123456789u; xs_y = 362436069u; xs_z = 521288629u; }
uint64_t orc_xorshf96(void)
{
    uint64_t t;
    xs_x ^= xs_x << 16;
    xs_x ^= xs_x >> 5;
    xs_x ^= xs_x << 1;
    t = xs_x;
    xs_x = xs_y;
    xs_y = xs_z;
    xs_z = t ^ xs_x ^ xs_y;
    return xs_z;
}

/* n consecutive outputs of the serial stream (bulk form for the tests) */
void orc_xorshf96_fill(uint64_t n, uint64_t *out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = orc_xorshf96();
}

/* The parallel replacement for the racy global stream:
 * a counter-based generator keyed by (seed, frame/sample, global pixel, draw#).
 * Integer-only so CPU and GPU agree bit for bit.  Not from the reference. */
static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
uint32_t orc_counter_rand(uint32_t seed, uint32_t frame, uint32_t pixel, uint32_t draw)
{
    uint32_t h = mix32(seed ^ (frame * 0x9E3779B1u));
    h = mix32(h ^ (pixel * 0x85EBCA77u));
    h = mix32(h ^ (draw * 0xC2B2AE3Du) ^ 0x27D4EB2Fu);
    return h;
}
/* utils.cpp:77-81: 100 levels */
float orc_get_rand_counter(uint32_t k, float mn, float mx)
{
    const float rand_val = (float)(k % 100u) / 100.0f;
    return mn + (mx - mn) * rand_val;
}

typedef struct {
    uint32_t mode, seed, frame, pixel, draw;
} rng_t;
static float rng_get(rng_t *r, float mn, float mx)
{
    if (r->mode == 1) {
        const float rand_val = (float)(orc_xorshf96() % 100u) / 100.0f;
        return mn + (mx - mn) * rand_val;
    }
    return orc_get_rand_counter(orc_counter_rand(r->seed, r->frame, r->pixel, r->draw++), mn, mx);
}

/* utils.cpp:94-100 via glm::rotate(mat4, angle, axis) and mat4 * mat4 (GLM
 * axis-angle form).  rot = columns m[0], m[1], m[2] of
 * mat3(ry * rx).  Host-side only; every +0 term is kept so signed zeros match. */
typedef struct { float c[4][4]; } m4; /* c[column][row] */
static m4 m4_identity(void)
{
    m4 m; memset(&m, 0, sizeof(m));
    m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
    return m;
}
static m4 m4_rotate(m4 m, float angle, v3 axis)
{
    const float c = cosf(angle), s = sinf(angle);
    const v3 a = vnormalize(axis);
    const v3 t = { (1.0f - c) * a.x, (1.0f - c) * a.y, (1.0f - c) * a.z };
    float R[3][3];
    R[0][0] = c + t.x * a.x;       R[0][1] = t.x * a.y + s * a.z; R[0][2] = t.x * a.z - s * a.y;
    R[1][0] = t.y * a.x - s * a.z; R[1][1] = c + t.y * a.y;       R[1][2] = t.y * a.z + s * a.x;
    R[2][0] = t.z * a.x + s * a.y; R[2][1] = t.z * a.y - s * a.x; R[2][2] = c + t.z * a.z;
    m4 out;
    for (int i = 0; i < 3; i++)
        for (int r = 0; r < 4; r++)
            out.c[i][r] = (m.c[0][r] * R[i][0] + m.c[1][r] * R[i][1]) + m.c[2][r] * R[i][2];
    for (int r = 0; r < 4; r++) out.c[3][r] = m.c[3][r];
    return out;
}
static m4 m4_mul(m4 A, m4 B)
{
    m4 out;
    for (int j = 0; j < 4; j++)
        for (int r = 0; r < 4; r++)
            out.c[j][r] = ((A.c[0][r] * B.c[j][0] + A.c[1][r] * B.c[j][1]) + A.c[2][r] * B.c[j][2]) + A.c[3][r] * B.c[j][3];
    return out;
}
void orc_make_rotation(float angle_x, float angle_y, float rot[9])
{
    const v3 Y = { 0.0f, 1.0f, 0.0f }, X = { 1.0f, 0.0f, 0.0f };
    const m4 rx = m4_rotate(m4_identity(), -angle_x, Y);
    const m4 ry = m4_rotate(m4_identity(), -angle_y, X);
    const m4 m = m4_mul(ry, rx);
    for (int j = 0; j < 3; j++)
        for (int r = 0; r < 3; r++)
            rot[j * 3 + r] = m.c[j][r];
}

/* ------------------------------------------------------------------------- */
/* S1..S4, R0                                                                */
/* ------------------------------------------------------------------------- */

typedef struct {
    const orc_lnode *nodes;
    uint32_t depth;
    const uint8_t *tex_top, *tex_side;
    v3 light;
    uint64_t rays, sum_complexity;
    uint16_t *trace;           /* analysis hook (orc_render_frame_trace): complexity of this pixel's rays, in cast order */
    uint32_t trace_n;
    uint8_t *ops;              /* analysis hook: this pixel's per-iteration op codes, ORC_TRACE_RAYS x ops_len bytes */
    uint32_t ops_len;
    float *geo;                /* analysis hook: this pixel's rays as {origin[3], hit position[3], hit scale, hit kind + 16 * direction signs}, ORC_TRACE_RAYS x 8 floats */
} shade_ctx;

static void ctx_cast(shade_ctx *c, v3 o, v3 d, float coef, float bias, orc_hit *out)
{
    const float of[3] = { o.x, o.y, o.z }, df[3] = { d.x, d.y, d.z };
    if (c->ops && c->trace_n < ORC_TRACE_RAYS) { orc_optrace = c->ops + (size_t)c->trace_n * c->ops_len; orc_optrace_cap = c->ops_len; }
    orc_cast_ray(c->nodes, c->depth, of, df, coef, bias, out);
    orc_optrace = 0;
    c->rays += 1;
    c->sum_complexity += out->complexity;
    if (c->geo && c->trace_n < ORC_TRACE_RAYS) {
        float *g = c->geo + (size_t)c->trace_n * 8;
        g[0] = o.x; g[1] = o.y; g[2] = o.z;
        g[3] = out->position[0]; g[4] = out->position[1]; g[5] = out->position[2];
        g[6] = (float)((out->hit >> 16) & 0xffu);
        g[7] = (float)((out->hit & 0xffu) + 16u * ((d.x > 0.0f ? 1u : 0u) | (d.y > 0.0f ? 2u : 0u) | (d.z > 0.0f ? 4u : 0u)));
    }
    if (c->trace && c->trace_n < ORC_TRACE_RAYS)
        c->trace[c->trace_n++] = (uint16_t)(out->complexity < 65535u ? out->complexity : 65535u);
}

/* camera_controller.hpp:51-54: row-vector * mat3 */
static inline v3 view_to_world(const float rot[9], v3 v)
{
    v3 r = { (rot[0] * v.x + rot[1] * v.y) + rot[2] * v.z,
             (rot[3] * v.x + rot[4] * v.y) + rot[5] * v.z,
             (rot[6] * v.x + rot[7] * v.y) + rot[8] * v.z };
    return r;
}

/* utils.cpp:43-48 */
static inline void color_mult(uint8_t c[3], float f)
{
    c[0] = (uint8_t)smin(255.0f, (float)c[0] * f);
    c[1] = (uint8_t)smin(255.0f, (float)c[1] * f);
    c[2] = (uint8_t)smin(255.0f, (float)c[2] * f);
}
/* utils.cpp:35-40 */
static inline void color_add(uint8_t a[3], const uint8_t b[3])
{
    for (int i = 0; i < 3; i++) {
        int s = (int)a[i] + (int)b[i];
        s = s < 255 ? s : 255;
        a[i] = (uint8_t)(s > 0 ? s : 0);
    }
}

/* utils.cpp:67-75 */
static inline float clampf(float v, float mn, float mx)
{
    if (v > mx) v = mx; else if (v < mn) v = mn;
    return v;
}

/* raycaster.hpp:209-240 (Cell is always {Solid, Grass}: lsvo.hpp:21-23) */
static void texture_color(const shade_ctx *c, const orc_hit *pt, uint8_t out[3])
{
    const uint8_t *img = pt->normal[1] ? c->tex_top : c->tex_side;
    const float u = clampf(pt->voxel_coord[0], 0.0f, 1.0f), v = clampf(pt->voxel_coord[1], 0.0f, 1.0f);
    const uint32_t tx = (uint32_t)(16.0f * u), ty = (uint32_t)(16.0f * v);
    uint32_t idx = tx + ty * 16u;  /* sf::Image::getPixel has no bounds check; u,v < 1 in practice */
    if (idx > 255u) idx = 255u;
    out[0] = img[idx * 3 + 0]; out[1] = img[idx * 3 + 1]; out[2] = img[idx * 3 + 2];
}

/* analysis hook: per-pixel op traces of the frame being rendered (set by orc_render_frame_optrace) */
static uint8_t *orc_frame_ops;
static uint32_t orc_frame_ops_len;
static float *orc_frame_geo;   /* ... and origin / hit position / hit scale of every ray (orc_render_frame_optrace_geo) */

/* raycaster.hpp:169-207; `level` > 0 only for the gi_bounces >= 2 extension. */
static float global_illumination(shade_ctx *c, const orc_hit *point, rng_t *rng, float inv_size,
                                 uint32_t level, uint32_t gi_bounces)
{
    /* :171-172 -- SCALE is hard-coded 1/512 in the reference; parametrised by
     * 2^-depth here (identical at depth 9). */
    const float n_normalizer = inv_size * 0.0078125f * 2.0f;
    const v3 normal = { point->normal[0], point->normal[1], point->normal[2] };
    const v3 gi_start = { point->position[0] + normal.x * n_normalizer,
                          point->position[1] + normal.y * n_normalizer,
                          point->position[2] + normal.z * n_normalizer };
    float acc = 0.0f;
    const float range = 1000.0f;
    {
        v3 noise = { 0.0f, 0.0f, 0.0f };  /* uninitialised in the reference if the normal is all zero */
        const float coord_1 = rng_get(rng, -range, range);
        const float coord_2 = rng_get(rng, -range, range);
        if (normal.x)      { noise.x = 0.0f;    noise.y = coord_1; noise.z = coord_2; }
        else if (normal.y) { noise.x = coord_1; noise.y = 0.0f;    noise.z = coord_2; }
        else if (normal.z) { noise.x = coord_1; noise.y = coord_2; noise.z = 0.0f; }
        /* :192-194 */
        const v3 pre = { (normal.x + noise.x) * n_normalizer, (normal.y + noise.y) * n_normalizer,
                         (normal.z + noise.z) * n_normalizer };
        const v3 gi_ray = vnormalize(pre);
        const float dot_gi = vdot(gi_ray, normal);
        orc_hit gi_point;
        ctx_cast(c, gi_start, gi_ray, 0.5f, 0.0f, &gi_point);
        if (gi_point.hit) {                                    /* :195-203 */
            const v3 gn = { gi_point.normal[0], gi_point.normal[1], gi_point.normal[2] };
            const v3 gls = { gi_point.position[0] + gn.x * n_normalizer,
                             gi_point.position[1] + gn.y * n_normalizer,
                             gi_point.position[2] + gn.z * n_normalizer };
            const v3 dl = { c->light.x - gls.x, c->light.y - gls.y, c->light.z - gls.z };
            const v3 to_light = vnormalize(dl);
            orc_hit gi_light;
            ctx_cast(c, gls, to_light, 0.5f, 0.0f, &gi_light);
            if (!gi_light.hit) {
                const float dot = vdot(gn, to_light);
                acc += 1000000.0f * smin(0.5f, smax(0.0f, dot) * dot_gi);
            }
            if (level + 1 < gi_bounces) /* extension, no reference behaviour */
                acc += dot_gi * global_illumination(c, &gi_point, rng, inv_size, level + 1, gi_bounces);
        }
    }
    return smax(0.0f, acc / 1.0f);                             /* :206 */
}

/* raycaster.hpp:118-167 */
static void shade_cast(shade_ctx *c, v3 start, v3 dir, const orc_frame_params *p, rng_t *rng,
                       float inv_size, uint8_t color[3], orc_hit *prim_out)
{
    color[0] = color[1] = color[2] = 0;                        /* ColorResult: Black */
    orc_hit inter;
    ctx_cast(c, start, dir, 0.0f, 0.0f, &inter);               /* :131 */
    if (prim_out) *prim_out = inter;
    if (!inter.hit) return;
    const v3 normal = { inter.normal[0], inter.normal[1], inter.normal[2] };
    /* :139 -- SCALE = 1/SVO_SIZE */
    const v3 hit_position = { inter.position[0] + normal.x * inv_size * 0.001f,
                              inter.position[1] + normal.y * inv_size * 0.001f,
                              inter.position[2] + normal.z * inv_size * 0.001f };
    texture_color(c, &inter, color);                           /* :141-145 */
    uint32_t shadow_sample = p->shadow_samples ? p->shadow_samples : (p->use_samples ? 4u : 1u); /* :147 */
    float light_intensity = 0.0f;
    for (uint32_t i = shadow_sample; i--;) {                   /* :150-158 */
        const v3 dl = { c->light.x - hit_position.x, c->light.y - hit_position.y, c->light.z - hit_position.z };
        const v3 point_to_light = vnormalize(dl);
        orc_hit li;
        ctx_cast(c, hit_position, point_to_light, 0.0f, 0.0f, &li);
        if (!li.hit) light_intensity = smax(0.0f, vdot(point_to_light, normal));
    }
    const float gi = p->use_gi ? global_illumination(c, &inter, rng, inv_size, 0, p->gi_bounces ? p->gi_bounces : 1u) : 0.0f;
    color_mult(color, smin(1.0f, smax(0.0f, light_intensity + gi))); /* :163 */
}

static int pixel_selected(const orc_frame_params *p, uint32_t x, uint32_t y)
{
    if (p->checker_parity >= 0) {
        /* main.cpp:131-132,140-143: 4x4 tiles of (W/4)x(H/4); inside a tile y
         * starts at tile_y0 + (x + offset) % 2 and steps by 2; pixels beyond
         * 4*(W/4) or 4*(H/4) are never rendered. */
        const uint32_t aw = p->width / 4u, ah = p->height / 4u;
        if (aw == 0u || ah == 0u || x >= 4u * aw || y >= 4u * ah) return 0;
        const uint32_t y0 = (y / ah) * ah;
        if (((y - y0) % 2u) != ((x + (uint32_t)p->checker_parity) % 2u)) return 0;
    }
    if (p->row_block && p->shard_count > 1 && ((y / p->row_block) % p->shard_count) != p->shard_index)
        return 0;
    return 1;
}

static void render_pixel(shade_ctx *c, const orc_camera *cam, const orc_frame_params *p,
                         uint32_t x, uint32_t y, uint8_t *image, uint32_t *accum, orc_hit *prim, uint16_t *trace)
{
    const uint32_t W = p->width, H = p->height;
    c->trace = trace ? trace + ((uint64_t)y * W + x) * ORC_TRACE_RAYS : 0;
    c->trace_n = 0;
    c->ops = (trace && orc_frame_ops) ? orc_frame_ops + ((uint64_t)y * W + x) * ORC_TRACE_RAYS * orc_frame_ops_len : 0;
    c->ops_len = orc_frame_ops_len;
    c->geo = (trace && orc_frame_geo) ? orc_frame_geo + ((uint64_t)y * W + x) * ORC_TRACE_RAYS * 8 : 0;
    const float inv_size = 1.0f / (float)(1u << c->depth);     /* main.cpp:82 */
    rng_t rng = { p->rng_mode, p->seed, p->frame_index, y * W + x, 0u };
    /* main.cpp:133,145-146 */
    const float aspect_ratio = (float)W / (float)H;
    const float lens_x = (float)x / (float)H - aspect_ratio * 0.5f;
    const float lens_y = (float)y / (float)H - 0.5f;
    /* camera_controller.hpp:34-49 -- always two draws, even at aperture 0 */
    const v3 screen = { lens_x, lens_y, cam->fov };
    const v3 ns = vnormalize(screen);
    const v3 focal_point = { ns.x * cam->focal_length, ns.y * cam->focal_length, ns.z * cam->focal_length };
    const float r0 = rng_get(&rng, -0.5f, 0.5f);
    const float r1 = rng_get(&rng, -0.5f, 0.5f);
    const v3 rand_vec = { cam->aperture * r0, cam->aperture * r1, cam->aperture * 0.0f };
    const v3 fd = { focal_point.x - rand_vec.x, focal_point.y - rand_vec.y, focal_point.z - rand_vec.z };
    const v3 ray = view_to_world(cam->rot, vnormalize(fd));
    const v3 wro = view_to_world(cam->rot, rand_vec);
    /* main.cpp:149 */
    const v3 start = { (cam->position[0] + wro.x) * inv_size + 1.0f,
                       (cam->position[1] + wro.y) * inv_size + 1.0f,
                       (cam->position[2] + wro.z) * inv_size + 1.0f };
    uint8_t col[3];
    shade_cast(c, start, ray, p, &rng, inv_size, col, prim ? &prim[(uint64_t)y * W + x] : 0);
    const uint64_t pi = (uint64_t)y * W + x;
    if (!p->use_samples) {                                     /* raycaster.hpp:77-85 */
        uint8_t old[3] = { image[pi * 4 + 0], image[pi * 4 + 1], image[pi * 4 + 2] };
        const float old_conservation = 0.4f;
        color_mult(old, old_conservation);
        color_mult(col, 1.0f - old_conservation);
        color_add(old, col);
        image[pi * 4 + 0] = old[0]; image[pi * 4 + 1] = old[1]; image[pi * 4 + 2] = old[2];
        /* alpha untouched (255 from sf::Image::create) */
    } else {                                                   /* :87-90 */
        accum[pi * 4 + 0] += col[0];
        accum[pi * 4 + 1] += col[1];
        accum[pi * 4 + 2] += col[2];
        accum[pi * 4 + 3] += 1u;
    }
}

void orc_render_frame(const orc_lnode *nodes, uint32_t depth,
                      const uint8_t tex_top[768], const uint8_t tex_side[768],
                      const orc_camera *cam, const orc_frame_params *p,
                      uint8_t *image, uint32_t *accum, orc_hit *prim,
                      orc_frame_stats *stats, int threads)
{
    orc_render_frame_trace(nodes, depth, tex_top, tex_side, cam, p, image, accum, prim, stats, threads, 0);
}

void orc_render_frame_trace(const orc_lnode *nodes, uint32_t depth,
                            const uint8_t tex_top[768], const uint8_t tex_side[768],
                            const orc_camera *cam, const orc_frame_params *p,
                            uint8_t *image, uint32_t *accum, orc_hit *prim,
                            orc_frame_stats *stats, int threads, uint16_t *trace)
{
    const uint32_t W = p->width, H = p->height;
    uint64_t rays = 0, sumc = 0, hits = 0;
    if (p->rng_mode == 1 || threads <= 1) {
        /* serial: the reference's own pixel order within its 4x4 tiles is
         * thread-racy; x-outer/y-inner (main.cpp:142-143) over the whole frame here */
        shade_ctx c = { nodes, depth, tex_top, tex_side,
                        { p->light_position[0], p->light_position[1], p->light_position[2] }, 0, 0, 0, 0, 0, 0, 0 };
        for (uint32_t x = 0; x < W; x++)
            for (uint32_t y = 0; y < H; y++)
                if (pixel_selected(p, x, y))
                    render_pixel(&c, cam, p, x, y, image, accum, prim, trace);
        rays = c.rays; sumc = c.sum_complexity;
    } else {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : rays, sumc)
        for (int32_t y = 0; y < (int32_t)H; y++) {
            shade_ctx c = { nodes, depth, tex_top, tex_side,
                            { p->light_position[0], p->light_position[1], p->light_position[2] }, 0, 0, 0, 0, 0, 0, 0 };
            for (uint32_t x = 0; x < W; x++)
                if (pixel_selected(p, x, (uint32_t)y))
                    render_pixel(&c, cam, p, x, (uint32_t)y, image, accum, prim, trace);
            rays += c.rays; sumc += c.sum_complexity;
        }
    }
    if (prim)
        for (uint64_t i = 0; i < (uint64_t)W * H; i++) hits += prim[i].hit ? 1 : 0;
    if (stats) { stats->rays = rays; stats->sum_complexity = sumc; stats->primary_hits = hits; }
}

/* orc_render_frame_trace plus, per pixel and ray, one op code per loop iteration (see orc_cast_ray) into
 * ops[(pixel * ORC_TRACE_RAYS + ray) * ops_len + iteration].  Not re-entrant (analysis only). */
void orc_render_frame_optrace(const orc_lnode *nodes, uint32_t depth,
                              const uint8_t tex_top[768], const uint8_t tex_side[768],
                              const orc_camera *cam, const orc_frame_params *p,
                              int threads, uint16_t *trace, uint8_t *ops, uint32_t ops_len)
{
    uint8_t *image = (uint8_t *)calloc((size_t)p->width * p->height, 4);
    uint32_t *accum = (uint32_t *)calloc((size_t)p->width * p->height, 16);
    orc_frame_ops = ops; orc_frame_ops_len = ops_len;
    orc_render_frame_trace(nodes, depth, tex_top, tex_side, cam, p, image, accum, 0, 0, threads, trace);
    orc_frame_ops = 0; orc_frame_ops_len = 0;
    free(image); free(accum);
}

/* ... plus, per pixel and ray, {origin[3], hit position[3], hit scale, hit kind + 16 * (bit i: the ray goes towards +axis i)}
 * into geo[(pixel * ORC_TRACE_RAYS + ray) * 8]:
 * what tests/tools/path_schedule.py needs to tell where the kernel starts each ray (below the root, next to the previous hit). */
void orc_render_frame_optrace_geo(const orc_lnode *nodes, uint32_t depth,
                                  const uint8_t tex_top[768], const uint8_t tex_side[768],
                                  const orc_camera *cam, const orc_frame_params *p,
                                  int threads, uint16_t *trace, uint8_t *ops, uint32_t ops_len, float *geo)
{
    orc_frame_geo = geo;
    orc_render_frame_optrace(nodes, depth, tex_top, tex_side, cam, p, threads, trace, ops, ops_len);
    orc_frame_geo = 0;
}

/* The same frame with the REFERENCE's own decomposition (main.cpp:90-92,139-152): 16 threads, thread t owns the
 * area (t % 4, t / 4) of (W/4) x (H/4) pixels and walks it x-outer / y-inner; pixels beyond 4*(W/4) x 4*(H/4) are
 * never rendered (as in the reference).  Static assignment, no work stealing: the frame takes as long as its
 * slowest area.  Used by bench.py's cpu_baseline leg ("decomp": "4x4x16"); results are identical to
 * orc_render_frame on the pixels both render. */
void orc_render_frame_areas16(const orc_lnode *nodes, uint32_t depth,
                              const uint8_t tex_top[768], const uint8_t tex_side[768],
                              const orc_camera *cam, const orc_frame_params *p,
                              uint8_t *image, uint32_t *accum, orc_frame_stats *stats)
{
    const uint32_t aw = p->width / 4u, ah = p->height / 4u;
    uint64_t rays = 0, sumc = 0;
#pragma omp parallel num_threads(16) reduction(+ : rays, sumc)
    {
#pragma omp for schedule(static, 1)
        for (int t = 0; t < 16; t++) {
            shade_ctx c = { nodes, depth, tex_top, tex_side,
                            { p->light_position[0], p->light_position[1], p->light_position[2] }, 0, 0, 0, 0, 0, 0, 0 };
            const uint32_t x0 = ((uint32_t)t % 4u) * aw, y0 = ((uint32_t)t / 4u) * ah;
            for (uint32_t x = x0; x < x0 + aw; x++)
                for (uint32_t y = y0; y < y0 + ah; y++)
                    if (pixel_selected(p, x, y))
                        render_pixel(&c, cam, p, x, y, image, accum, 0, 0);
            rays += c.rays; sumc += c.sum_complexity;
        }
    }
    if (stats) { stats->rays = rays; stats->sum_complexity = sumc; stats->primary_hits = 0; }
}

/* One worker's share of a frame under the reference's decomposition, as a job a thread pool can call: worker `id` renders
 * area (id % 4, id / 4) of (W/4) x (H/4) pixels, x-outer / y-inner -- the body of the swarm lambda, main.cpp:139-152.
 * `user` is an orc_area_frame; rays / complexity are summed per worker (no shared counters).  bench.py hands this
 * function to the reference's own swarm pool (oracle/_ref/libref_swarm.so). */
void orc_area_job(uint32_t id, uint32_t worker_count, void *user)
{
    (void)worker_count;
    orc_area_frame *f = (orc_area_frame *)user;
    const orc_frame_params *p = f->params;
    const uint32_t aw = p->width / 4u, ah = p->height / 4u;
    if (id >= 16u) return;
    shade_ctx c = { f->nodes, f->depth, f->tex_top, f->tex_side,
                    { p->light_position[0], p->light_position[1], p->light_position[2] }, 0, 0, 0, 0, 0, 0, 0 };
    const uint32_t x0 = (id % 4u) * aw, y0 = (id / 4u) * ah;
    for (uint32_t x = x0; x < x0 + aw; x++)
        for (uint32_t y = y0; y < y0 + ah; y++)
            if (pixel_selected(p, x, y))
                render_pixel(&c, f->cam, p, x, y, f->image, f->accum, 0, 0);
    f->rays[id] = c.rays; f->sum_complexity[id] = c.sum_complexity;
}

/* raycaster.hpp:94-103: uint8(sum / count) in double == integer division for
 * integer sums (see vrc_oracle.h).  count == 0 (never-sampled pixel) is NaN ->
 * UB in the reference; black here. */
void orc_samples_to_image(const uint32_t *accum, uint32_t width, uint32_t height, uint8_t *image)
{
    for (uint64_t i = 0; i < (uint64_t)width * height; i++) {
        const uint32_t n = accum[i * 4 + 3];
        for (int k = 0; k < 3; k++) {
            const double v = n ? (double)accum[i * 4 + k] / (double)n : 0.0;
            image[i * 4 + k] = (uint8_t)v;
        }
        image[i * 4 + 3] = 255;
    }
}


/* ------------------------------------------------------------------------- */
/* N2: frame post-process / present chain (main.cpp:160-182)                  */
/* ------------------------------------------------------------------------- */

/* Fixed-function GL blending on an RGBA8 render target, restated in integers: a UNORM8 product a/255 * b/255 is
 * stored as round(a*b/255) (OpenGL 4.x 2.3.5: conversion to UNORM rounds to nearest; 255 is odd, so there are no
 * ties).  Hardware may differ from this by one LSB -- that is the stated tolerance of the blend steps against a real
 * GL device; the HIP path equals THIS definition bit for bit. */
static inline uint8_t unorm8_mul(uint32_t a, uint32_t b) { return (uint8_t)((a * b + 127u) / 255u); }

/* Median of 9 / 25 values as the min/max exchange networks of res/median_3.frag and res/median.frag
 * (McGuire & Whitson, 2006; the exchange lists are data, written out in vrc_present.hip's tables too).
 *   s2(a, b): t = a; a = min(a, b); b = max(t, b) */
#define ORC_S2(a, b) do { const uint8_t t_ = (a); (a) = (a) < (b) ? (a) : (b); (b) = t_ > (b) ? t_ : (b); } while (0)

static uint8_t median9_net(const uint8_t w[9])
{
    /* median_3.frag:57-82; w in fetch order (-1,-1) (0,-1) (+1,-1) (-1,0) (0,0) (+1,0) (-1,+1) (0,+1) (+1,+1) */
    uint8_t v[6] = { w[0], w[1], w[2], w[3], w[4], w[5] };
    ORC_S2(v[0], v[3]); ORC_S2(v[1], v[4]); ORC_S2(v[2], v[5]);            /* mnmx6 */
    ORC_S2(v[0], v[1]); ORC_S2(v[0], v[2]);
    ORC_S2(v[4], v[5]); ORC_S2(v[3], v[5]);
    v[5] = w[6];
    ORC_S2(v[1], v[2]); ORC_S2(v[3], v[4]);                                /* mnmx5 */
    ORC_S2(v[1], v[3]); ORC_S2(v[1], v[5]);
    ORC_S2(v[4], v[5]); ORC_S2(v[2], v[5]);
    v[5] = w[7];
    ORC_S2(v[2], v[3]); ORC_S2(v[4], v[5]); ORC_S2(v[2], v[4]); ORC_S2(v[3], v[5]);   /* mnmx4 */
    v[5] = w[8];
    ORC_S2(v[4], v[5]); ORC_S2(v[3], v[5]); ORC_S2(v[3], v[4]);            /* mnmx3 */
    return v[4];
}

static const uint8_t MEDIAN25_PAIRS[][2] = {                               /* median.frag:68-86 */
    {0,1},{3,4},{2,4},{2,3},{6,7}, {5,7},{5,6},{9,7},{1,7},{1,4}, {12,13},{11,13},{11,12},{15,16},{14,16},
    {14,15},{18,19},{17,19},{17,18},{21,22}, {20,22},{20,21},{23,24},{2,5},{3,6}, {0,6},{0,3},{4,7},{1,7},{1,4},
    {11,14},{8,14},{8,11},{12,15},{9,15}, {9,12},{13,16},{10,16},{10,13},{20,23}, {17,23},{17,20},{21,24},{18,24},{18,21},
    {19,22},{8,17},{9,18},{0,18},{0,9}, {10,19},{1,19},{1,10},{11,20},{2,20}, {2,11},{12,21},{3,21},{3,12},{13,22},
    {4,22},{4,13},{14,23},{5,23},{5,14}, {15,24},{6,24},{6,15},{7,16},{7,19}, {3,11},{5,17},{11,17},{9,17},{4,10},
    {6,12},{7,14},{4,6},{4,7},{12,14}, {10,14},{6,7},{10,12},{6,10},{6,17}, {12,17},{7,17},{7,10},{12,18},{7,12},
    {10,18},{12,20},{10,20},{10,12},
};

static uint8_t median25_net(uint8_t v[25])
{
    for (size_t i = 0; i < sizeof(MEDIAN25_PAIRS) / sizeof(MEDIAN25_PAIRS[0]); i++)
        ORC_S2(v[MEDIAN25_PAIRS[i][0]], v[MEDIAN25_PAIRS[i][1]]);
    return v[12];
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orc_present(const uint8_t *image, uint8_t *denoised, uint32_t width, uint32_t height,
                 float old_value_conservation, uint32_t median, uint8_t *window, uint32_t out_width, uint32_t out_height)
{
    /* main.cpp:161-166: sf::Color(255 * c, ...) truncates the float to Uint8 */
    const uint32_t c1 = (uint32_t)(uint8_t)(255 * old_value_conservation);
    const uint32_t c2 = (uint32_t)(uint8_t)(255 * (1.0f - old_value_conservation));
    const uint64_t n = (uint64_t)width * height;
    for (uint64_t i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            const uint32_t r = unorm8_mul(image[i * 4 + k], c2);            /* :169-172 sprite drawn opaque, then x cache2 (BlendMultiply) */
            const uint32_t d = unorm8_mul(denoised[i * 4 + k], c1);         /* :175 denoised x cache1 (BlendMultiply) */
            const uint32_t sum = d + r;                                     /* :176-177 BlendAdd: src * srcAlpha(=1) + dst, clamped */
            denoised[i * 4 + k] = (uint8_t)(sum > 255u ? 255u : sum);
        }
        denoised[i * 4 + 3] = 255;                                          /* alpha: 255 * 255 / 255, then min(1, a + a) */
    }
    if (!window) return;
    /* :179-182: the sprite scaled by 1 / render_scale with nearest sampling (textures are not smooth): window pixel
     * (X, Y) takes the texel that contains (X + 0.5) * width / out_width; optional median around that texel, taps clamped
     * to the edge (SFML textures are not repeated), exactly one texel apart (the shaders hard-code 1/800) */
    for (uint32_t Y = 0; Y < out_height; Y++) {
        const uint32_t sy = (uint32_t)(((uint64_t)(2u * Y + 1u) * height) / (2ull * out_height));
        for (uint32_t X = 0; X < out_width; X++) {
            const uint32_t sx = (uint32_t)(((uint64_t)(2u * X + 1u) * width) / (2ull * out_width));
            uint8_t *o = window + ((uint64_t)Y * out_width + X) * 4;
            for (int k = 0; k < 3; k++) {
                if (median == 3) {
                    uint8_t w[9];
                    int q = 0;
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++)
                            w[q++] = denoised[((uint64_t)clampi((int)sy + dy, 0, (int)height - 1) * width + (uint32_t)clampi((int)sx + dx, 0, (int)width - 1)) * 4 + k];
                    o[k] = median9_net(w);
                } else if (median == 5) {
                    uint8_t v[25];
                    for (int dX = -2; dX <= 2; dX++)                         /* median.frag:54-62: v[(dX + 2) * 5 + (dY + 2)] */
                        for (int dY = -2; dY <= 2; dY++)
                            v[(dX + 2) * 5 + (dY + 2)] = denoised[((uint64_t)clampi((int)sy + dY, 0, (int)height - 1) * width + (uint32_t)clampi((int)sx + dX, 0, (int)width - 1)) * 4 + k];
                    o[k] = median25_net(v);
                } else {
                    o[k] = denoised[((uint64_t)sy * width + sx) * 4 + k];
                }
            }
            o[3] = 255;
        }
    }
}
