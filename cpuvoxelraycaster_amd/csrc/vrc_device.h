// vrc_device.h -- device-side restatement of the reference's per-ray arithmetic
// for gfx950 (wave64).  Included by vrc_kernels.hip only.
//
// Bit-exactness contract: the octree walk is
// driven by float comparisons, so every operation below is written in the
// reference's order and the file is compiled with -ffp-contract=off (hipcc
// fuses a*b+c into v_fma_f32 by default, which changes hit voxels) and with
// correctly rounded f32 divide / sqrt.  std::max / std::min are restated as
// the exact ternaries (argument order matters for NaN and signed zeros).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vrc {

__device__ __forceinline__ float smax(float a, float b) { return (a < b) ? b : a; }  // std::max(a,b)
__device__ __forceinline__ float smin(float a, float b) { return (b < a) ? b : a; }  // std::min(a,b)
// Inside the walk every t value is finite-or-inf, never NaN (non-finite rays are
// rejected in ray_setup; products px*t_coef are finite and differences of a
// finite and an infinite value are infinite) and never -0 (a float difference is
// -0 only for (-0) - (+0), and px*t_coef != 0).  On that domain std::min/max equal
// IEEE minNum/maxNum, so the loop may use v_min_f32 / v_min3_f32 (1 instruction)
// instead of v_cmp + v_cndmask (2): same bits.
// (inline asm: via __builtin_fminf hipcc adds a canonicalising v_max x,x per operand.)
__device__ __forceinline__ float tmin(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float tmin3(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }       // utils.cpp:109-112
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }       // utils.cpp:115-118
// utils.cpp:60-64 std::modf fractional part: f - trunc(f) is exact; copysign keeps
// modf's -0; modf(+-inf) = +-0 (inf - inf would be NaN).
__device__ __forceinline__ float fracf_(float f)
{
    const float r = (fabsf(f) == __builtin_inff()) ? 0.0f : f - truncf(f);
    return copysignf(r, f);
}
// glm::dot / glm::normalize (stock scalar GLM)
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;
}
__device__ __forceinline__ float exact_rcp(float x);
// sqrt(x), correctly rounded, for 2^-100 <= x < 2^126: y = v_rsq_f32(x), g = x * y ~ sqrt, h = y / 2; the residual
// d = x - g * g is exact in one FMA and g + d * h rounds to the nearest float of the true root
// (tools/micro/sqrt_exact.hip: equal to the IEEE sqrtf for every bit pattern with exponent field 25..254).
__device__ __forceinline__ float exact_sqrt(float x)
{
    float y;
    asm("v_rsq_f32_e32 %0, %1" : "=v"(y) : "v"(x));
    const float g = x * y, h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
// glm::normalize: v * (1 / sqrt(dot(v, v))) -- two correctly rounded operations, then three products.  In the range
// where the short sequences are proven exact they replace the IEEE expansions (~22 instructions -> 8); a zero, tiny,
// huge, infinite or NaN squared length takes the IEEE path (same results by definition).
__device__ __forceinline__ void normalize3(float& x, float& y, float& z)
{
    const float s = dot3(x, y, z, x, y, z);
    float inv;
    if (f2u(s) - 0x0D800000u < 0x7E800000u - 0x0D800000u) inv = exact_rcp(exact_sqrt(s));   // 2^-100 <= s < 2^126
    else inv = 1.0f / sqrtf(s);
    x = x * inv; y = y * inv; z = z * inv;
}

// 1 / x, correctly rounded, for 2^-126 <= x < 2^126 (see ray_setup): y0 = v_rcp_f32(x) is within 1 ulp; the residual
// e = 1 - x * y0 is exact in one FMA, and y0 + y0 * e rounds to the nearest float of the true quotient.  These two FMAs
// are the Newton step itself, not a contraction of reference arithmetic (the reference's operation here is one division).
__device__ __forceinline__ float exact_rcp(float x)
{
    float y0;
    asm("v_rcp_f32_e32 %0, %1" : "=v"(y0) : "v"(x));
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(y0, e, y0);
}

// Counter-based RNG replacing the reference's racy global xorshf96
// (utils.cpp:11-25); identical integer arithmetic to oracle/vrc_oracle.c.
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t counter_rand(uint32_t seed, uint32_t frame, uint32_t pixel, uint32_t draw)
{
    uint32_t h = mix32(seed ^ (frame * 0x9E3779B1u));
    h = mix32(h ^ (pixel * 0x85EBCA77u));
    h = mix32(h ^ (draw * 0xC2B2AE3Du) ^ 0x27D4EB2Fu);
    return h;
}
// getRand (utils.cpp:77-81): 100 levels
__device__ __forceinline__ float get_rand(uint32_t k, float mn, float mx)
{
    // float(m) / 100.0f for m = 0..99, correctly rounded: q0 = m * RN(1/100), exact residual, one correction -- equal to
    // the IEEE quotient for all 100 values (tools/micro/sqrt_exact.hip; vrc_selftest_exact_arith), 3 instructions for ~10
    const float m = (float)(k % 100u);
    const float y = 1.0f / 100.0f;
    const float q0 = m * y;
    const float rand_val = __builtin_fmaf(__builtin_fmaf(-100.0f, q0, m), y, q0);
    return mn + (mx - mn) * rand_val;
}

constexpr int SVO_MAX = 23;  // lsvo.hpp:37
typedef uint32_t stack_t;    // one stack entry: a parent index (see ray_step); a walk needs `depth` rows of them
// camera paths in the frame kernels' LDS (camera_path): 8 variants x 16 entries (rows by scale, entry 15 = the path's scale), then
// one entry with the lowest of those scales and 7 of padding
constexpr uint32_t VRC_CAMERA_PATH_BYTES = 8u * 16u * 4u + 32u;
// Safety net (same constant as the oracle): no finite ray needs anywhere near
// this many loop iterations; guarantees the kernels terminate.
constexpr uint32_t MAX_ITERATIONS = 65536u;

// Traversal state of one ray (one lane).  LSVO<N>::castRay, lsvo.hpp:33-172.
struct Ray {
    float tcx, tcy, tcz;      // t_coef   :47
    float tox, toy, toz;      // t_offset :48-52
    float px, py, pz;         // pos      :63
    float t_min, t_max, h;    // :54-58; t_max stays min(1, exit of the root) during the walk (see ray_step)
    float scale_f;            // :64
    uint32_t parent_id;       // :60
    int scale;                // :62 (int8_t in the reference; values stay in [0,23])
    uint32_t child_offset;    // :61
    uint32_t mirror;          // :49
    uint32_t normal;          // :69 last step mask
    uint32_t complexity;      // HitPoint::complexity :73
    uint32_t kind;            // 0 = running / miss, 1 = leaf hit, 2 = LOD cut
    uint32_t shift;           // child_shift at the hit
    uint2 nd;                 // raw_data[parent_id], reloaded only when parent_id changes
};

__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    // |v| < inf is false for NaN and +-inf
    return (fabsf(x) < __builtin_inff()) && (fabsf(y) < __builtin_inff()) && (fabsf(z) < __builtin_inff());
}

// lsvo.hpp:36-70.  (dx,dy,dz) is conditioned in place (:44-46).  Returns
// whether the walk may start: false for non-finite input, for which the
// reference's loop never terminates (no comparison at :116-118 holds) -- defined
// as an immediate miss with complexity 0, in the oracle too.
__device__ __forceinline__ bool ray_setup(Ray& r, const uint2 root,
                                          float ox, float oy, float oz, float& dx, float& dy, float& dz)
{
    const float EPS = 1.0f / (float)(1 << SVO_MAX);
    if (fabsf(dx) < EPS) dx = copysignf(EPS, dx);
    if (fabsf(dy) < EPS) dy = copysignf(EPS, dy);
    if (fabsf(dz) < EPS) dz = copysignf(EPS, dz);
    // :47  t_coef = -1 / |d|, correctly rounded.  For 2^-126 <= x < 2^126 one Newton step on v_rcp_f32 with two FMAs IS the
    // correctly rounded quotient (tools/micro/rcp_exact.hip compares all 2^32 bit patterns against the IEEE division on
    // gfx950: the only differences are denormal x, x >= 2^126 and infinities), 3 instructions instead of ~10.  |d| is at
    // least 2^-23 here; components of 2^126 and more (and infinities) take the IEEE division.
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    if (!(fmaxf(fmaxf(ax, ay), az) < 0x1p126f)) {
        r.tcx = -1.0f / ax; r.tcy = -1.0f / ay; r.tcz = -1.0f / az;
    } else {
        r.tcx = -exact_rcp(ax); r.tcy = -exact_rcp(ay); r.tcz = -exact_rcp(az);
    }
    r.tox = ox * r.tcx; r.toy = oy * r.tcy; r.toz = oz * r.tcz;
    r.mirror = 7u;
    if (dx > 0.0f) { r.mirror ^= 1u; r.tox = 3.0f * r.tcx - r.tox; }
    if (dy > 0.0f) { r.mirror ^= 2u; r.toy = 3.0f * r.tcy - r.toy; }
    if (dz > 0.0f) { r.mirror ^= 4u; r.toz = 3.0f * r.tcz - r.toz; }
    r.t_min = smax(2.0f * r.tcx - r.tox, smax(2.0f * r.tcy - r.toy, 2.0f * r.tcz - r.toz));
    r.t_max = smin(r.tcx - r.tox, smin(r.tcy - r.toy, r.tcz - r.toz));
    r.h = r.t_max;
    r.t_min = smax(0.0f, r.t_min);
    r.t_max = smin(1.0f, r.t_max);
    r.parent_id = 0u;
    r.child_offset = 0u;
    r.scale = SVO_MAX - 1;
    r.px = 1.0f; r.py = 1.0f; r.pz = 1.0f;
    r.scale_f = 0.5f;
    if (1.5f * r.tcx - r.tox > r.t_min) { r.child_offset ^= 1u; r.px = 1.5f; }
    if (1.5f * r.tcy - r.toy > r.t_min) { r.child_offset ^= 2u; r.py = 1.5f; }
    if (1.5f * r.tcz - r.toz > r.t_min) { r.child_offset ^= 4u; r.pz = 1.5f; }
    r.normal = 0u;
    r.complexity = 0u;
    r.kind = 0u;
    r.shift = 0u;
    r.nd = root;       // raw_data[0], loaded once per kernel by the caller
    return finite3(ox, oy, oz) && finite3(dx, dy, dz);
}

// One iteration of the while loop at lsvo.hpp:72-146.  `stk` points at this
// lane's column of the LDS stack, STRIDE = lanes per row, so slot s lives at
// stk[s*stride]; slots are (scale - DEPTH_OFFSET) in [0, depth-1]: one per scale a parent can have (slot 0, the scale of the
// leaf voxels, is only written at the end of a walk: the parent of the hit, see ray_walk).
// The stack holds parent indices only.  The reference also keeps t_max per level (:97-100, :134-136), but t_max is always
// min(1, exit of the parent's cell): :57-58 set it so for the root, :110 keeps it -- tc_max, the exit of the child cell, is
// never larger than the exit of the cell around it (x -> x * t_coef - t_offset is monotone in float arithmetic as well, and
// a child's corner is never below its parent's) --, :134-136 restore it.  So tv_max (:86) = min(1, tc_max) = min(T1, tc_max)
// and `t_min <= t_max` (:81) = `t_min <= T1` with the per-ray constant T1 = min(1, exit of the root) that ray_setup leaves
// in r.t_max: t_min, the exit of a cell inside the parent, never exceeds the parent's exit once the ray is inside the cube.
// (oracle/vrc_walk_model.c restates this formulation on the CPU; tests/test_walk_model.py holds it to the oracle.)
// Returns the loop condition for the next iteration.
// LOD = false: the caller guarantees ray_size_coef == 0 and ray_size_bias == 0 (primary and
// shadow rays, raycaster.hpp:131,153).  Then tc_max * 0 + 0 is 0 (or NaN for an infinite
// tc_max) and never >= scale_f > 0, so the LOD test of :82 cannot fire and is not evaluated.
template <int STRIDE, bool LOD = true>
__device__ __forceinline__ bool ray_step(Ray& r, const uint2* __restrict__ nodes, stack_t* stk,
                                         int depth, float coef, float bias)
{
    const int depth_off = SVO_MAX - depth;
    if (r.complexity >= MAX_ITERATIONS) return false;
    ++r.complexity;                                                     // :73
    const uint2 nd = r.nd;                                              // :74
    const float cx = r.px * r.tcx - r.tox;                              // :76
    const float cy = r.py * r.tcy - r.toy;
    const float cz = r.pz * r.tcz - r.toz;
    const float tc_max = tmin3(cx, cy, cz);                             // :77
    const uint32_t child_shift = r.child_offset ^ r.mirror;            // :79
    const uint32_t masks = nd.x >> child_shift;                         // bit 8: child, bit 16: leaf
    // The nested tests of :81-95 as flat predicates (same comparisons, same order of
    // evaluation where it matters: every operand is a plain value, no side effects).
    // :81 `child exists && t_min <= t_max` guards both the LOD test (:82) and the span test (:89).  The
    // span test alone implies it: tv_max = min(t_max, tc_max) <= t_max (no NaN in the walk), so the
    // separate comparison is only evaluated where the LOD test needs it.
    const bool child = (masks & 0x100u) != 0u;
    const bool lod = LOD && child && r.t_min <= r.t_max && (tc_max * coef + bias >= r.scale_f);   // :81-82
    const float tv_max = tmin(r.t_max, tc_max);                         // :86
    const bool inside = child && !lod && r.t_min <= tv_max;             // :81,:89
    const bool leaf = inside && (masks & 0x10000u);                     // :90-92
    if (lod || leaf) {                                                  // :83-84 / :93-94
        r.kind = lod ? 2u : 1u; r.shift = child_shift;
        return false;
    }
    if (inside) {                                                       // descend, :96-111
        const float half = r.scale_f * 0.5f;                            // :87
        const float thx = half * r.tcx + cx;                            // :88
        const float thy = half * r.tcy + cy;
        const float thz = half * r.tcz + cz;
        if (tc_max < r.h)                                               // :97-100
            stk[(r.scale - depth_off) * STRIDE] = r.parent_id;
        r.h = tc_max;                                                   // :101
        r.parent_id += nd.y + child_shift;                              // :103
        r.nd = nodes[r.parent_id];
        --r.scale;
        r.scale_f = half;
        const bool bx = thx > r.t_min, by = thy > r.t_min, bz = thz > r.t_min;   // :107-109
        r.child_offset = (bx ? 1u : 0u) | (by ? 2u : 0u) | (bz ? 4u : 0u);
        r.px += bx ? half : 0.0f;      // x + 0 == x bit for bit (x is never -0 here)
        r.py += by ? half : 0.0f;
        r.pz += bz ? half : 0.0f;
        // :110 t_max = tv_max: not kept (see above)
        // :72 `scale < 23 && scale > N` holds after every descend for N <= VRC_MAX_DEPTH = 11: leaf
        // level children sit at scale 23 - N, so a descend ends at scale >= 24 - N >= 13 > N.
        return true;
    }
    const uint32_t opx = f2u(r.px), opy = f2u(r.py), opz = f2u(r.pz);   // pos before the step
    const bool sx = cx <= tc_max, sy = cy <= tc_max, sz = cz <= tc_max;   // :115-118
    const uint32_t step_mask = (sx ? 1u : 0u) | (sy ? 2u : 0u) | (sz ? 4u : 0u);
    r.px -= sx ? r.scale_f : 0.0f;     // x - 0 == x bit for bit
    r.py -= sy ? r.scale_f : 0.0f;
    r.pz -= sz ? r.scale_f : 0.0f;
    r.t_min = tc_max;                                                   // :120
    r.child_offset ^= step_mask;                                        // :121
    r.normal = step_mask;                                               // :122
    if (r.child_offset & step_mask) {                                   // :124 pop
        // :125-131 differing_bits = OR over the stepped axes of bits(pos) ^ bits(pos + scale_f).
        // pos and scale_f are exact multiples of 2^-23 in [0.5, 2), so pos - scale_f + scale_f
        // is bit-identical to the pre-step pos: the XOR against the saved bits is the same
        // value, and it is 0 on the axes that did not step -- no per-axis select needed.
        const uint32_t ipx = f2u(r.px), ipy = f2u(r.py), ipz = f2u(r.pz);
        const uint32_t differing_bits = (ipx ^ opx) | (ipy ^ opy) | (ipz ^ opz);
        // :132 -- exponent of (float)differing_bits.  pos stays in [0.5, 2) so
        // differing_bits < 2^24 converts exactly and the exponent is the index
        // of the highest set bit.  differing_bits == 0 (unreachable, UB in the
        // reference) yields scale = -1 and ends the walk as a miss.
        r.scale = 31 - __clz((int)differing_bits);
        if ((uint32_t)r.scale >= (uint32_t)SVO_MAX) {
            // ray left the root (scale == 23; or the unreachable -1): the reference reads the never
            // written stack[N] here and then leaves the loop (App. A.9).  A pop never lands at
            // scale <= N for N <= 11 (see the descend comment), so :72's lower bound is not tested.
            return false;
        }
        r.scale_f = u2f((uint32_t)(r.scale - SVO_MAX + 127) << SVO_MAX);  // :133
        r.parent_id = stk[(r.scale - depth_off) * STRIDE];               // :134-136
        r.nd = nodes[r.parent_id];
        // :137-143  (i >> scale) << scale == i & (~0 << scale); child bit = bit `scale` of i
        const uint32_t keep = 0xffffffffu << r.scale;
        r.px = u2f(ipx & keep); r.py = u2f(ipy & keep); r.pz = u2f(ipz & keep);
        r.child_offset = __builtin_amdgcn_ubfe(ipx, (uint32_t)r.scale, 1u) | (__builtin_amdgcn_ubfe(ipy, (uint32_t)r.scale, 1u) << 1u) |
                         (__builtin_amdgcn_ubfe(ipz, (uint32_t)r.scale, 1u) << 2u);
        r.h = 0.0f;                                                      // :144
    }
    return true;
}

}  // namespace vrc
#include "vrc_walk.h"
namespace vrc {

// The whole walk of one ray.  LOD_MODE: 0 = coef == bias == 0 (primary / shadow rays), 1 = bias == 0, 2 = general.
template <int STRIDE, int LOD_MODE, bool BELOW = false>   // BELOW: the ray may start below the root (ray_start_below, further down)
__device__ __forceinline__ void ray_run(Ray& r, bool active, const uint2* __restrict__ nodes, stack_t* stk,
                                        int depth, float coef, float bias)
{
#ifdef VRC_WALK_CPP
    while (active) active = ray_step<STRIDE, LOD_MODE != 0>(r, nodes, stk, depth, coef, bias);
#else
    static_assert(STRIDE == 256 || STRIDE == 512, "vrc_walk.h addresses LDS stack rows of 256 lanes, 1 or 2 KiB apart");
    if (BELOW && active) r.complexity = (uint32_t)(SVO_MAX - 1 - r.scale);       // :73 for the iterations not executed
    ray_walk<LOD_MODE, true, BELOW, STRIDE == 512>(r, active, nodes, stk, depth, coef, bias);
#endif
}

// The same for callers that only total rays and iterations (the stage-synchronous frame kernels): books the call in the
// wave's LDS counters -- ctr[0] += loop iterations of all lanes, ctr[1] += lanes that cast a ray -- and leaves
// r.complexity alone.  One no-return LDS add pair by the first lane that ran; no per-lane counter stays live.
// The stage-synchronous kernels keep, per wave, {iterations executed, rays, primary hits, pixel-samples} in four consecutive
// u64 (16 of them for the workgroup's four waves), the camera paths (VRC_CAMERA_PATH_BYTES) after those, and then the waves' counts of
// iterations NOT executed (rays that start below the root): ctr[VRC_CTR_NOT_EXECUTED] from a wave's own ctr.
#define VRC_CTR_NOT_EXECUTED ((16u * 8u + VRC_CAMERA_PATH_BYTES) / 8u - 3u * (threadIdx.x >> 6))
__device__ __forceinline__ void book_walk(unsigned long long* ctr, uint32_t wave_steps)
{
    // every lane that makes the call casts a ray (one that ray_setup rejected is a ray of complexity 0, as in the oracle)
    const unsigned long long here = __ballot(true);
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u)) == 0u) {
        __hip_atomic_fetch_add(ctr + 0, (unsigned long long)wave_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(ctr + 1, (unsigned long long)__popcll(here), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

// ... and for a walk that kept per-lane complexities (primary-hit capture on): every lane books its own
__device__ __forceinline__ void book_walk_lanes(unsigned long long* ctr, uint32_t complexity)
{
    const unsigned long long here = __ballot(true);
    __hip_atomic_fetch_add(ctr + 0, (unsigned long long)complexity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u)) == 0u)
        __hip_atomic_fetch_add(ctr + 1, (unsigned long long)__popcll(here), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <int STRIDE, int LOD_MODE>
__device__ __forceinline__ void ray_run_total(Ray& r, bool active, const uint2* __restrict__ nodes, stack_t* stk,
                                              int depth, float coef, float bias, unsigned long long* ctr)
{
#if defined(VRC_WALK_CPP) || defined(VRC_PROBE)
    ray_run<STRIDE, LOD_MODE>(r, active, nodes, stk, depth, coef, bias);
    book_walk_lanes(ctr, r.complexity);
#else
    static_assert(STRIDE == 256 || STRIDE == 512, "vrc_walk.h addresses LDS stack rows of 256 lanes, 1 or 2 KiB apart");
    const uint32_t wave_steps = ray_walk<LOD_MODE, false, false, STRIDE == 512>(r, active, nodes, stk, depth, coef, bias);
    book_walk(ctr, wave_steps);
#endif
}

// ---- secondary rays that start below the root --------------------------------------------------------------------------
// A ray that starts next to the hit of the ray before it (raycaster.hpp:139,153 the shadow ray; :171-176 the GI ray; :195-198
// the GI shadow ray) begins, in the reference, with pure descends from the root to the cell that holds its origin: the child
// is present and no leaf (the previous walk went through the same nodes), t_min = 0 lies before every exit, and the child
// chosen at :66-68 / :107-109 is the one around the origin.  Those iterations leave exactly the state of a pop (:133-143) to
// their last scale with pos = the origin, so the walk may start there: parent from the stack the previous walk left (its
// rows above the hit are the path from the root; ray_walk stores the hit's parent in the row of the hit's scale), complexity
// 22 - scale for the iterations not executed.  This holds when every comparison of the skipped iterations is decided with a
// margin.  start_scale_next_to returns the smallest scale s0 that satisfies
//   - origin and hit position agree in all bits above s0: both in [1, 2) and in the same cells down to the parent of scale s0,
//   - s0 >= the scale of the hit (the deeper rows of the stack are not on the hit's path),
//   - no multiple of 2^s0 ulps within 8 ulps of an origin coordinate: the comparisons test the centre planes of the cells
//     above (c - o) * t_coef-wise, their rounding error is below 4.5 ulps of the coordinate,
// or a value >= 22 when there is none (the ray then starts at the root).  A ray cast with an LOD coefficient needs one thing
// more: the LOD test (:82) must not fire in a cell that holds the origin.  tc_max <= scale_f / max|d_i| there, so with bias 0
// it cannot when coef < max|d_i| -- start_scale_next_to_lod, for the frame kernels' coef = 0.5: every unit vector has
// max|d_i| >= 0.577; a direction that collapsed (the light at 1e30: normalize() returns 0, :44-46 make it +-EPS per axis, and
// the reference's first iteration IS an LOD hit) starts at the root.
// oracle/vrc_walk_model.c restates this on the CPU and tests/test_walk_model.py holds it to the oracle ray by ray.
__device__ __forceinline__ uint32_t start_scale_next_to(float ox, float oy, float oz, float hx, float hy, float hz, uint32_t hit_scale)
{
    const uint32_t a = f2u(ox), b = f2u(oy), c = f2u(oz);
    const uint32_t db = (a ^ f2u(hx)) | (b ^ f2u(hy)) | (c ^ f2u(hz));
    const uint32_t mb = ((a - 8u) ^ (a + 8u)) | ((b - 8u) ^ (b + 8u)) | ((c - 8u) ^ (c + 8u));
    const uint32_t bits = db | (mb + mb);
    const int s = 31 - __clz((int)bits);                  // -1 for no bit at all
    return (uint32_t)max(s, (int)hit_scale);
}

// ... for a ray cast with LOD coefficient 0.5 and bias 0; (dx, dy, dz) = the direction as ray_setup conditioned it
__device__ __forceinline__ uint32_t start_scale_next_to_lod(float ox, float oy, float oz, float hx, float hy, float hz, uint32_t hit_scale,
                                                            float dx, float dy, float dz)
{
    const uint32_t s0 = start_scale_next_to(ox, oy, oz, hx, hy, hz, hit_scale);
    return fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) > 0.55f ? s0 : (uint32_t)(SVO_MAX - 1);
}

// The rows of one lane's stack column copied to another column: a walk that starts below the root on the copy leaves the
// original, the path to the previous hit, to the next ray that starts from the same hit.  The two stacks' rows are
// interleaved (STRIDE = 2 x 256 entries from one row to the next, dst = src + 256), so every address is the lane's base plus
// a constant: LDS reads and writes with immediate offsets, no address arithmetic.
template <int STRIDE>
__device__ __forceinline__ void copy_stack(stack_t* dst, const stack_t* src, int depth)
{
    // three rows per LDS round trip (the reads of a group are issued together), then the one or two that are left; plain
    // counted loops: unrolled with `i < depth` tests the compiler keeps a mask per test in SGPRs, which the kernels do not have
#pragma unroll 1
    for (int g = depth / 3; g > 0; --g) {
        const stack_t a = src[0], b = src[STRIDE], c = src[2 * STRIDE];
        dst[0] = a; dst[STRIDE] = b; dst[2 * STRIDE] = c;
        src += 3 * STRIDE; dst += 3 * STRIDE;
    }
#pragma unroll 1
    for (int g = depth % 3; g > 0; --g) {
        dst[0] = src[0];
        src += STRIDE; dst += STRIDE;
    }
}

// The same idea for camera rays (main.cpp:149: every ray of a frame starts at, or with a lens within the aperture of, the
// camera position): the path from the root to the cell around the camera, followed once per workgroup by position --
// at every scale the child around the point (child_shift = the inverted position bits, whatever the ray's direction: the
// mirrored frame inverts them once more) must exist and be no leaf, and the point must keep the margin of
// start_scale_next_to from the planes of the next scale.  path[scale - depth_off] = parent at that scale, for the scales from
// the returned one up to 22; a camera ray then starts like a ray next to a hit at the camera position with that scale.
//
// A coordinate that is EXACTLY 1.5, the centre plane of the cube -- the reference's default camera sits at size / 2 on x and z
// (main.cpp:51: (256, 200, 256) at 512^3) -- fails that margin at every scale, but needs none: there the first child choice (:66-68) is
// `1.5 * t_coef - t_offset > t_min` with t_offset = fl(1.5 * t_coef) (a mirrored axis: fl(3 t_coef) - fl(1.5 t_coef) =
// 2 fl(1.5 t_coef) - fl(1.5 t_coef), the same number): exactly 0 > 0, false.  In the walk's mirrored frame every ray takes the
// half BELOW 1.5, the half it travels into, and from there on the origin is the upper face of its cell at every level: the
// centre planes are half a cell away, the exits (lower corners) a whole one.  So with `centre_planes` such an axis is left out
// of the margin test and its coordinate counts as 1.5 - 1 ulp in the mirrored frame: the cells just below 1.5 for a ray towards
// -axis, just above for a ray towards +axis -- one path per `variant` (bit i: towards +axis i), 8 of them.
// (oracle/vrc_walk_model.c: orc_model_camera_paths; tests/test_walk_model.py::test_camera_rays_start_on_the_camera_path.)
__device__ __forceinline__ uint32_t camera_path(const uint2* __restrict__ nodes, int depth, float cx, float cy, float cz,
                                                bool centre_planes, uint32_t variant, stack_t* path)
{
    uint32_t a = f2u(cx), b = f2u(cy), c = f2u(cz);
    const bool px = centre_planes && cx == 1.5f, py = centre_planes && cy == 1.5f, pz = centre_planes && cz == 1.5f;
    const uint32_t mb = (px ? 0u : (a - 8u) ^ (a + 8u)) | (py ? 0u : (b - 8u) ^ (b + 8u)) | (pz ? 0u : (c - 8u) ^ (c + 8u));
    // all three in [1, 2): sign 0, exponent 127 (the margin test covers the ends of the interval)
    const bool inside = ((a >> 23) == 127u) && ((b >> 23) == 127u) && ((c >> 23) == 127u);
    if (px && !(variant & 1u)) a -= 1u;
    if (py && !(variant & 2u)) b -= 1u;
    if (pz && !(variant & 4u)) c -= 1u;
    const int depth_off = SVO_MAX - depth;
    // (copy_camera_path moves the rows in groups of three from the top: no deeper than the groups that fit the column)
    const int lowest = max(max(31 - __clz((int)(mb + mb)), depth_off), SVO_MAX - 3 * (depth / 3));
    int s = SVO_MAX - 1;
    uint32_t parent = 0u;
    if (inside) {
        while (s > lowest) {
            const uint2 nd = nodes[parent];
            const uint32_t cs = 7u ^ (((a >> s) & 1u) | (((b >> s) & 1u) << 1) | (((c >> s) & 1u) << 2));
            const uint32_t masks = nd.x >> cs;
            if ((masks & 0x10100u) != 0x100u) break;      // no child (the walk advances here) or a leaf (it ends here)
            path[s - depth_off] = parent;
            parent += nd.y + cs;
            --s;
        }
    }
    path[s - depth_off] = parent;
    return (uint32_t)s;
}

// the rows of a camera path into a lane's stack column (before every camera ray: walks overwrite them), in groups of three
// from the top row down to `lowest_scale` (what a group carries below a path's own scale is never read)
template <int STRIDE>
__device__ __forceinline__ void copy_camera_path(stack_t* dst, const stack_t* path, int depth, uint32_t lowest_scale)
{
    dst += (depth - 3) * STRIDE; path += depth - 3;
#pragma unroll 1
    for (int g = lowest_scale < 22u ? (SVO_MAX - (int)lowest_scale + 2) / 3 : 0; g > 0; --g) {     // paths that end at the root: nothing to copy
        const stack_t a = path[0], b = path[1], c = path[2];
        dst[0] = a; dst[STRIDE] = b; dst[2 * STRIDE] = c;
        dst -= 3 * STRIDE; path -= 3;
    }
}

// Turns the state ray_setup left (start at the root) into a start at scale s0 < 22 next to the previous hit: the position
// is the ray's origin in the mirrored frame of the walk (:49-52; 3 - x is exact for x in [1, 2)); ray_walk<.., BELOW> derives
// the rest (parent, scale_f, cell, child bits) the way a pop does.
__device__ __forceinline__ void ray_start_below(Ray& r, uint32_t s0, float ox, float oy, float oz)
{
#if !(defined(VRC_WALK_CPP) || defined(VRC_PROBE) || defined(VRC_NO_START_BELOW))
    if (s0 < 22u) {
        r.px = (r.mirror & 1u) ? ox : 3.0f - ox;
        r.py = (r.mirror & 2u) ? oy : 3.0f - oy;
        r.pz = (r.mirror & 4u) ? oz : 3.0f - oz;
        r.scale = (int)s0;
    }
#endif
}

// ray_start_below for a camera ray (origin = the camera position): an axis on the cube's centre plane (camera_path) counts as
// 1.5 - 1 ulp in the mirrored frame, whichever way the ray goes
__device__ __forceinline__ void ray_start_at_camera(Ray& r, uint32_t s0, float ox, float oy, float oz)
{
    ray_start_below(r, s0, ox, oy, oz);
#if !(defined(VRC_WALK_CPP) || defined(VRC_PROBE) || defined(VRC_NO_START_BELOW))
    if (s0 < 22u) {
        const float below = u2f(f2u(1.5f) - 1u);
        if (ox == 1.5f) r.px = below;
        if (oy == 1.5f) r.py = below;
        if (oz == 1.5f) r.pz = below;
    }
#endif
}

// :73 for the iterations a ray that starts below the root does not execute (call before the walk: r.scale is its start scale)
__device__ __forceinline__ uint32_t book_not_executed(unsigned long long* ctr, bool active, const Ray& r)
{
    const uint32_t not_executed = (uint32_t)(SVO_MAX - 1 - r.scale);
    if (active && not_executed)
        __hip_atomic_fetch_add(ctr + VRC_CTR_NOT_EXECUTED, (unsigned long long)not_executed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    return active ? not_executed : 0u;
}

// ray_run_total for a ray that may start below the root (ray_start_below)
template <int STRIDE, int LOD_MODE>
__device__ __forceinline__ void ray_run_total_below(Ray& r, bool active, const uint2* __restrict__ nodes, stack_t* stk,
                                                    int depth, float coef, float bias, unsigned long long* ctr)
{
#if defined(VRC_WALK_CPP) || defined(VRC_PROBE) || defined(VRC_NO_START_BELOW)
    ray_run_total<STRIDE, LOD_MODE>(r, active, nodes, stk, depth, coef, bias, ctr);     // from the root: the same walk
#else
    static_assert(STRIDE == 256 || STRIDE == 512, "vrc_walk.h addresses LDS stack rows of 256 lanes, 1 or 2 KiB apart");
    book_not_executed(ctr, active, r);
    const uint32_t wave_steps = ray_walk<LOD_MODE, false, true, STRIDE == 512>(r, active, nodes, stk, depth, coef, bias);
    book_walk(ctr, wave_steps);
#endif
}

struct Hit {
    float px, py, pz;     // position
    float nx, ny, nz;     // normal
    float u, v;           // voxel_coord
    float distance;
    uint32_t hit;         // kind | shift<<8 | scale<<16
    uint32_t node;
    uint32_t complexity;
};

// lsvo.hpp:148-169 epilogue.  (ox..) = ray origin, (dx..) = conditioned direction.
__device__ __forceinline__ void ray_finish(const Ray& r, int depth, float ox, float oy, float oz,
                                           float dx, float dy, float dz, Hit& out)
{
    out.px = out.py = out.pz = 0.0f;
    out.nx = out.ny = out.nz = 0.0f;
    out.u = out.v = 0.0f;
    out.distance = 0.0f;
    out.hit = 0u; out.node = 0u;
    out.complexity = r.complexity;
    if (!r.kind) return;
    const float EPS = 1.0f / (float)(1 << SVO_MAX);
    const float SVO_SIZE = (float)(1 << depth);
    // :149  -sign(d) * (float(normal&1), float(normal&2), float(normal&4)); d != 0 after conditioning
    out.nx = ((dx > 0.0f) ? -1.0f : ((dx < 0.0f) ? 1.0f : -0.0f)) * (float)(r.normal & 1u);
    out.ny = ((dy > 0.0f) ? -1.0f : ((dy < 0.0f) ? 1.0f : -0.0f)) * (float)(r.normal & 2u);
    out.nz = ((dz > 0.0f) ? -1.0f : ((dz < 0.0f) ? 1.0f : -0.0f)) * (float)(r.normal & 4u);
    float qx = r.px, qy = r.py, qz = r.pz;
    if ((r.mirror & 1u) == 0u) qx = 3.0f - r.scale_f - qx;               // :151-153
    if ((r.mirror & 2u) == 0u) qy = 3.0f - r.scale_f - qy;
    if ((r.mirror & 4u) == 0u) qz = 3.0f - r.scale_f - qz;
    out.distance = r.t_min;                                              // :155
    out.px = smin(smax(ox + r.t_min * dx, qx + EPS), qx + r.scale_f - EPS);  // :156-158
    out.py = smin(smax(oy + r.t_min * dy, qy + EPS), qy + r.scale_f - EPS);
    out.pz = smin(smax(oz + r.t_min * dz, qz + EPS), qz + r.scale_f - EPS);
    if (out.nx != 0.0f) {                                                // :160-168
        out.u = fracf_(out.pz * SVO_SIZE); out.v = fracf_(out.py * SVO_SIZE);
    } else if (out.ny != 0.0f) {
        out.u = fracf_(out.px * SVO_SIZE); out.v = fracf_(out.pz * SVO_SIZE);
    } else if (out.nz != 0.0f) {
        out.u = fracf_(out.px * SVO_SIZE); out.v = fracf_(out.py * SVO_SIZE);
    }
    out.hit = r.kind | (r.shift << 8) | ((uint32_t)(r.scale & 0xff) << 16);
    out.node = r.parent_id;
}

}  // namespace vrc
