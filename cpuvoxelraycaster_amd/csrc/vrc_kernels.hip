// vrc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the
// voxel ray-traversal hot path, plus their launchers.
//
//   k_cast_rays(_nolod)  T1   batch LSVO<N>::castRay (lsvo.hpp:33-172), one lane per ray
//   k_cast_ray_chains    T1   two chained casts, the second started below the root next to the first one's hit
//   k_render_sync*  R0 + S1..S4  the frame kernel: persistent waves take (tile, sample chunk) units from a work queue and run
//                   each stage of a sample -- primary, shadow, GI, GI shadow -- for all 64 lanes together; a wave's tile is
//                   8 x 8 pixels, or 4 x 4 pixels with four samples abreast (render_sync_body's NS)
//   k_grid_cast     G1   Grid3D::castRay (grid_3d.hpp:36-132)
//   k_resolve / k_resolve_shard / k_fill_u32 / k_pack_shard / k_unpack_shards   S4 sink + multi-GPU shard plumbing
//
// Rounds 1-3 carried two more schedulers of the same frame (k_render: persistent lanes running a mixed-stage state machine
// with refill from a global queue; k_render_tile: lockstep primary / shadow, then GI jobs from a per-wave LDS pool filled by
// ballot + prefix-sum compaction -- north_star's wording).  Both lost to the stage-synchronous kernel by 2x in round 1, the
// refill scheme again in round 3's simulation at today's walk cost (tests/tools/refill_study.py: +5 % at its best threshold),
// and were removed in round 4.
// No MFMA anywhere: this is dependent 8-byte gathers, not a contraction.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/vrc.h"
#include "vrc_device.h"
#include "vrc_internal.h"

using namespace vrc;

// ---------------------------------------------------------------------------
// T1: batch per-ray operator
// ---------------------------------------------------------------------------

// LOD = false when the caller passes neither ray_size_coef nor ray_size_bias (both 0): the LOD test
// of lsvo.hpp:82 cannot fire then and is compiled out (see ray_step).
template <bool LOD>
__device__ __forceinline__ void cast_rays_body(const uint2* __restrict__ nodes, int depth, uint64_t n,
                                               const float* __restrict__ org, const float* __restrict__ dir,
                                               const float* __restrict__ coef_a, const float* __restrict__ bias_a,
                                               vrc_hit* __restrict__ out)
{
    extern __shared__ stack_t lds_stack[];  // [depth][blockDim.x]
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ox = org[3 * i + 0], oy = org[3 * i + 1], oz = org[3 * i + 2];
    float dx = dir[3 * i + 0], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
    const float coef = coef_a ? coef_a[i] : 0.0f;
    const float bias = bias_a ? bias_a[i] : 0.0f;
    const uint2 root = nodes[0];
    Ray r;
    bool active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
    stack_t* stk = lds_stack + threadIdx.x;
    ray_run<256, LOD ? 2 : 0>(r, active, nodes, stk, depth, coef, bias);
    Hit h;
    ray_finish(r, depth, ox, oy, oz, dx, dy, dz, h);
    // 48-byte record as three 16-byte stores
    uint4* o = reinterpret_cast<uint4*>(out + i);
    o[0] = make_uint4(f2u(h.px), f2u(h.py), f2u(h.pz), f2u(h.nx));
    o[1] = make_uint4(f2u(h.ny), f2u(h.nz), f2u(h.u), f2u(h.v));
    o[2] = make_uint4(h.hit, h.node, f2u(h.distance), h.complexity);
}

extern "C" __global__ void __launch_bounds__(256)
k_cast_rays(const uint2* __restrict__ nodes, int depth, uint64_t n, const float* __restrict__ org, const float* __restrict__ dir,
            const float* __restrict__ coef_a, const float* __restrict__ bias_a, vrc_hit* __restrict__ out)
{
    cast_rays_body<true>(nodes, depth, n, org, dir, coef_a, bias_a, out);
}

extern "C" __global__ void __launch_bounds__(256)
k_cast_rays_nolod(const uint2* __restrict__ nodes, int depth, uint64_t n, const float* __restrict__ org,
                  const float* __restrict__ dir, vrc_hit* __restrict__ out)
{
    cast_rays_body<false>(nodes, depth, n, org, dir, nullptr, nullptr, out);
}

// Chains of two casts (vrc_cast_ray_chains): ray A from the root; ray B next to A's hit, started below the root on the path A's
// walk left in the lane's stack column -- ray_setup / start_scale_next_to[_lod] / ray_start_below / ray_walk<.., BELOW> exactly as
// the frame kernels use them for their shadow, GI and GI-shadow rays, here with the caller's origins and directions
// (tests/test_gpu_start_below.py puts them k ulps either side of the planes of every scale).
template <bool LOD>
__device__ __forceinline__ void cast_ray_chains_body(const uint2* __restrict__ nodes, int depth, uint64_t n, const float* __restrict__ org_a,
                                                     const float* __restrict__ dir_a, const float* __restrict__ org_b,
                                                     const float* __restrict__ dir_b, float coef_b, vrc_hit* __restrict__ out_a,
                                                     vrc_hit* __restrict__ out_b, uint32_t* __restrict__ not_executed)
{
    extern __shared__ stack_t lds_stack[];  // [depth][blockDim.x]
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    stack_t* stk = lds_stack + threadIdx.x;
    const uint2 root = nodes[0];
    float ox = org_a[3 * i + 0], oy = org_a[3 * i + 1], oz = org_a[3 * i + 2];
    float dx = dir_a[3 * i + 0], dy = dir_a[3 * i + 1], dz = dir_a[3 * i + 2];
    Ray r;
    bool active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
    ray_run<256, 0>(r, active, nodes, stk, depth, 0.0f, 0.0f);
    Hit h;
    ray_finish(r, depth, ox, oy, oz, dx, dy, dz, h);
    uint4* o = reinterpret_cast<uint4*>(out_a + i);
    o[0] = make_uint4(f2u(h.px), f2u(h.py), f2u(h.pz), f2u(h.nx));
    o[1] = make_uint4(f2u(h.ny), f2u(h.nz), f2u(h.u), f2u(h.v));
    o[2] = make_uint4(h.hit, h.node, f2u(h.distance), h.complexity);
    ox = org_b[3 * i + 0]; oy = org_b[3 * i + 1]; oz = org_b[3 * i + 2];
    dx = dir_b[3 * i + 0]; dy = dir_b[3 * i + 1]; dz = dir_b[3 * i + 2];
    active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
    uint32_t s0 = 22u;
    if (h.hit)
        s0 = LOD ? start_scale_next_to_lod(ox, oy, oz, h.px, h.py, h.pz, (h.hit >> 16) & 0xffu, dx, dy, dz)
                 : start_scale_next_to(ox, oy, oz, h.px, h.py, h.pz, (h.hit >> 16) & 0xffu);
    ray_start_below(r, s0, ox, oy, oz);
    const uint32_t skipped = active ? (uint32_t)(SVO_MAX - 1 - r.scale) : 0u;
    ray_run<256, LOD ? 1 : 0, true>(r, active, nodes, stk, depth, coef_b, 0.0f);
    Hit g;
    ray_finish(r, depth, ox, oy, oz, dx, dy, dz, g);
    o = reinterpret_cast<uint4*>(out_b + i);
    o[0] = make_uint4(f2u(g.px), f2u(g.py), f2u(g.pz), f2u(g.nx));
    o[1] = make_uint4(f2u(g.ny), f2u(g.nz), f2u(g.u), f2u(g.v));
    o[2] = make_uint4(g.hit, g.node, f2u(g.distance), g.complexity);
    if (not_executed) not_executed[i] = skipped;
}

extern "C" __global__ void __launch_bounds__(256)
k_cast_ray_chains(const uint2* __restrict__ nodes, int depth, uint64_t n, const float* __restrict__ org_a, const float* __restrict__ dir_a,
                  const float* __restrict__ org_b, const float* __restrict__ dir_b, float coef_b, vrc_hit* __restrict__ out_a,
                  vrc_hit* __restrict__ out_b, uint32_t* __restrict__ not_executed)
{
    cast_ray_chains_body<true>(nodes, depth, n, org_a, dir_a, org_b, dir_b, coef_b, out_a, out_b, not_executed);
}

extern "C" __global__ void __launch_bounds__(256)
k_cast_ray_chains_nolod(const uint2* __restrict__ nodes, int depth, uint64_t n, const float* __restrict__ org_a, const float* __restrict__ dir_a,
                        const float* __restrict__ org_b, const float* __restrict__ dir_b, vrc_hit* __restrict__ out_a,
                        vrc_hit* __restrict__ out_b, uint32_t* __restrict__ not_executed)
{
    cast_ray_chains_body<false>(nodes, depth, n, org_a, dir_a, org_b, dir_b, 0.0f, out_a, out_b, not_executed);
}

// ---------------------------------------------------------------------------
// G1: dense grid DDA
// ---------------------------------------------------------------------------

extern "C" __global__ void __launch_bounds__(256)
k_grid_cast(const uint8_t* __restrict__ cells, int X, int Y, int Z, uint64_t n,
            const float* __restrict__ org, const float* __restrict__ dir, vrc_hit* __restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ox = org[3 * i + 0], oy = org[3 * i + 1], oz = org[3 * i + 2];
    const float dx = dir[3 * i + 0], dy = dir[3 * i + 1], dz = dir[3 * i + 2];
    // grid_3d.hpp:42-64
    const float t_dx = fabsf(1.0f / dx), t_dy = fabsf(1.0f / dy), t_dz = fabsf(1.0f / dz);
    const int step_x = dx < 0 ? -1 : 1, step_y = dy < 0 ? -1 : 1, step_z = dz < 0 ? -1 : 1;
    const int dir_x = step_x > 0 ? 1 : 0, dir_y = step_y > 0 ? 1 : 0, dir_z = step_z > 0 ? 1 : 0;
    int cell_x = (int)ox, cell_y = (int)oy, cell_z = (int)oz;
    float t_max_x = ((float)(cell_x + dir_x) - ox) / dx;
    float t_max_y = ((float)(cell_y + dir_y) - oy) / dy;
    float t_max_z = ((float)(cell_z + dir_z) - oz) / dz;
    uint32_t hit_side = 0, iter = 0, hit = 0, node = 0;
    float hx = 0, hy = 0, hz = 0, nx = 0, ny = 0, nz = 0, u = 0, v = 0, dist = 0;
    uint32_t complexity = 0;
    // :70
    while (cell_x >= 0 && cell_y >= 0 && cell_z >= 0 && cell_x < X && cell_y < Y && cell_z < Z && iter < 2048u) {
        float t_max_min;
        ++iter;
        if (t_max_x < t_max_y) {  // :73-100
            if (t_max_x < t_max_z) { t_max_min = t_max_x; t_max_x += t_dx; cell_x += step_x; hit_side = 0; }
            else                   { t_max_min = t_max_z; t_max_z += t_dz; cell_z += step_z; hit_side = 2; }
        } else {
            if (t_max_y < t_max_z) { t_max_min = t_max_y; t_max_y += t_dy; cell_y += step_y; hit_side = 1; }
            else                   { t_max_min = t_max_z; t_max_z += t_dz; cell_z += step_z; hit_side = 2; }
        }
        if (cell_x >= 0 && cell_y >= 0 && cell_z >= 0 && cell_x < X && cell_y < Y && cell_z < Z) {  // :102
            const int64_t idx = ((int64_t)cell_x * Y + cell_y) * Z + cell_z;
            if (cells[idx] != 0) {
                hx = ox + t_max_min * dx; hy = oy + t_max_min * dy; hz = oz + t_max_min * dz;
                hit = 1u | (hit_side << 8); node = (uint32_t)idx;
                if (hit_side == 0)      { nx = (float)(-step_x); u = 1.0f - fracf_(hz); v = fracf_(hy); }
                else if (hit_side == 1) { ny = (float)(-step_y); u = fracf_(hx); v = fracf_(hz); }
                else                    { nz = (float)(-step_z); u = fracf_(hx); v = fracf_(hy); }
                dist = t_max_min; complexity = iter;
                break;
            }
        }
    }
    uint4* o = reinterpret_cast<uint4*>(out + i);
    o[0] = make_uint4(f2u(hx), f2u(hy), f2u(hz), f2u(nx));
    o[1] = make_uint4(f2u(ny), f2u(nz), f2u(u), f2u(v));
    o[2] = make_uint4(hit, node, f2u(dist), complexity);
}

// ---------------------------------------------------------------------------
// R0 + S1..S4: colour arithmetic and pixel selection shared by the frame kernels
// ---------------------------------------------------------------------------

namespace {

// utils.cpp:43-48 on a packed 0x00BBGGRR colour
__device__ __forceinline__ uint32_t color_mult(uint32_t c, float f)
{
    const uint32_t r = (uint32_t)smin(255.0f, (float)(c & 0xffu) * f);
    const uint32_t g = (uint32_t)smin(255.0f, (float)((c >> 8) & 0xffu) * f);
    const uint32_t b = (uint32_t)smin(255.0f, (float)((c >> 16) & 0xffu) * f);
    return r | (g << 8) | (b << 16);
}
// utils.cpp:35-40
__device__ __forceinline__ uint32_t color_add_sat(uint32_t a, uint32_t b)
{
    const uint32_t r = min(255u, (a & 0xffu) + (b & 0xffu));
    const uint32_t g = min(255u, ((a >> 8) & 0xffu) + ((b >> 8) & 0xffu));
    const uint32_t bl = min(255u, ((a >> 16) & 0xffu) + ((b >> 16) & 0xffu));
    return r | (g << 8) | (bl << 16);
}
// utils.cpp:67-75
__device__ __forceinline__ float clampf_(float v, float mn, float mx)
{
    if (v > mx) v = mx; else if (v < mn) v = mn;
    return v;
}

// main.cpp:131-132,140-143 + shard ownership (same predicate as the oracle's pixel_selected)
__device__ __forceinline__ bool pixel_selected(const vrc_frame_params& p, uint32_t W, uint32_t H, uint32_t x, uint32_t y)
{
    if (p.checker_parity >= 0) {
        const uint32_t aw = W / 4u, ah = H / 4u;
        if (aw == 0u || ah == 0u || x >= 4u * aw || y >= 4u * ah) return false;
        const uint32_t y0 = (y / ah) * ah;
        if (((y - y0) % 2u) != ((x + (uint32_t)p.checker_parity) % 2u)) return false;
    }
    return true;
}

}  // namespace

// ---------------------------------------------------------------------------
// R0 + S1..S4, stage-synchronous per tile ("sync" kernel)
// ---------------------------------------------------------------------------
//
// One wave owns one 8x8 pixel tile at a time; for every sample all 64 lanes run
// the same stage together: camera ray -> walk -> shade -> shadow ray(s) -> walk ->
// GI ray -> walk -> GI shadow ray -> walk -> sink.  Lanes whose pixel-sample has
// ended (miss, GI miss) idle until the tile's stage is over.  Everything lives in
// registers and the per-wave LDS stack; no job pool, no state machine, so the
// kernel needs few registers (high occupancy) and every transition (ray
// generation, the three IEEE divides of ray_setup, finish, shading) runs at full
// lane utilisation.  Measured against the two refill schedulers of rounds 1-3 on C3
// it was the fastest: what refill machinery saves in idle lanes it loses in transition
// code executed at partial utilisation and in occupancy.

// entries from one row of a stack to the next: the sync kernels keep two stacks with interleaved rows (render_sync_body)
#define VRC_SYNC_ROW (2 * VRC_RENDER_BLOCK)

#ifndef VRC_SYNC_S4_WAVES
#define VRC_SYNC_S4_WAVES 6          // waves per SIMD the samples-abreast kernels are built for (pinhole / lens)
#endif
#ifndef VRC_SYNC_S4_WAVES_LENS
#define VRC_SYNC_S4_WAVES_LENS 6
#endif
#ifndef VRC_SYNC_MIN_WAVES
#define VRC_SYNC_MIN_WAVES 6
#endif

#ifdef VRC_PROBE
__device__ __forceinline__ uint32_t probe_wave_max(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off));
    return v;
}
#define VRC_PROBE_WALK(r, k) probe_c##k = (r).complexity;
#else
#define VRC_PROBE_WALK(r, k)
#endif

// raycaster.hpp:169-207 getGlobalIllumination for one pixel-sample (+ the nested second bounce of the extension): GI
// ray from the point (bp, bn), cast with LOD coefficient 0.5; if it hits, a shadow ray from the GI hit towards the
// light.  Called by all lanes of a wave whose pixel-sample has a primary hit (`in exec`); the two walks are wave-wide.
// `stk_hit` holds the path to the point's hit (the stack its walk left); the chain's walks use the other stack (its rows lie
// VRC_RENDER_BLOCK entries further): the GI ray starts below the root on a copy of that path, the GI shadow ray on what the
// GI walk left (vrc_device.h, ray_start_below).
template <bool ONE_BOUNCE>   // true: gi_bounces == 1 is known (the reference's semantics), the chain is straight-line code
__device__ __forceinline__ float gi_chain(const uint2* __restrict__ nodes, stack_t* const stk_hit, const uint32_t base_scale, const bool from_root,
                                          const uint2 root, const int depth,
                                          const float n_normalizer, const float lx, const float ly, const float lz,
                                          const uint32_t gi_bounces_rt, const uint32_t seed, const uint32_t fr, const uint32_t pix,
                                          float base_px, float base_py, float base_pz, float base_nx, float base_ny, float base_nz,
                                          unsigned long long* ctr)
{
    const uint32_t gi_bounces = ONE_BOUNCE ? 1u : gi_bounces_rt;
    stack_t* const stk = stk_hit + VRC_RENDER_BLOCK;
    Ray r;
    bool active;
    float ox, oy, oz, dx, dy, dz;
    float acc[2] = {0.0f, 0.0f}, dgi[2] = {0.0f, 0.0f};
    uint32_t draw = 2u, level = 0u;
    bool chain = true;
    while (chain) {
        ox = base_px + base_nx * n_normalizer;                           // :174
        oy = base_py + base_ny * n_normalizer;
        oz = base_pz + base_nz * n_normalizer;
        const float c1 = get_rand(counter_rand(seed, fr, pix, draw), -1000.0f, 1000.0f);
        const float c2 = get_rand(counter_rand(seed, fr, pix, draw + 1u), -1000.0f, 1000.0f);
        draw += 2u;
        float qx = 0.0f, qy = 0.0f, qz = 0.0f;
        if (base_nx != 0.0f)      { qy = c1; qz = c2; }
        else if (base_ny != 0.0f) { qx = c1; qz = c2; }
        else if (base_nz != 0.0f) { qx = c1; qy = c2; }
        dx = (base_nx + qx) * n_normalizer;                              // :192
        dy = (base_ny + qy) * n_normalizer;
        dz = (base_nz + qz) * n_normalizer;
        normalize3(dx, dy, dz);
        const float dot_gi = dot3(dx, dy, dz, base_nx, base_ny, base_nz);   // :193
        if (level == 0u) dgi[0] = dot_gi; else dgi[1] = dot_gi;
        active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
        if (level == 0u) {     // (a deeper level starts at the root: the GI shadow walk has used the stack since its hit)
            if (!from_root) copy_stack<VRC_SYNC_ROW>(stk, stk_hit, depth);
            ray_start_below(r, (from_root ? 22u : start_scale_next_to_lod(ox, oy, oz, base_px, base_py, base_pz, base_scale, dx, dy, dz)), ox, oy, oz);
        }
        ray_run_total_below<VRC_SYNC_ROW, 1>(r, active, nodes, stk, depth, 0.5f, 0.0f, ctr);   // :194
        Hit g;
        ray_finish(r, depth, ox, oy, oz, dx, dy, dz, g);
        chain = false;
        if (g.hit) {                                                     // :195-203
            ox = g.px + g.nx * n_normalizer;
            oy = g.py + g.ny * n_normalizer;
            oz = g.pz + g.nz * n_normalizer;
            dx = lx - ox; dy = ly - oy; dz = lz - oz;
            normalize3(dx, dy, dz);
            const float dotl = dot3(g.nx, g.ny, g.nz, dx, dy, dz);         // :200
            const float contrib = 1000000.0f * smin(0.5f, smax(0.0f, dotl) * dot_gi);   // :201
            active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
            ray_start_below(r, (from_root ? 22u : start_scale_next_to_lod(ox, oy, oz, g.px, g.py, g.pz, (g.hit >> 16) & 0xffu, dx, dy, dz)), ox, oy, oz);
            ray_run_total_below<VRC_SYNC_ROW, 1>(r, active, nodes, stk, depth, 0.5f, 0.0f, ctr);   // :198
            if (!r.kind) { if (level == 0u) acc[0] += contrib; else acc[1] += contrib; }
            if (level + 1u < gi_bounces) {
                base_px = g.px; base_py = g.py; base_pz = g.pz;
                base_nx = g.nx; base_ny = g.ny; base_nz = g.nz;
                level += 1u;
                chain = true;
            }
        }
    }
    // :206, innermost level first: gi = max(0, acc0 [+ dot_gi0 * max(0, acc1)])
    if (level == 0u) return smax(0.0f, acc[0] / 1.0f);
    acc[0] += dgi[0] * smax(0.0f, acc[1] / 1.0f);
    return smax(0.0f, acc[0] / 1.0f);
}

// The 2-bounce extension as straight-line code (the kernels built for it always run exactly two levels: vrc_render_frame
// rejects gi_bounces > 2, launch_render sends gi_bounces <= 1 to the one-bounce kernels).  Same operations in the same
// order as gi_chain<false> with gi_bounces = 2 -- level 0 from the primary hit with draws 2, 3, level 1 from the GI hit with
// draws 4, 5, gi = max(0, acc0 + dot_gi0 * max(0, acc1)) -- without the loop-carried level / draw / base state.
__device__ __forceinline__ float gi_two_levels(const uint2* __restrict__ nodes, stack_t* const stk_hit, const uint32_t base_scale, const bool from_root,
                                               const uint2 root, const int depth,
                                               const float n_normalizer, const float lx, const float ly, const float lz,
                                               const uint32_t seed, const uint32_t fr, const uint32_t pix,
                                               const float b0px, const float b0py, const float b0pz,
                                               const float b0nx, const float b0ny, const float b0nz, unsigned long long* ctr)
{
    float acc0 = 0.0f, acc1 = 0.0f, dgi0 = 0.0f;
    stack_t* const stk = stk_hit + VRC_RENDER_BLOCK;
    Ray r;
    bool active;
    float ox, oy, oz, dx, dy, dz;
    {   // ---- level 0
        ox = b0px + b0nx * n_normalizer;                                 // :174
        oy = b0py + b0ny * n_normalizer;
        oz = b0pz + b0nz * n_normalizer;
        const float c1 = get_rand(counter_rand(seed, fr, pix, 2u), -1000.0f, 1000.0f);
        const float c2 = get_rand(counter_rand(seed, fr, pix, 3u), -1000.0f, 1000.0f);
        float qx = 0.0f, qy = 0.0f, qz = 0.0f;
        if (b0nx != 0.0f)      { qy = c1; qz = c2; }
        else if (b0ny != 0.0f) { qx = c1; qz = c2; }
        else if (b0nz != 0.0f) { qx = c1; qy = c2; }
        dx = (b0nx + qx) * n_normalizer;                                 // :192
        dy = (b0ny + qy) * n_normalizer;
        dz = (b0nz + qz) * n_normalizer;
        normalize3(dx, dy, dz);
        dgi0 = dot3(dx, dy, dz, b0nx, b0ny, b0nz);                        // :193
        active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
        if (!from_root) copy_stack<VRC_SYNC_ROW>(stk, stk_hit, depth);
        ray_start_below(r, (from_root ? 22u : start_scale_next_to_lod(ox, oy, oz, b0px, b0py, b0pz, base_scale, dx, dy, dz)), ox, oy, oz);
        ray_run_total_below<VRC_SYNC_ROW, 1>(r, active, nodes, stk, depth, 0.5f, 0.0f, ctr);   // :194
    }
    Hit g;
    ray_finish(r, depth, ox, oy, oz, dx, dy, dz, g);
    if (g.hit) {                                                         // :195-203
        ox = g.px + g.nx * n_normalizer;
        oy = g.py + g.ny * n_normalizer;
        oz = g.pz + g.nz * n_normalizer;
        dx = lx - ox; dy = ly - oy; dz = lz - oz;
        normalize3(dx, dy, dz);
        const float dotl = dot3(g.nx, g.ny, g.nz, dx, dy, dz);             // :200
        const float contrib = 1000000.0f * smin(0.5f, smax(0.0f, dotl) * dgi0);   // :201
        active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
        ray_start_below(r, (from_root ? 22u : start_scale_next_to_lod(ox, oy, oz, g.px, g.py, g.pz, (g.hit >> 16) & 0xffu, dx, dy, dz)), ox, oy, oz);
        ray_run_total_below<VRC_SYNC_ROW, 1>(r, active, nodes, stk, depth, 0.5f, 0.0f, ctr);   // :198
        if (!r.kind) acc0 += contrib;
        // ---- level 1: the same from the GI hit
        const float b1nx = g.nx, b1ny = g.ny, b1nz = g.nz;
        ox = g.px + b1nx * n_normalizer;
        oy = g.py + b1ny * n_normalizer;
        oz = g.pz + b1nz * n_normalizer;
        const float c1 = get_rand(counter_rand(seed, fr, pix, 4u), -1000.0f, 1000.0f);
        const float c2 = get_rand(counter_rand(seed, fr, pix, 5u), -1000.0f, 1000.0f);
        float qx = 0.0f, qy = 0.0f, qz = 0.0f;
        if (b1nx != 0.0f)      { qy = c1; qz = c2; }
        else if (b1ny != 0.0f) { qx = c1; qz = c2; }
        else if (b1nz != 0.0f) { qx = c1; qy = c2; }
        dx = (b1nx + qx) * n_normalizer;
        dy = (b1ny + qy) * n_normalizer;
        dz = (b1nz + qz) * n_normalizer;
        normalize3(dx, dy, dz);
        const float dgi1 = dot3(dx, dy, dz, b1nx, b1ny, b1nz);
        active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
        ray_run_total<VRC_SYNC_ROW, 1>(r, active, nodes, stk, depth, 0.5f, 0.0f, ctr);
        Hit g1;
        ray_finish(r, depth, ox, oy, oz, dx, dy, dz, g1);
        if (g1.hit) {
            ox = g1.px + g1.nx * n_normalizer;
            oy = g1.py + g1.ny * n_normalizer;
            oz = g1.pz + g1.nz * n_normalizer;
            dx = lx - ox; dy = ly - oy; dz = lz - oz;
            normalize3(dx, dy, dz);
            const float dotl1 = dot3(g1.nx, g1.ny, g1.nz, dx, dy, dz);
            const float contrib1 = 1000000.0f * smin(0.5f, smax(0.0f, dotl1) * dgi1);
            active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
            ray_start_below(r, (from_root ? 22u : start_scale_next_to_lod(ox, oy, oz, g1.px, g1.py, g1.pz, (g1.hit >> 16) & 0xffu, dx, dy, dz)), ox, oy, oz);
            ray_run_total_below<VRC_SYNC_ROW, 1>(r, active, nodes, stk, depth, 0.5f, 0.0f, ctr);
            if (!r.kind) acc1 += contrib1;
        }
        acc0 += dgi0 * smax(0.0f, acc1 / 1.0f);                            // :206, innermost level first
        return smax(0.0f, acc0 / 1.0f);
    }
    return smax(0.0f, acc0 / 1.0f);
}

// ---- Quadrant walks (round 5): the pinhole kernel's sample-invariant walks in a 4 x 4 pixels x 4 samples lane map -----------
// With a pinhole camera a pixel's primary ray and the shadow ray of its hit are the same for every sample, and the reference
// walks them once per sample (raycaster.hpp:131,153).  The plain pinhole kernel does that as n walks of 64 pixels x 1 sample: a
// stage lasts as long as the longest ray of 64 neighbouring pixels.  Here the same 64 x n walks are laid out as 4 x (n / 4)
// walks of one QUADRANT of the tile -- 16 pixels, each in four lanes (lane = pixel-in-quadrant + 16 x sample) --: every ray the
// reference casts is still walked, each by its own lane, but a stage now lasts as long as the longest ray of 16 pixels, and a
// round issues fewer of the loop's blocks (tests/tools/tile_shape.py: primary 1715 -> 1487, shadow 1522 -> 1216 VALU
// instructions per wave-sample on the metric's pose).  Unlike the samples-abreast kernels (NS = 4) the set-up stays hoisted:
// rays are generated ONCE per pixel at full lane utilisation in the 8 x 8 map and handed to the quadrant's lanes with
// ds_bpermute (no VALU, no LDS storage); the four lanes of a pixel walk in step on the pixel's OWN stack column (identical
// pushes to one address), so the path to the primary hit ends up where the GI stages, which keep the 8 x 8 map, expect it.

constexpr uint32_t VRC_QUAD_LDS_BYTES = 0u;                           // (a walk's final state waits in the secondary stack: depth >= 8 rows)
// A walk's pixels: a block of the 8 x 8 tile -- 4 x 4 (each pixel in 4 lanes), 4 x 2 (8 lanes) or 2 x 2 (16 lanes): the more walks a
// pixel's ray needs (samples per unit, x shadow samples), the fewer distinct rays share a walk (C3 pose, VALU instructions per
// tile-sample, primary + shadow: 8 x 8 x 1 3518, 4 x 4 x 4 2926, 2 x 2 x 16 2503; tests/tools/path_schedule.py --quad).
// lg = log2 of the lanes per pixel (2, 3 or 4).  The 8 x 8 map's lane of pixel q of block k:
__device__ __forceinline__ uint32_t quad_lane(uint32_t k, uint32_t q, uint32_t lg)
{
    const uint32_t bw_lg = lg == 4u ? 1u : 2u, bh_lg = lg == 2u ? 2u : 1u;     // block = 2^bw_lg x 2^bh_lg pixels
    const uint32_t kx = k & ((8u >> bw_lg) - 1u), ky = k >> (3u - bw_lg);
    const uint32_t qx = q & ((1u << bw_lg) - 1u), qy = q >> bw_lg;
    return (((ky << bh_lg) + qy) << 3) + (kx << bw_lg) + qx;
}
// how many lanes a pixel's ray gets when it has to be walked n times (n % 4 == 0): log2
__device__ __forceinline__ uint32_t quad_lanes_lg(uint32_t n) { return (n & 15u) == 0u ? 4u : ((n & 7u) == 0u ? 3u : 2u); }

// ctr[0] += wave_steps, ctr[1] += the lanes with `casts` set: book_walk for a call that ALL lanes of the wave make
__device__ __forceinline__ void book_walk_of(unsigned long long* ctr, uint32_t wave_steps, bool casts)
{
    const unsigned long long m = __ballot(casts);
    if ((threadIdx.x & 63u) == 0u) {
        __hip_atomic_fetch_add(ctr + 0, (unsigned long long)wave_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_add(ctr + 1, (unsigned long long)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

// what ray_setup + ray_start_below left in lane `src` of this wave, as this lane's ray; pk = scale | child << 5 | mirror << 8 |
// accepted by ray_setup << 11 | casts a ray << 12 of that lane
__device__ __forceinline__ void quad_gather(Ray& r, bool& casts, bool& active, const Ray& s, const uint32_t spk, const uint32_t src, const uint2 root)
{
    const int a = (int)(src << 2);
#define VRC_QG(f) u2f((uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)f2u(s.f)))
    r.tcx = VRC_QG(tcx); r.tcy = VRC_QG(tcy); r.tcz = VRC_QG(tcz);
    r.tox = VRC_QG(tox); r.toy = VRC_QG(toy); r.toz = VRC_QG(toz);
    r.px = VRC_QG(px); r.py = VRC_QG(py); r.pz = VRC_QG(pz);
    r.t_min = VRC_QG(t_min); r.t_max = VRC_QG(t_max);
#undef VRC_QG
    const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)spk);
    r.scale = (int)(pk & 31u); r.child_offset = (pk >> 5) & 7u; r.mirror = (pk >> 8) & 7u;
    active = ((pk >> 11) & 1u) != 0u; casts = ((pk >> 12) & 1u) != 0u;
    r.scale_f = u2f(((uint32_t)r.scale + 104u) << 23);                // 2^(scale - 23); a start below the root derives its own
    r.h = r.t_max; r.parent_id = 0u; r.normal = 0u; r.complexity = 0u; r.kind = 0u; r.shift = 0u; r.nd = root;
}

__device__ __forceinline__ uint32_t quad_pack(const Ray& r, bool active, bool casts)
{
    return casts ? ((uint32_t)r.scale | (r.child_offset << 5) | (r.mirror << 8) | ((active ? 1u : 0u) << 11) | (1u << 12)) : 0u;
}

// sink of one pixel-sample, raycaster.hpp:77-91: sum in registers (sample mode; one accumulator update per unit) or
// the 0.4 / 0.6 temporal blend into render_image
__device__ __forceinline__ void sink_sample(const FrameArgs& a, const uint32_t pix, const uint32_t color,
                                            uint32_t& acc_r, uint32_t& acc_g, uint32_t& acc_b)
{
    if (a.p.use_samples) { acc_r += color & 0xffu; acc_g += (color >> 8) & 0xffu; acc_b += (color >> 16) & 0xffu; }
    else {
        uint32_t old = reinterpret_cast<uint32_t*>(a.image)[pix];
        const uint32_t alpha = old & 0xff000000u;
        old = color_mult(old & 0xffffffu, 0.4f);
        const uint32_t nw = color_mult(color, 1.0f - 0.4f);
        reinterpret_cast<uint32_t*>(a.image)[pix] = color_add_sat(old, nw) | alpha;
    }
}

// raycaster.hpp:209-240: albedo of a primary hit from the two 16 x 16 tables staged in LDS
__device__ __forceinline__ uint32_t albedo_of(const uint8_t* tex, const Hit& h)
{
    const uint8_t* img = (h.ny != 0.0f) ? tex : tex + 768;
    const float tu = clampf_(h.u, 0.0f, 1.0f), tv = clampf_(h.v, 0.0f, 1.0f);
    uint32_t idx = (uint32_t)(16.0f * tu) + (uint32_t)(16.0f * tv) * 16u;
    idx = min(idx, 255u);
    return (uint32_t)img[idx * 3] | ((uint32_t)img[idx * 3 + 1] << 8) | ((uint32_t)img[idx * 3 + 2] << 16);
}

// PINHOLE: the camera has aperture exactly +0 (the reference's default) -- decided by the host, see vrc_render_frame.
// Then a pixel's camera ray is the same for every sample, and so are its primary hit, albedo and shadow ray(s): a work
// unit sets them up ONCE and walks them once per sample of the unit (every ray the reference casts is still walked and
// counted; only the identical arithmetic around the walks is not repeated).  The per-sample work is the GI pair.
//
// NS: the lane <-> (pixel, sample) map of a wave.  NS = 1: 8 x 8 pixels, the wave loops over the samples of its unit.  NS = 4
// ("samples abreast"): 4 x 4 pixels x 4 samples -- lane = pixel + 16 * (sample % 4) -- for sample-mode frames whose spp is a
// multiple of 4 (the host's choice, vrc_api.cpp).  The image cannot tell (counter_rand is keyed by the global pixel and the
// sample, the sums are integers), the walks can: the four lanes of a pixel share the primary hit, so a stage's longest ray is
// the longest of 16 neighbouring pixels' instead of 64 and a round issues fewer of the loop's blocks -- 6-11 % fewer VALU
// instructions in the walks of the C3 frame, 4 % with a lens (tests/tools/tile_shape.py -> profiles/r04/tile_shape.json).
// Every ray the reference casts is still walked and counted, each by its own lane.
// QUAD (pinhole, 8 x 8 map): the sample-invariant walks run quadrant by quadrant, four samples abreast (see quad_gather above);
// for units whose sample count is a multiple of four, without invariant-ray reuse and without the primary-hit capture.
template <bool PINHOLE, bool ONE_BOUNCE, bool FUSED, bool FROM_ROOT = false, uint32_t NS = 1u, bool QUAD = false>
__device__ __forceinline__ void render_sync_body(const FrameArgs& a)
{
    static_assert(!QUAD || (PINHOLE && !FROM_ROOT && NS == 1u), "quadrant walks exist for the pinhole kernels on the 8 x 8 map");
    // [depth][2][256] two stacks, their rows interleaved | 1536 B textures | 4 waves x 4 counters | camera paths (8 x 16 entries + 8) |
    // 4 waves' count of iterations not executed
    extern __shared__ stack_t lds_dyn[];
    stack_t* stk = lds_dyn + threadIdx.x;                              // the path to the primary hit, kept until the next primary walk
    // (secondary rays use the other stack, stk + VRC_RENDER_BLOCK: a copy of that path, then their own)
    uint8_t* tex = reinterpret_cast<uint8_t*>(lds_dyn + 2u * a.depth * VRC_RENDER_BLOCK);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = threadIdx.x; i < 1536u / 4u; i += VRC_RENDER_BLOCK)
        reinterpret_cast<uint32_t*>(tex)[i] = reinterpret_cast<const uint32_t*>(a.tex)[i];
    // this wave's {loop iterations, rays, primary hits, pixel-samples}: booked per walk / per unit by one lane with
    // no-return LDS adds, so that no per-lane counter is live across the walks (the kernel is at its register budget)
    unsigned long long* const ctr = reinterpret_cast<unsigned long long*>(tex + 1536) + 4u * (threadIdx.x >> 6);
    if (lane < 4u) ctr[lane] = 0ull;
    if (lane == 4u) ctr[VRC_CTR_NOT_EXECUTED] = 0ull;
    const uint2* __restrict__ nodes = a.nodes;
    const int depth = (int)a.depth;
    // the paths from the root to the camera's cell (vrc_device.h, camera_path): camera rays start at their ends.  One per
    // combination of direction signs (they differ only for a pinhole camera on the cube's centre planes -- the reference's
    // default pose; with a lens the lens point decides, and a camera on a centre plane starts its rays at the root)
    stack_t* const cam_path = reinterpret_cast<stack_t*>(tex + 1536 + 128);
    if (threadIdx.x < 8u) {
        const float isz = 1.0f / (float)(1u << a.depth);
        cam_path[threadIdx.x * 16u + 15u] = camera_path(nodes, depth, (a.cam.position[0] + 0.0f) * isz + 1.0f, (a.cam.position[1] + 0.0f) * isz + 1.0f,
                                                        (a.cam.position[2] + 0.0f) * isz + 1.0f, PINHOLE, threadIdx.x, cam_path + threadIdx.x * 16u);
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t lowest = 22u;
        for (uint32_t v = 0; v < 8u; ++v) lowest = min(lowest, cam_path[v * 16u + 15u]);
        cam_path[128] = lowest;
    }
    __syncthreads();
    // (FROM_ROOT: the builds behind vrc_renderer_set_walk_from_root, a measurement switch: no ray starts below the root.  A
    // kernel argument instead costs every frame 1 %: one more live SGPR and a select per ray)
    constexpr bool from_root = FROM_ROOT;
    const uint32_t cam_scale = from_root ? 22u : __builtin_amdgcn_readfirstlane(cam_path[128]);     // the lowest scale a camera ray starts at

    const uint2 root = nodes[0];
    const uint32_t W = a.width, H = a.height;
    const float inv_size = 1.0f / (float)(1u << a.depth);                 // main.cpp:82
    const float n_normalizer = inv_size * 0.0078125f * 2.0f;              // raycaster.hpp:171-172
    const float lx = a.p.light_position[0], ly = a.p.light_position[1], lz = a.p.light_position[2];
    const uint32_t shadow_samples = a.p.shadow_samples ? a.p.shadow_samples : (a.p.use_samples ? 4u : 1u);
    const uint32_t gi_bounces = a.p.gi_bounces ? a.p.gi_bounces : 1u;
    const uint32_t spp = a.p.spp ? a.p.spp : 1u;
    // Checkerboard frames (main.cpp:137,143: every other pixel of each swarm area) use tiles of 16 x 8 pixels, of
    // which a lane takes the selected pixel of its pair, so that all 64 lanes of a wave have work
    const bool checker = a.checker_wide != 0u;
    static_assert(NS == 1u || NS == 4u, "lane maps: 8 x 8 pixels, or 4 x 4 pixels x 4 samples");
    constexpr uint32_t TW = NS == 4u ? 4u : 8u;                          // a wave's tile is TW x TW pixels
    const uint32_t sub = NS == 1u ? 0u : lane >> 4;                      // which of the NS samples abreast this lane takes
    const uint32_t tiles_per_row = checker ? (W + 15u) / 16u : (W + TW - 1u) / TW;
    const uint32_t n_tiles = a.n_items >> 6;
    const bool sharded = a.p.row_block && a.p.shard_count > 1u;
    constexpr bool pinhole = PINHOLE;     // see launch_render for the host's predicate
    // (arguments used once per launch or per work unit are read from the kernel-argument segment where they are needed
    // rather than held in SGPRs throughout: the kernel is at its register budget)
    const volatile FrameArgs* ka = (const volatile FrameArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    if (blockIdx.x == 0u && threadIdx.x < VRC_QUEUE_HEADS) {
        uint32_t* const other = ka->queue_other;
        if (other) other[threadIdx.x * VRC_QUEUE_STRIDE] = 0u;
    }
    // Work unit = (tile, chunk of `sample_chunk` samples).  With few tiles per GPU (multi-GPU shards)
    // the host splits a pixel's samples over several units so every wave still gets several units;
    // the accumulators are then updated with integer atomics (exact, order-independent).
    // The last `tail_tiles` tiles may use a smaller chunk (`sample_chunk_tail`): units get shorter towards the end of
    // the launch, when what matters is how long the last ones take.
    const uint32_t chunk_head = a.sample_chunk ? a.sample_chunk : spp;
    const uint32_t cpt_head = (spp + chunk_head - 1u) / chunk_head;
    const uint32_t chunk_tail = a.sample_chunk_tail ? a.sample_chunk_tail : chunk_head;
    const uint32_t cpt_tail = (spp + chunk_tail - 1u) / chunk_tail;
    const uint32_t tail_tiles = min(a.tail_tiles, n_tiles), head_tiles = n_tiles - tail_tiles;
    const uint32_t head_units = head_tiles * cpt_head;
    const uint32_t n_units = head_units + tail_tiles * cpt_tail;

    uint32_t n_hits = 0u, n_pix = 0u;     // per lane; the general kernel only (a lane's samples differ in their hits)

    // Work queue with 8 heads (one 128-byte line each): head q hands out units q, q+8, q+16, ...
    // A wave starts on the head of its workgroup's XCD group (blockIdx % 8: workgroups b and b+8
    // share an XCD -- a speed hint only) and moves on to the next head when one runs dry, so all
    // units are handed out whatever the placement.  One head saturates at ~88 dequeues/us, which a
    // small multi-GPU shard (16 k units in 0.3 ms) would hit.
    uint32_t head = blockIdx.x & 7u, heads_tried = 0u;
#ifdef VRC_PROBE   // diagnostics build only: per-wave start / end wall clock (100 MHz) and unit count into the capture buffer
    const uint64_t probe_t0 = wall_clock64();
    uint32_t probe_units = 0u, probe_last_unit = 0xffffffffu, probe_longest_unit = 0u, probe_iters = 0u;
    uint64_t probe_last_t0 = probe_t0, probe_longest = 0u;
#endif

    for (;;) {
        uint32_t unit = 0u;
        if (lane == 0u) {
            // after the first dry head, look before taking: a dry head then costs a load, not an atomic
            uint32_t* q = a.queue + head * VRC_QUEUE_STRIDE;
            unit = heads_tried ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (unit * 8u + head < n_units) unit = atomicAdd(q, 1u);
        }
        unit = __builtin_amdgcn_readfirstlane(unit) * 8u + head;
        if (unit >= n_units) {
            if (++heads_tried == 8u) break;
            head = (head + 1u) & 7u;
            continue;
        }
#ifdef VRC_PROBE
        ++probe_units;
        const uint64_t probe_t1 = wall_clock64();
        const uint64_t probe_prev_t0 = probe_last_t0;
        if (probe_t1 - probe_last_t0 > probe_longest) { probe_longest = probe_t1 - probe_last_t0; probe_longest_unit = probe_last_unit; }
        probe_last_t0 = probe_t1;
#endif
#ifdef VRC_PROBE
        {   // experiment: per-unit durations at +1 MiB, optional unit order table at +2 MiB (flag at +2 MiB - 4)
            uint32_t* pb = reinterpret_cast<uint32_t*>(a.prim);
            if (probe_last_unit != 0xffffffffu && lane == 0u) {
                pb[(1u << 18) + probe_last_unit] = (uint32_t)(probe_t1 - probe_prev_t0);
                pb[(3u << 18) + probe_last_unit] = probe_iters;
            }
            probe_iters = 0u;
            if (pb[(1u << 19) - 1u]) unit = pb[(1u << 19) + unit];
            probe_last_unit = unit;
        }
#endif
        const bool in_tail = unit >= head_units;
        const uint32_t chunks_per_tile = in_tail ? cpt_tail : cpt_head, chunk = in_tail ? chunk_tail : chunk_head;
        const uint32_t u = in_tail ? unit - head_units : unit;
        const uint32_t tile = (in_tail ? head_tiles : 0u) + u / chunks_per_tile;
        const uint32_t s_begin = (u % chunks_per_tile) * chunk, s_end = min(spp, s_begin + chunk);
        const uint32_t tx = tile % tiles_per_row, ty = tile / tiles_per_row;
        uint32_t px_x = tx * TW + (lane & (TW - 1u));
        uint32_t px_y = ty * TW + ((lane / TW) & (TW - 1u));
        if (sharded) px_y = ((px_y / a.p.row_block) * a.p.shard_count + a.p.shard_index) * a.p.row_block + (px_y % a.p.row_block);
        if (checker) {
            // the pixel of the pair (2c, 2c + 1) that pixel_selected accepts in this row: x = (y - y0 + parity) mod 2
            const uint32_t ah = H / 4u, y0 = ah ? (px_y / ah) * ah : 0u;
            px_x = tx * 16u + 2u * (lane & 7u) + ((px_y - y0 + (uint32_t)a.p.checker_parity) & 1u);
        }
        const bool valid = px_x < W && px_y < H && pixel_selected(a.p, W, H, px_x, px_y);
        const uint32_t pix = px_y * W + px_x;
        uint32_t acc_r = 0u, acc_g = 0u, acc_b = 0u;

        if constexpr (PINHOLE && NS == 1u) {
            // ---- the stages that do not depend on the sample: set up once, walked once per sample of the unit
            const uint32_t n_samples = s_end - s_begin;
            // the reference walks the identical primary and shadow ray again for every sample, and so does this kernel
            // unless the caller allowed reuse: then one walk serves all the samples of the unit (same hit, same image;
            // the counters then count the walks executed, not the reference's)
            const bool reuse = ka->reuse_invariant != 0u;
            const uint32_t n_walks = reuse ? 1u : n_samples, n_shadow_walks = reuse ? 1u : n_samples * shadow_samples;
            bool hit = false;
            uint32_t albedo = 0u;                                              // | scale of the hit << 24
            float light_intensity = 0.0f;
            float hpx = 0.0f, hpy = 0.0f, hpz = 0.0f, hnx = 0.0f, hny = 0.0f, hnz = 0.0f;
            if constexpr (QUAD) {
                // ---- the same stages with the walks laid out quadrant by quadrant (the host guarantees n_samples % 4 == 0,
                // no reuse, no capture).  8 x 8 map: generate the pixel's camera ray, once
                stack_t* const stk_wave = stk - lane;                              // column of this wave's lane 0
                // the camera block: ONE scalar load of 16 dwords from the kernel-argument segment, live in this prologue only
                typedef uint32_t cam16_t __attribute__((ext_vector_type(16)));
                cam16_t cw;
                asm volatile("s_load_dwordx16 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(cw) : "s"((const void*)ka), "i"((int)__builtin_offsetof(FrameArgs, cam)));
                const float cpos0 = u2f(cw[0]), cpos1 = u2f(cw[1]), cpos2 = u2f(cw[2]);
                const float cfov = u2f(cw[12]), cfocal = u2f(cw[14]);
                // (what a pixel's walks leave for the 8 x 8 map waits in the pixel's column of the SECONDARY stack: rows 3..7 after
                // the primary walks, which do not touch that stack; row 0 after the shadow walks of its quadrant, which are done with it)
                // lanes per pixel in the primary walks (the unit's samples) and in the shadow walks (x shadow samples): 4, 8 or 16
                const uint32_t lg_p = __builtin_amdgcn_readfirstlane(quad_lanes_lg(n_samples));
                const uint32_t lg_s = __builtin_amdgcn_readfirstlane(quad_lanes_lg(n_samples * shadow_samples));
                Ray r0;
                bool active = false;
                float ox = 0.0f, oy = 0.0f, oz = 0.0f, dx = 0.0f, dy = 0.0f, dz = 0.0f;
                r0.tcx = r0.tcy = r0.tcz = r0.tox = r0.toy = r0.toz = 0.0f; r0.px = r0.py = r0.pz = 1.0f; r0.t_min = r0.t_max = 0.0f;
                r0.scale = SVO_MAX - 1; r0.child_offset = 0u; r0.mirror = 0u;
                if (valid) {
                    // R0: main.cpp:133,145-149; camera_controller.hpp:34-54 with rand_vec = +-0 (see PINHOLE above)
                    const float aspect_ratio = (float)W / (float)H;
                    const float lens_x = (float)px_x / (float)H - aspect_ratio * 0.5f;
                    const float lens_y = (float)px_y / (float)H - 0.5f;
                    float sx = lens_x, sy = lens_y, sz = cfov;
                    normalize3(sx, sy, sz);
                    float vx = sx * cfocal, vy = sy * cfocal, vz = sz * cfocal;
                    const float m[9] = {u2f(cw[3]), u2f(cw[4]), u2f(cw[5]), u2f(cw[6]), u2f(cw[7]), u2f(cw[8]), u2f(cw[9]), u2f(cw[10]), u2f(cw[11])};
                    normalize3(vx, vy, vz);
                    dx = (m[0] * vx + m[1] * vy) + m[2] * vz;
                    dy = (m[3] * vx + m[4] * vy) + m[5] * vz;
                    dz = (m[6] * vx + m[7] * vy) + m[8] * vz;
                    ox = (cpos0 + 0.0f) * inv_size + 1.0f;
                    oy = (cpos1 + 0.0f) * inv_size + 1.0f;
                    oz = (cpos2 + 0.0f) * inv_size + 1.0f;
                    active = ray_setup(r0, root, ox, oy, oz, dx, dy, dz);
                    const stack_t* const my_cam_path = cam_path + (r0.mirror ^ 7u) * 16u;
                    ray_start_at_camera(r0, cam_scale < 22u ? my_cam_path[15] : 22u, ox, oy, oz);
                }
                uint32_t pk = quad_pack(r0, active, valid);
                // (the kernel is at its register budget: what the pixel needs again after the walks waits in LDS meanwhile -- the
                // direction in this lane's column of the secondary stack, which no primary walk touches)
                stk[VRC_RENDER_BLOCK] = f2u(dx); stk[VRC_RENDER_BLOCK + VRC_SYNC_ROW] = f2u(dy); stk[VRC_RENDER_BLOCK + 2 * VRC_SYNC_ROW] = f2u(dz);
                // quadrant by quadrant: the four lanes of a pixel walk its camera ray, once per sample (raycaster.hpp:131)
#pragma unroll 1
                for (uint32_t k = 0; k < (1u << lg_p); ++k) {                      // (as many blocks as lanes per pixel)
                    const uint32_t src = quad_lane(k, lane & ((64u >> lg_p) - 1u), lg_p);
                    stack_t* const col = stk_wave + src;                           // the pixel's own column, shared by its lanes
                    // (a block's ray lives in the block's iteration only, and the walks run at least once: otherwise the compiler
                    // carries `r` round the unit loop and spills it -- 8 scratch stores and loads per unit, found in the ISA)
                    Ray r;
                    uint32_t rep = n_samples >> lg_p;
                    do {
                        bool casts, act;
                        quad_gather(r, casts, act, r0, pk, src, root);
                        copy_camera_path<VRC_SYNC_ROW>(col, cam_path + (r.mirror ^ 7u) * 16u, depth, cam_scale);
                        book_not_executed(ctr, act, r);
                        book_walk_of(ctr, ray_walk<0, false, true, true>(r, act, nodes, col, depth, 0.0f, 0.0f), casts);
                    } while (--rep);
                    if (lane < (64u >> lg_p)) {                                    // (a pixel's lanes hold the same result)
                        stack_t* const sv = col + VRC_RENDER_BLOCK;
                        sv[3 * VRC_SYNC_ROW] = f2u(r.px); sv[4 * VRC_SYNC_ROW] = f2u(r.py); sv[5 * VRC_SYNC_ROW] = f2u(r.pz); sv[6 * VRC_SYNC_ROW] = f2u(r.t_min);
                        sv[7 * VRC_SYNC_ROW] = (uint32_t)r.scale | (r.child_offset << 5) | (r.normal << 8) | (r.kind << 11);
                    }
                }
                if (valid) {
                    Ray r = r0;                                                    // 8 x 8 map again: the state the pixel's walk ended in
                    {
                        const stack_t* const sv = stk + VRC_RENDER_BLOCK;
                        const uint32_t e = sv[7 * VRC_SYNC_ROW];
                        r.px = u2f(sv[3 * VRC_SYNC_ROW]); r.py = u2f(sv[4 * VRC_SYNC_ROW]); r.pz = u2f(sv[5 * VRC_SYNC_ROW]); r.t_min = u2f(sv[6 * VRC_SYNC_ROW]);
                        r.scale = (int)(e & 31u); r.child_offset = (e >> 5) & 7u; r.normal = (e >> 8) & 7u; r.kind = (e >> 11) & 3u;
                        r.scale_f = u2f(((uint32_t)r.scale + 104u) << 23);
                        r.shift = r.child_offset ^ r.mirror; r.complexity = 0u; r.parent_id = 0u;
                    }
                    Hit h;
                    ox = (cpos0 + 0.0f) * inv_size + 1.0f;
                    oy = (cpos1 + 0.0f) * inv_size + 1.0f;
                    oz = (cpos2 + 0.0f) * inv_size + 1.0f;
                    dx = u2f(stk[VRC_RENDER_BLOCK]); dy = u2f(stk[VRC_RENDER_BLOCK + VRC_SYNC_ROW]); dz = u2f(stk[VRC_RENDER_BLOCK + 2 * VRC_SYNC_ROW]);
                    ray_finish(r, depth, ox, oy, oz, dx, dy, dz, h);
                    active = false;
                    if (h.hit) {
                        hit = true;
                        albedo = albedo_of(tex, h) | (h.hit >> 16 << 24);               // raycaster.hpp:141-145
                        hpx = h.px; hpy = h.py; hpz = h.pz; hnx = h.nx; hny = h.ny; hnz = h.nz;
                        // raycaster.hpp:139,150-158 shadow ray(s): the reference re-casts the identical ray
                        ox = h.px + h.nx * inv_size * 0.001f;
                        oy = h.py + h.ny * inv_size * 0.001f;
                        oz = h.pz + h.nz * inv_size * 0.001f;
                        dx = lx - ox; dy = ly - oy; dz = lz - oz;
                        normalize3(dx, dy, dz);
                        light_intensity = smax(0.0f, dot3(dx, dy, dz, h.nx, h.ny, h.nz));       // :156 (taken back below if occluded)
                        active = ray_setup(r0, root, ox, oy, oz, dx, dy, dz);
                        ray_start_below(r0, start_scale_next_to(ox, oy, oz, h.px, h.py, h.pz, h.hit >> 16), ox, oy, oz);
                    }
                }
                pk = quad_pack(r0, active, hit);
#pragma unroll 1
                for (uint32_t k = 0; k < (1u << lg_s); ++k) {
                    const uint32_t src = quad_lane(k, lane & ((64u >> lg_s) - 1u), lg_s);
                    stack_t* const col = stk_wave + src;
                    Ray r;
                    uint32_t rep = (n_samples * shadow_samples) >> lg_s;
                    do {
                        bool casts, act;
                        quad_gather(r, casts, act, r0, pk, src, root);
                        copy_stack<VRC_SYNC_ROW>(col + VRC_RENDER_BLOCK, col, depth);
                        book_not_executed(ctr, act, r);
                        book_walk_of(ctr, ray_walk<0, false, true, true>(r, act, nodes, col + VRC_RENDER_BLOCK, depth, 0.0f, 0.0f), casts);
                    } while (--rep);
                    if (lane < (64u >> lg_s)) col[VRC_RENDER_BLOCK] = r.kind;
                }
                if (hit && stk[VRC_RENDER_BLOCK]) light_intensity = 0.0f;                // :155-157
            } else
            if (valid) {
                typedef uint32_t cam16_t __attribute__((ext_vector_type(16)));      // (the camera block: see the quadrant-walk prologue)
                cam16_t cw;
                asm volatile("s_load_dwordx16 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(cw) : "s"((const void*)ka), "i"((int)__builtin_offsetof(FrameArgs, cam)));
                // R0: main.cpp:133,145-149; camera_controller.hpp:34-54 with rand_vec = +-0 (see PINHOLE above)
                const float aspect_ratio = (float)W / (float)H;
                const float lens_x = (float)px_x / (float)H - aspect_ratio * 0.5f;
                const float lens_y = (float)px_y / (float)H - 0.5f;
                float sx = lens_x, sy = lens_y, sz = u2f(cw[12]);
                normalize3(sx, sy, sz);
                float vx = sx * u2f(cw[14]), vy = sy * u2f(cw[14]), vz = sz * u2f(cw[14]);
                const float m[9] = {u2f(cw[3]), u2f(cw[4]), u2f(cw[5]), u2f(cw[6]), u2f(cw[7]), u2f(cw[8]), u2f(cw[9]), u2f(cw[10]), u2f(cw[11])};
                normalize3(vx, vy, vz);
                float dx = (m[0] * vx + m[1] * vy) + m[2] * vz;
                float dy = (m[3] * vx + m[4] * vy) + m[5] * vz;
                float dz = (m[6] * vx + m[7] * vy) + m[8] * vz;
                float ox = (u2f(cw[0]) + 0.0f) * inv_size + 1.0f;
                float oy = (u2f(cw[1]) + 0.0f) * inv_size + 1.0f;
                float oz = (u2f(cw[2]) + 0.0f) * inv_size + 1.0f;
                Ray r0, r;
                bool active = ray_setup(r0, root, ox, oy, oz, dx, dy, dz);
                // every camera ray starts at the camera, on the path of its direction signs
                const stack_t* const my_cam_path = cam_path + (r0.mirror ^ 7u) * 16u;
                ray_start_at_camera(r0, cam_scale < 22u ? my_cam_path[15] : 22u, ox, oy, oz);
                r = r0;
                for (uint32_t k = 0; k < n_walks; ++k) {                            // raycaster.hpp:131, once per sample
                    r = r0;
                    copy_camera_path<VRC_SYNC_ROW>(stk, my_cam_path, depth, cam_scale);
                    if (a.prim) {                                                   // the capture records the ray's own complexity
                        const uint32_t not_executed = book_not_executed(ctr, active, r);
                        ray_run<VRC_SYNC_ROW, 0, true>(r, active, nodes, stk, depth, 0.0f, 0.0f);
                        book_walk_lanes(ctr, r.complexity - not_executed);
                    } else {
                        ray_run_total_below<VRC_SYNC_ROW, 0>(r, active, nodes, stk, depth, 0.0f, 0.0f, ctr);
                    }
                }
                Hit h;
                ray_finish(r, depth, ox, oy, oz, dx, dy, dz, h);
#ifndef VRC_PROBE
                if (a.prim && s_begin == 0u) {
                    uint4* o = reinterpret_cast<uint4*>(a.prim + pix);
                    o[0] = make_uint4(f2u(h.px), f2u(h.py), f2u(h.pz), f2u(h.nx));
                    o[1] = make_uint4(f2u(h.ny), f2u(h.nz), f2u(h.u), f2u(h.v));
                    o[2] = make_uint4(h.hit, h.node, f2u(h.distance), h.complexity);
                }
#endif
                if (h.hit) {
                    hit = true;
                    albedo = albedo_of(tex, h) | (h.hit >> 16 << 24);               // raycaster.hpp:141-145
                    hpx = h.px; hpy = h.py; hpz = h.pz; hnx = h.nx; hny = h.ny; hnz = h.nz;
                    // raycaster.hpp:139,150-158 shadow ray(s): the reference re-casts the identical ray
                    ox = h.px + h.nx * inv_size * 0.001f;
                    oy = h.py + h.ny * inv_size * 0.001f;
                    oz = h.pz + h.nz * inv_size * 0.001f;
                    dx = lx - ox; dy = ly - oy; dz = lz - oz;
                    normalize3(dx, dy, dz);
                    const float lit = smax(0.0f, dot3(dx, dy, dz, h.nx, h.ny, h.nz));           // :156
                    active = ray_setup(r0, root, ox, oy, oz, dx, dy, dz);
                    ray_start_below(r0, from_root ? 22u : start_scale_next_to(ox, oy, oz, h.px, h.py, h.pz, h.hit >> 16), ox, oy, oz);
                    for (uint32_t k = n_shadow_walks; k--;) {
                        r = r0;
                        if (!from_root) copy_stack<VRC_SYNC_ROW>(stk + VRC_RENDER_BLOCK, stk, depth);
                        ray_run_total_below<VRC_SYNC_ROW, 0>(r, active, nodes, stk + VRC_RENDER_BLOCK, depth, 0.0f, 0.0f, ctr);
                    }
                    if (!r.kind) light_intensity = lit;                                      // :155-157
                }
            }
            // ---- per sample: the GI pair (raycaster.hpp:161,169-207), shading, sink
            for (uint32_t sample = s_begin; sample < s_end; ++sample) {
                if (valid) {
                    uint32_t color = 0u;                                             // ColorResult: Black
                    if (hit) {
                        float gi_intensity = 0.0f;
                        if (a.p.use_gi) {
                            if constexpr (ONE_BOUNCE)
                                gi_intensity = gi_chain<ONE_BOUNCE>(nodes, stk, albedo >> 24, from_root, root, depth, n_normalizer, lx, ly, lz, gi_bounces, a.p.seed,
                                                    a.p.frame_index + sample, pix, hpx, hpy, hpz, hnx, hny, hnz, ctr);
                            else
                                gi_intensity = gi_two_levels(nodes, stk, albedo >> 24, from_root, root, depth, n_normalizer, lx, ly, lz, a.p.seed,
                                                    a.p.frame_index + sample, pix, hpx, hpy, hpz, hnx, hny, hnz, ctr);
                        }
                        color = color_mult(albedo, smin(1.0f, smax(0.0f, light_intensity + gi_intensity)));   // :163
                    }
                    sink_sample(a, pix, color, acc_r, acc_g, acc_b);
                }
            }
            {   // the unit's primary hits and pixel-samples: a lane's samples all share the hit
                const unsigned long long mv = __ballot(valid), mh = __ballot(hit);
                if (lane == 0u) {
                    __hip_atomic_fetch_add(ctr + 2, (unsigned long long)__popcll(mh) * n_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __hip_atomic_fetch_add(ctr + 3, (unsigned long long)__popcll(mv) * n_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
        } else {
            for (uint32_t s0 = s_begin; s0 < s_end; s0 += NS) {
                const uint32_t sample = s0 + sub;                                    // NS samples abreast: this lane's
                const uint32_t fr = a.p.frame_index + sample;
                uint32_t color = 0u;                                                 // ColorResult: Black
#ifdef VRC_PROBE
                uint32_t probe_c0 = 0u, probe_c1 = 0u, probe_c2 = 0u, probe_c3 = 0u;
#endif
                if (valid && (NS == 1u || sample < s_end)) {
                    // the camera block: ONE scalar load of 16 dwords from the kernel-argument segment per sample, live in the ray generation only
                    typedef uint32_t cam16_t __attribute__((ext_vector_type(16)));
                    cam16_t cw;
                    asm volatile("s_load_dwordx16 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=s"(cw) : "s"((const void*)ka), "i"((int)__builtin_offsetof(FrameArgs, cam)));
                    const float cpos0 = u2f(cw[0]), cpos1 = u2f(cw[1]), cpos2 = u2f(cw[2]);
                    const float cfov = u2f(cw[12]), caperture = u2f(cw[13]), cfocal = u2f(cw[14]);
                    // R0: main.cpp:133,145-149; camera_controller.hpp:34-54
                    // (lens_x / lens_y / the focal point depend on the pixel only, camera_controller.hpp:36-39; hoisting them out of the
                    // sample loop was measured in round 4: 1.254 vs 1.246 ms with frames in flight, 1.336 vs 1.348 alone -- noise)
                    const float aspect_ratio = (float)W / (float)H;
                    const float lens_x = (float)px_x / (float)H - aspect_ratio * 0.5f;
                    const float lens_y = (float)px_y / (float)H - 0.5f;
                    float sx = lens_x, sy = lens_y, sz = cfov;
                    normalize3(sx, sy, sz);
                    const float fx = sx * cfocal, fy = sy * cfocal, fz = sz * cfocal;
                    float vx = fx, vy = fy, vz = fz;
                    float wx = 0.0f, wy = 0.0f, wz = 0.0f;
                    const float m[9] = {u2f(cw[3]), u2f(cw[4]), u2f(cw[5]), u2f(cw[6]), u2f(cw[7]), u2f(cw[8]), u2f(cw[9]), u2f(cw[10]), u2f(cw[11])};
                    if (!pinhole) {
                        const float r0 = get_rand(counter_rand(a.p.seed, fr, pix, 0u), -0.5f, 0.5f);
                        const float r1 = get_rand(counter_rand(a.p.seed, fr, pix, 1u), -0.5f, 0.5f);
                        const float rvx = caperture * r0, rvy = caperture * r1, rvz = caperture * 0.0f;
                        vx = fx - rvx; vy = fy - rvy; vz = fz - rvz;
                        wx = (m[0] * rvx + m[1] * rvy) + m[2] * rvz;
                        wy = (m[3] * rvx + m[4] * rvy) + m[5] * rvz;
                        wz = (m[6] * rvx + m[7] * rvy) + m[8] * rvz;
                    }
                    normalize3(vx, vy, vz);
                    float dx = (m[0] * vx + m[1] * vy) + m[2] * vz;
                    float dy = (m[3] * vx + m[4] * vy) + m[5] * vz;
                    float dz = (m[6] * vx + m[7] * vy) + m[8] * vz;
                    float ox = (cpos0 + wx) * inv_size + 1.0f;
                    float oy = (cpos1 + wy) * inv_size + 1.0f;
                    float oz = (cpos2 + wz) * inv_size + 1.0f;
                    Ray r;
                    bool active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
                    if constexpr (PINHOLE) {
                        // every camera ray starts at the camera, on the path of its direction signs
                        const stack_t* const my_cam_path = cam_path + (r.mirror ^ 7u) * 16u;
                        ray_start_at_camera(r, cam_scale < 22u ? my_cam_path[15] : 22u, ox, oy, oz);
                        copy_camera_path<VRC_SYNC_ROW>(stk, my_cam_path, depth, cam_scale);
                    } else {
                        // The lens point is next to the camera position: it starts where their paths from the root part.  (A camera
                        // ON a centre plane of the cube -- the reference's pose, x = z = 1.5 -- starts its lens rays at the root: the
                        // margin test fails at every scale.  Starting them on the camera path of their side of the plane was built
                        // in round 4 -- 32 M iterations fewer per C4 frame, proven against the oracle -- and measured 1 % SLOWER:
                        // the iterations saved are lockstep descends, the side selection costs registers the 7-wave build
                        // does not have.  DESIGN.md section 9.)
                        copy_camera_path<VRC_SYNC_ROW>(stk, cam_path, depth, cam_scale);
                        ray_start_below(r, start_scale_next_to(ox, oy, oz, (cpos0 + 0.0f) * inv_size + 1.0f,
                                                               (cpos1 + 0.0f) * inv_size + 1.0f,
                                                               (cpos2 + 0.0f) * inv_size + 1.0f, cam_scale), ox, oy, oz);
                    }
                    if (a.prim) {                                                   // the capture records the ray's own complexity
                        const uint32_t not_executed = book_not_executed(ctr, active, r);
                        ray_run<VRC_SYNC_ROW, 0, true>(r, active, nodes, stk, depth, 0.0f, 0.0f);
                        book_walk_lanes(ctr, r.complexity - not_executed);
                    } else {
                        ray_run_total_below<VRC_SYNC_ROW, 0>(r, active, nodes, stk, depth, 0.0f, 0.0f, ctr);
                    }
                    VRC_PROBE_WALK(r, 0)
                    Hit h;
                    ray_finish(r, depth, ox, oy, oz, dx, dy, dz, h);                   // raycaster.hpp:131
#ifndef VRC_PROBE
                    if (a.prim && sample == 0u) {
                        uint4* o = reinterpret_cast<uint4*>(a.prim + pix);
                        o[0] = make_uint4(f2u(h.px), f2u(h.py), f2u(h.pz), f2u(h.nx));
                        o[1] = make_uint4(f2u(h.ny), f2u(h.nz), f2u(h.u), f2u(h.v));
                        o[2] = make_uint4(h.hit, h.node, f2u(h.distance), h.complexity);
                    }
#endif
                    if (h.hit) {
                        n_hits += 1u;
                        const uint32_t albedo = albedo_of(tex, h);                             // raycaster.hpp:141-145
                        // raycaster.hpp:139,150-158 shadow ray(s): the reference re-casts the identical ray
                        float light_intensity = 0.0f;
                        ox = h.px + h.nx * inv_size * 0.001f;
                        oy = h.py + h.ny * inv_size * 0.001f;
                        oz = h.pz + h.nz * inv_size * 0.001f;
                        for (uint32_t i = shadow_samples; i--;) {
                            dx = lx - ox; dy = ly - oy; dz = lz - oz;
                            normalize3(dx, dy, dz);
                            const float lit = smax(0.0f, dot3(dx, dy, dz, h.nx, h.ny, h.nz));       // :156
                            active = ray_setup(r, root, ox, oy, oz, dx, dy, dz);
                            if (!from_root) copy_stack<VRC_SYNC_ROW>(stk + VRC_RENDER_BLOCK, stk, depth);
                            ray_start_below(r, from_root ? 22u : start_scale_next_to(ox, oy, oz, h.px, h.py, h.pz, (h.hit >> 16) & 0xffu), ox, oy, oz);
                            ray_run_total_below<VRC_SYNC_ROW, 0>(r, active, nodes, stk + VRC_RENDER_BLOCK, depth, 0.0f, 0.0f, ctr);
                            VRC_PROBE_WALK(r, 1)
                            if (!r.kind) light_intensity = lit;                                  // :155-157
                        }
                        // raycaster.hpp:161,169-207 GI (+ the nested second bounce of the extension)
                        float gi_intensity = 0.0f;
                        if (a.p.use_gi) {
                            if constexpr (ONE_BOUNCE)
                                gi_intensity = gi_chain<ONE_BOUNCE>(nodes, stk, (h.hit >> 16) & 0xffu, from_root, root, depth, n_normalizer, lx, ly, lz, gi_bounces, a.p.seed, fr, pix,
                                                    h.px, h.py, h.pz, h.nx, h.ny, h.nz, ctr);
                            else
                                gi_intensity = gi_two_levels(nodes, stk, (h.hit >> 16) & 0xffu, from_root, root, depth, n_normalizer, lx, ly, lz, a.p.seed, fr, pix,
                                                    h.px, h.py, h.pz, h.nx, h.ny, h.nz, ctr);
                        }
                        color = color_mult(albedo, smin(1.0f, smax(0.0f, light_intensity + gi_intensity)));   // :163
                    }
                    n_pix += 1u;
                    sink_sample(a, pix, color, acc_r, acc_g, acc_b);
                }
#ifdef VRC_PROBE
                probe_iters += probe_wave_max(probe_c0) + probe_wave_max(probe_c1) + probe_wave_max(probe_c2) + probe_wave_max(probe_c3);
#endif
            }
        }
        if constexpr (NS == 4u) {
            // the four samples of a pixel sit 16 lanes apart: every lane of the pixel gets their sum, the first one books it
            acc_r += __shfl_xor(acc_r, 16); acc_g += __shfl_xor(acc_g, 16); acc_b += __shfl_xor(acc_b, 16);
            acc_r += __shfl_xor(acc_r, 32); acc_g += __shfl_xor(acc_g, 32); acc_b += __shfl_xor(acc_b, 32);
        }
        // The sums' destinations are read from the kernel-argument segment HERE, once per unit, by the 8 x 8 maps' kernels (long
        // units: the scalar loads' latency does not show, and the pointers' four SGPRs are free during the walks: C3 -0.7 %);
        // the samples-abreast kernels have four times the units and keep them in registers (measured: +0.8 % with the loads)
        uint32_t* const unit_accum = NS == 1u ? (uint32_t*)ka->accum : a.accum;
        uint8_t* const unit_image = NS == 1u ? (uint8_t*)ka->image : a.image;
        const bool writer = valid && sub == 0u;                                      // the lane that owns the pixel's sums
        if (a.p.use_samples && !FUSED) {                                             // raycaster.hpp:87-90
            if (writer) {
                if (chunks_per_tile == 1u) {
                    uint4* ap = reinterpret_cast<uint4*>(unit_accum) + pix;
                    uint4 v = *ap;
                    v.x += acc_r; v.y += acc_g; v.z += acc_b; v.w += spp;
                    *ap = v;
                } else {
                    // two 64-bit adds per pixel: (r, g) and (b, count) as pairs of u32 -- a sum never carries out of its
                    // low word (sums < 2^32, as the accumulator format requires), so each half adds independently
                    unsigned long long* ap = reinterpret_cast<unsigned long long*>(unit_accum + 4ull * pix);
                    atomicAdd(ap + 0, (unsigned long long)acc_r | ((unsigned long long)acc_g << 32));
                    atomicAdd(ap + 1, (unsigned long long)acc_b | ((unsigned long long)(s_end - s_begin) << 32));
                }
            }
        } else if (a.p.use_samples) {
            // Fused resolve: samples_to_image (raycaster.hpp:94-103: uint8(sum / count)), the shard pack and resetSamples
            // (:105-116) for this tile, done by whichever of the tile's work units finishes last.  The accumulators are
            // zero between frames in this mode, so a tile whose samples are all in one unit never touches them.
            const uint32_t compact_row = ty * TW + ((lane / TW) & (TW - 1u));   // row inside this shard's packed buffer
            bool resolve = true;
            uint32_t sum_r = acc_r, sum_g = acc_g, sum_b = acc_b, count = s_end - s_begin;
            if (chunks_per_tile != 1u) {
                unsigned long long* ap = reinterpret_cast<unsigned long long*>(unit_accum + 4ull * pix);
                if (writer) {
                    atomicAdd(ap + 0, (unsigned long long)acc_r | ((unsigned long long)acc_g << 32));
                    atomicAdd(ap + 1, (unsigned long long)acc_b | ((unsigned long long)(s_end - s_begin) << 32));
                }
                // Every lane's adds are performed at the memory side before this unit counts as arrived; the counter add
                // is a device-scope atomic too, and the last arriver takes the sums with device-scope exchanges -- never
                // with loads: the 128-byte line of a pixel's sums also holds neighbours' (of another tile when the width
                // is not a multiple of 8), and a line a load left in this XCD's L2 is not updated by other XCDs' adds
                // (MI355X_MICROARCH.md, inter-workgroup visibility).  The exchange also leaves the zero resetSamples wants.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint32_t arrived = 0u;
                uint32_t* const tile_done = ka->tile_done;
                if (lane == 0u) arrived = atomicAdd(tile_done + tile, 1u) + 1u;
                arrived = __builtin_amdgcn_readfirstlane(arrived);
                resolve = arrived == chunks_per_tile;
                if (resolve) {
                    if (writer) {
                        const unsigned long long rg = atomicExch(ap + 0, 0ull);
                        const unsigned long long bc = atomicExch(ap + 1, 0ull);
                        sum_r = (uint32_t)rg; sum_g = (uint32_t)(rg >> 32); sum_b = (uint32_t)bc; count = (uint32_t)(bc >> 32);
                    }
                    if (lane == 0u) atomicExch(tile_done + tile, 0u);                             // ... and for the tile
                }
            }
            if (resolve && writer) {
                uint32_t c = 0xff000000u;
                if (count) c |= (sum_r / count) | ((sum_g / count) << 8) | ((sum_b / count) << 16);
                reinterpret_cast<uint32_t*>(unit_image)[pix] = c;
                uint32_t* const dst = ka->resolve_dst;
                if (dst) dst[(uint64_t)compact_row * W + px_x] = c;
            }
        }
    }

#ifdef VRC_PROBE
    if (a.prim && lane == 0u) {
        uint64_t* o = reinterpret_cast<uint64_t*>(a.prim) + 8ull * (blockIdx.x * (VRC_RENDER_BLOCK / 64u) + (threadIdx.x >> 6));
        o[0] = probe_t0; o[1] = wall_clock64(); o[2] = probe_units | ((uint64_t)probe_last_unit << 32);
        if (probe_last_unit != 0xffffffffu) {
            reinterpret_cast<uint32_t*>(a.prim)[(1u << 18) + probe_last_unit] = (uint32_t)(o[1] - probe_last_t0);
            reinterpret_cast<uint32_t*>(a.prim)[(3u << 18) + probe_last_unit] = probe_iters;
        }
        o[3] = probe_last_t0; o[4] = probe_longest; o[5] = probe_longest_unit;
        o[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    // per-lane counters (the general kernel's hits and pixel-samples) summed over the wave, plus the wave's LDS counters
    uint64_t s_hits = n_hits, s_pix = n_pix;
    for (int off = 32; off > 0; off >>= 1) {
        s_hits += __shfl_down(s_hits, off);
        s_pix += __shfl_down(s_pix, off);
    }
    if (lane == 0u) {
        uint64_t* slot = (uint64_t*)ka->stats + 8u * (blockIdx.x % VRC_STAT_SLOTS);      // (once per kernel: not worth two SGPRs until then)
        atomicAdd(reinterpret_cast<unsigned long long*>(slot + 0), ctr[1]);
        atomicAdd(reinterpret_cast<unsigned long long*>(slot + 1), ctr[0] + ctr[VRC_CTR_NOT_EXECUTED]);
        atomicAdd(reinterpret_cast<unsigned long long*>(slot + 2), (unsigned long long)s_hits + ctr[2]);
        atomicAdd(reinterpret_cast<unsigned long long*>(slot + 3), (unsigned long long)s_pix + ctr[3]);
        atomicAdd(reinterpret_cast<unsigned long long*>(slot + 4), ctr[VRC_CTR_NOT_EXECUTED]);
    }
}

// k_render_sync: any camera, the reference's one indirect bounce (k_render_sync2: the 2-bounce extension).
// k_render_sync_pinhole: aperture +0 and one bounce -- the configuration the reference ships with and BASELINE's metric
// is quoted on (k_render_sync_pinhole2: with the extension).  The bounce count is a compile-time fact in the one-bounce
// kernels: gi_chain is straight-line code there, which is worth a third of the register spills.  The ..._resolved
// kernels are the same four with the fused resolve (vrc_render_frame_resolved) compiled in; the plain ones do not
// carry its code or its arguments' registers.
#define VRC_SYNC_KERNEL_NS(name, PINHOLE, ONE_BOUNCE, FUSED, WAVES, NS)                            \
    extern "C" __global__ void __launch_bounds__(VRC_RENDER_BLOCK, WAVES) name(const FrameArgs a)  \
    {                                                                                              \
        render_sync_body<PINHOLE, ONE_BOUNCE, FUSED, false, NS>(a);                                \
    }
#define VRC_SYNC_KERNEL(name, PINHOLE, ONE_BOUNCE, FUSED, WAVES) VRC_SYNC_KERNEL_NS(name, PINHOLE, ONE_BOUNCE, FUSED, WAVES, 1u)
VRC_SYNC_KERNEL(k_render_sync, false, true, false, VRC_SYNC_MIN_WAVES)
VRC_SYNC_KERNEL(k_render_sync2, false, false, false, VRC_SYNC_MIN_WAVES)
VRC_SYNC_KERNEL(k_render_sync_pinhole, true, true, false, VRC_SYNC_MIN_WAVES)
VRC_SYNC_KERNEL(k_render_sync_pinhole2, true, false, false, VRC_SYNC_MIN_WAVES)
VRC_SYNC_KERNEL(k_render_sync_resolved, false, true, true, VRC_SYNC_MIN_WAVES)
VRC_SYNC_KERNEL(k_render_sync2_resolved, false, false, true, VRC_SYNC_MIN_WAVES)
VRC_SYNC_KERNEL(k_render_sync_pinhole_resolved, true, true, true, VRC_SYNC_MIN_WAVES)
VRC_SYNC_KERNEL(k_render_sync_pinhole2_resolved, true, false, true, VRC_SYNC_MIN_WAVES)
// the lens (DOF) one-bounce kernel also at 7 waves per SIMD (vrc_internal.h: sync_waves_for).  Rounds 2 and 3 shipped pinhole
// builds for 7 and 8 waves as well; since the walks start below the root (fewer, less coherent iterations; two stacks) the
// 6-wave build ties or beats them on every configuration and pose (profiles/r03/sweep_waves_below.txt), without their spills.
VRC_SYNC_KERNEL(k_render_sync_w7, false, true, false, 7)
VRC_SYNC_KERNEL(k_render_sync_resolved_w7, false, true, true, 7)
// four samples abreast (NS = 4, see render_sync_body): the one-bounce kernels, for sample-mode frames with spp % 4 == 0
VRC_SYNC_KERNEL_NS(k_render_sync_s4, false, true, false, VRC_SYNC_S4_WAVES_LENS, 4u)
VRC_SYNC_KERNEL_NS(k_render_sync_resolved_s4, false, true, true, VRC_SYNC_S4_WAVES_LENS, 4u)
VRC_SYNC_KERNEL_NS(k_render_sync_pinhole_s4, true, true, false, VRC_SYNC_S4_WAVES, 4u)
VRC_SYNC_KERNEL_NS(k_render_sync_pinhole_resolved_s4, true, true, true, VRC_SYNC_S4_WAVES, 4u)
// the pinhole kernels with quadrant walks (render_sync_body's QUAD)
// (the variant builds -- the C++ walk, the probes, every ray from the root -- have no quadrant-walk kernels: quad_available())
#ifndef VRC_SYNC_Q_WAVES
#define VRC_SYNC_Q_WAVES 6
#endif
#if !(defined(VRC_WALK_CPP) || defined(VRC_PROBE) || defined(VRC_NO_START_BELOW))
#define VRC_HAVE_QUAD 1
#define VRC_SYNC_KERNEL_Q(name, ONE_BOUNCE, FUSED)                                                         \
    extern "C" __global__ void __launch_bounds__(VRC_RENDER_BLOCK, VRC_SYNC_Q_WAVES) name(const FrameArgs a)  \
    {                                                                                                      \
        render_sync_body<true, ONE_BOUNCE, FUSED, false, 1u, true>(a);                                     \
    }
VRC_SYNC_KERNEL_Q(k_render_sync_pinhole_q, true, false)
VRC_SYNC_KERNEL_Q(k_render_sync_pinhole_resolved_q, true, true)
VRC_SYNC_KERNEL_Q(k_render_sync_pinhole2_q, false, false)
VRC_SYNC_KERNEL_Q(k_render_sync_pinhole2_resolved_q, false, true)
#undef VRC_SYNC_KERNEL_Q
#endif
#undef VRC_SYNC_KERNEL
#undef VRC_SYNC_KERNEL_NS
// every kernel of the 8 x 8 map once more with every ray started at the root, as lsvo.hpp:60-72 does
// (vrc_renderer_set_walk_from_root: a measurement and A/B switch -- bench.py's extra.every_ray_from_the_root says what the start
// below the root is worth; tests/test_gpu_start_below.py renders pinhole, lens and 2-bounce frames both ways and wants them equal)
#define VRC_SYNC_KERNEL_ROOT(name, PINHOLE, ONE_BOUNCE, FUSED)                                                  \
    extern "C" __global__ void __launch_bounds__(VRC_RENDER_BLOCK, VRC_SYNC_MIN_WAVES) name(const FrameArgs a)  \
    {                                                                                                           \
        render_sync_body<PINHOLE, ONE_BOUNCE, FUSED, true>(a);                                                  \
    }
VRC_SYNC_KERNEL_ROOT(k_render_sync_pinhole_from_root, true, true, false)
VRC_SYNC_KERNEL_ROOT(k_render_sync_pinhole_resolved_from_root, true, true, true)
VRC_SYNC_KERNEL_ROOT(k_render_sync_pinhole2_from_root, true, false, false)
VRC_SYNC_KERNEL_ROOT(k_render_sync_pinhole2_resolved_from_root, true, false, true)
VRC_SYNC_KERNEL_ROOT(k_render_sync_from_root, false, true, false)
VRC_SYNC_KERNEL_ROOT(k_render_sync_resolved_from_root, false, true, true)
VRC_SYNC_KERNEL_ROOT(k_render_sync2_from_root, false, false, false)
VRC_SYNC_KERNEL_ROOT(k_render_sync2_resolved_from_root, false, false, true)
#undef VRC_SYNC_KERNEL_ROOT

// ---------------------------------------------------------------------------
// S4 sink + plumbing
// ---------------------------------------------------------------------------

// raycaster.hpp:94-103: uint8(sum / count) in double == integer division for integer sums
extern "C" __global__ void k_resolve(const uint4* __restrict__ accum, uint32_t* __restrict__ image, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 v = accum[i];
    uint32_t c = 0u;
    if (v.w) c = (v.x / v.w) | ((v.y / v.w) << 8) | ((v.z / v.w) << 16);
    image[i] = c | 0xff000000u;
}

extern "C" __global__ void k_fill_u32(uint32_t* __restrict__ p, uint32_t value, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = value;
}

// Scene upload: leaf_mask &= child_mask.  The walk looks at a leaf bit only when the child bit is set
// (lsvo.hpp:81,90), so clearing stray leaf bits changes no result; it lets vrc_walk.h test both bits
// with one AND.  compileSVO output (and both builders here) already has leaf_mask a subset of child_mask.
extern "C" __global__ void k_sanitize_nodes(uint2* __restrict__ nodes, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = nodes[i].x;
    const uint32_t y = x & (0xff00ffffu | ((x & 0xff00u) << 8));
    if (y != x) nodes[i].x = y;
}

// vrc_selftest_exact_arith: the short division / square-root sequences of vrc_device.h against the IEEE operations they
// replace, over EVERY float bit pattern of the domains they are used on (out[0]: 1 / x for exponent fields 1..252, both
// signs; out[1]: sqrt(x) for positive x with exponent fields 27..252; out[2]: normalize3's 1 / sqrt(s) composition on the
// same range; out[3]: get_rand for all 100 quantisation levels and both ranges the renderer uses).  Counts mismatches.
extern "C" __global__ void k_selftest_exact_arith(unsigned long long* __restrict__ out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // 2^24 threads x 256 bit patterns
    uint32_t bad_rcp = 0u, bad_sqrt = 0u, bad_norm = 0u, bad_rand = 0u;
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t bits = (uint32_t)(tid * 256u + k);
        const uint32_t e = (bits >> 23) & 0xffu;
        const float x = u2f(bits);
        if (e >= 1u && e <= 252u && f2u(exact_rcp(x)) != f2u(1.0f / x)) ++bad_rcp;
        if (!(bits >> 31) && e >= 27u && e <= 252u) {
            const float s = sqrtf(x);
            if (f2u(exact_sqrt(x)) != f2u(s)) ++bad_sqrt;
            // normalize3 on (x', 0, 0) with x' * x' == x is not constructible for every x: check its inner composition
            if (f2u(exact_rcp(exact_sqrt(x))) != f2u(1.0f / s)) ++bad_norm;
        }
    }
    if (tid < 100u) {
        const uint32_t kk = (uint32_t)tid + 100u * 1234567u;                 // k % 100 == tid
        const float rv = (float)(kk % 100u) / 100.0f;
        if (f2u(get_rand(kk, -1000.0f, 1000.0f)) != f2u(-1000.0f + (1000.0f - -1000.0f) * rv)) ++bad_rand;
        if (f2u(get_rand(kk, -0.5f, 0.5f)) != f2u(-0.5f + (0.5f - -0.5f) * rv)) ++bad_rand;
    }
    if (bad_rcp) atomicAdd(out + 0, (unsigned long long)bad_rcp);
    if (bad_sqrt) atomicAdd(out + 1, (unsigned long long)bad_sqrt);
    if (bad_norm) atomicAdd(out + 2, (unsigned long long)bad_norm);
    if (bad_rand) atomicAdd(out + 3, (unsigned long long)bad_rand);
}

// vrc_scene_create: structural validation of a caller-supplied array, so that a truncated or malformed one is an error
// code and not an out-of-bounds gather in the walk.  Pass L visits the nodes first reached at level L (root = 0) and
// hands level L + 1 to the child blocks of their non-leaf children.
extern "C" __global__ void k_validate_level(const uint2* __restrict__ nodes, uint64_t n, uint32_t depth, uint32_t L,
                                            uint8_t* __restrict__ level, uint32_t* __restrict__ flags)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || level[i] != (uint8_t)L) return;
    const uint2 nd = nodes[i];
    const uint32_t internal = (nd.x >> 8) & 0xffu & ~(nd.x >> 16);   // children that exist and are not leaves
    if (!internal) return;
    if (L + 1u >= depth) { atomicOr(flags, 2u); return; }            // would descend below the unit voxels
    const uint32_t hi = 31u - (uint32_t)__clz((int)internal);
    if (i + (uint64_t)nd.y + hi >= n) { atomicOr(flags, 1u); return; }
    for (uint32_t c = 0; c < 8u; ++c) {
        if (!((internal >> c) & 1u)) continue;
        const uint64_t k = i + nd.y + c;
        const uint8_t seen = level[k];
        if (seen == 0xffu) level[k] = (uint8_t)(L + 1u);            // every writer of this pass writes the same value
        else if (seen != (uint8_t)(L + 1u)) atomicOr(flags, 4u);
    }
}

// compact the row blocks owned by `shard` into dst: slot k <- block shard + k*count
extern "C" __global__ void k_pack_shard(const uint32_t* __restrict__ image, uint32_t W, uint32_t H, uint32_t row_block,
                                        uint32_t shard, uint32_t count, uint32_t slots, uint32_t* __restrict__ dst)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_slot = (uint64_t)row_block * W;
    if (i >= per_slot * slots) return;
    const uint32_t k = (uint32_t)(i / per_slot);
    const uint64_t rem = i % per_slot;
    const uint32_t y = (k * count + shard) * row_block + (uint32_t)(rem / W);
    const uint32_t x = (uint32_t)(rem % W);
    dst[i] = (y < H) ? image[(uint64_t)y * W + x] : 0u;
}

// k_resolve + k_pack_shard (+ the reset of the accumulators and of the work-queue heads) in one pass over the
// shard's own row blocks: a multi-GPU frame is then render -> this -> all-gather -> unpack.
extern "C" __global__ void k_resolve_shard(uint4* __restrict__ accum, uint32_t* __restrict__ image, uint32_t W, uint32_t H,
                                           uint32_t row_block, uint32_t shard, uint32_t count, uint32_t slots,
                                           uint32_t* __restrict__ dst, uint32_t reset, uint32_t* __restrict__ queue)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (reset && blockIdx.x == 0 && threadIdx.x < VRC_QUEUE_HEADS) queue[threadIdx.x * VRC_QUEUE_STRIDE] = 0u;
    const uint64_t per_slot = (uint64_t)row_block * W;
    if (i >= per_slot * slots) return;
    const uint32_t k = (uint32_t)(i / per_slot);
    const uint64_t rem = i % per_slot;
    const uint32_t y = (k * count + shard) * row_block + (uint32_t)(rem / W);
    const uint32_t x = (uint32_t)(rem % W);
    uint32_t c = 0u;
    if (y < H) {
        const uint64_t p = (uint64_t)y * W + x;
        const uint4 v = accum[p];
        if (v.w) c = (v.x / v.w) | ((v.y / v.w) << 8) | ((v.z / v.w) << 16);   // raycaster.hpp:94-103
        c |= 0xff000000u;
        image[p] = c;
        if (reset) accum[p] = make_uint4(0u, 0u, 0u, 0u);                      // raycaster.hpp:105-116
    }
    if (dst) dst[i] = c;
}

// inverse: gathered[shard][slot][row_block][W] -> image
extern "C" __global__ void k_unpack_shards(const uint32_t* __restrict__ gathered, uint32_t W, uint32_t H, uint32_t row_block,
                                           uint32_t count, uint32_t slots, uint32_t* __restrict__ image)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)W * H) return;
    const uint32_t y = (uint32_t)(i / W), x = (uint32_t)(i % W);
    const uint32_t b = y / row_block;
    const uint32_t shard = b % count, k = b / count;
    image[i] = gathered[(((uint64_t)shard * slots + k) * row_block + (y % row_block)) * W + x];
}

// ---------------------------------------------------------------------------
// launchers (called from vrc_api.cpp through vrc_internal.h)
// ---------------------------------------------------------------------------

namespace vrc {

hipError_t launch_cast_rays(const void* nodes, int depth, uint64_t n, const float* org, const float* dir,
                            const float* coef, const float* bias, vrc_hit* out, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const int block = 256;
    const uint64_t grid = (n + block - 1) / block;
    const size_t lds = (size_t)depth * block * sizeof(stack_t);
    if (!coef && !bias)
        hipLaunchKernelGGL(k_cast_rays_nolod, dim3((uint32_t)grid), dim3(block), lds, st, (const uint2*)nodes, depth, n, org, dir, out);
    else
        hipLaunchKernelGGL(k_cast_rays, dim3((uint32_t)grid), dim3(block), lds, st,
                           (const uint2*)nodes, depth, n, org, dir, coef, bias, out);
    return hipGetLastError();
}

hipError_t launch_cast_ray_chains(const void* nodes, int depth, uint64_t n, const float* org_a, const float* dir_a, const float* org_b,
                                  const float* dir_b, float coef_b, vrc_hit* out_a, vrc_hit* out_b, uint32_t* not_executed, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const int block = 256;
    const uint64_t grid = (n + block - 1) / block;
    const size_t lds = (size_t)depth * block * sizeof(stack_t);
    if (coef_b == 0.0f)
        hipLaunchKernelGGL(k_cast_ray_chains_nolod, dim3((uint32_t)grid), dim3(block), lds, st, (const uint2*)nodes, depth, n, org_a, dir_a,
                           org_b, dir_b, out_a, out_b, not_executed);
    else
        hipLaunchKernelGGL(k_cast_ray_chains, dim3((uint32_t)grid), dim3(block), lds, st, (const uint2*)nodes, depth, n, org_a, dir_a,
                           org_b, dir_b, coef_b, out_a, out_b, not_executed);
    return hipGetLastError();
}

hipError_t launch_grid_cast(const uint8_t* cells, int X, int Y, int Z, uint64_t n, const float* org, const float* dir,
                            vrc_hit* out, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const int block = 256;
    const uint64_t grid = (n + block - 1) / block;
    hipLaunchKernelGGL(k_grid_cast, dim3((uint32_t)grid), dim3(block), 0, st, cells, X, Y, Z, n, org, dir, out);
    return hipGetLastError();
}

uint32_t sync_max_blocks_per_cu() { return VRC_SYNC_MIN_WAVES; }
uint32_t quad_lds_bytes() { return VRC_QUAD_LDS_BYTES; }
uint32_t quad_waves() { return VRC_SYNC_Q_WAVES; }
bool quad_available()
{
#ifdef VRC_HAVE_QUAD
    return true;
#else
    return false;
#endif
}
uint32_t sync_s4_waves(bool pinhole) { return pinhole ? VRC_SYNC_S4_WAVES : VRC_SYNC_S4_WAVES_LENS; }
uint32_t sync_waves_for(bool pinhole, bool one_bounce, bool use_gi, bool whole_spp_units, uint64_t pixel_samples, uint32_t requested)
{
    // the builds that exist: 6 for every kernel; the lens one-bounce kernel also 7
    (void)use_gi; (void)pixel_samples;
    if (pinhole || !one_bounce) return VRC_SYNC_MIN_WAVES;
    if (requested) return requested >= 7u ? 7u : VRC_SYNC_MIN_WAVES;   // the caller's choice, rounded down to a build that exists
    // tools/sweep_waves.sh (profiles/r03/sweep_waves_below.txt: C2..C5 x pitch 0 / -0.5 / -1.2 x frames in flight 3 / 1)
    return whole_spp_units ? 7u : VRC_SYNC_MIN_WAVES;                  // lens: 7 by 1-1.5 % with frames in flight, 6 alone on the chip
}

bool camera_is_pinhole(const vrc_camera& cam)
{
    // Pinhole camera (aperture exactly +0, the reference's default): rand_vec = 0 * getRand() is +-0, which changes
    // neither focal_point - rand_vec (focal_point.x / .y are never -0 when focal_length >= 1e-30: lens.x / .y are +0
    // or non-zero, camera_controller.hpp:36-42) nor, after the `+ 1`, the origin -- so the two lens draws and the
    // products with them are skipped.  Any other aperture or focal_length takes the general kernel.
    uint32_t ap_bits;
    memcpy(&ap_bits, &cam.aperture, 4);
    bool pinhole = ap_bits == 0u && cam.focal_length >= 1e-30f && cam.focal_length < __builtin_inff();
#ifdef VRC_NO_PINHOLE_KERNEL   // A/B builds only
    pinhole = false;
#endif
    return pinhole;
}

hipError_t launch_render(const FrameArgs& a, uint32_t grid_blocks, hipStream_t st, const char** launched)
{
    const char* name = "";
    {
        size_t lds = 2 * (size_t)a.depth * VRC_RENDER_BLOCK * sizeof(stack_t) + 1536 + 128 + VRC_CAMERA_PATH_BYTES + 32;   // two stacks, tables, the waves' counters, camera paths, iterations not executed
        const bool pinhole = camera_is_pinhole(a.cam);
        const bool one = a.p.gi_bounces <= 1u, fused = a.fused_resolve != 0u;
        const uint32_t waves = a.waves;
        struct Build { void (*fn)(const FrameArgs); const char* name; };
#define VRC_BUILD(k) Build{k, #k}
        const bool s4 = a.lane_samples == 4u && one;
#ifdef VRC_HAVE_QUAD
        const bool quad = a.quad_walks != 0u && pinhole && !s4 && !a.walk_from_root;
        if (quad) lds += VRC_QUAD_LDS_BYTES;
#endif
        const Build b =
#ifdef VRC_HAVE_QUAD
            quad ? (one ? (fused ? VRC_BUILD(k_render_sync_pinhole_resolved_q) : VRC_BUILD(k_render_sync_pinhole_q))
                        : (fused ? VRC_BUILD(k_render_sync_pinhole2_resolved_q) : VRC_BUILD(k_render_sync_pinhole2_q))) :
#endif
            s4 ? (pinhole ? (fused ? VRC_BUILD(k_render_sync_pinhole_resolved_s4) : VRC_BUILD(k_render_sync_pinhole_s4))
                          : (fused ? VRC_BUILD(k_render_sync_resolved_s4) : VRC_BUILD(k_render_sync_s4))) :
            a.walk_from_root ? (pinhole ? (one ? (fused ? VRC_BUILD(k_render_sync_pinhole_resolved_from_root) : VRC_BUILD(k_render_sync_pinhole_from_root))
                                               : (fused ? VRC_BUILD(k_render_sync_pinhole2_resolved_from_root) : VRC_BUILD(k_render_sync_pinhole2_from_root)))
                                        : (one ? (fused ? VRC_BUILD(k_render_sync_resolved_from_root) : VRC_BUILD(k_render_sync_from_root))
                                               : (fused ? VRC_BUILD(k_render_sync2_resolved_from_root) : VRC_BUILD(k_render_sync2_from_root)))) :
            (waves == 7u && !pinhole && one) ? (fused ? VRC_BUILD(k_render_sync_resolved_w7) : VRC_BUILD(k_render_sync_w7)) :
            pinhole ? (one ? (fused ? VRC_BUILD(k_render_sync_pinhole_resolved) : VRC_BUILD(k_render_sync_pinhole))
                           : (fused ? VRC_BUILD(k_render_sync_pinhole2_resolved) : VRC_BUILD(k_render_sync_pinhole2)))
                    : (one ? (fused ? VRC_BUILD(k_render_sync_resolved) : VRC_BUILD(k_render_sync))
                           : (fused ? VRC_BUILD(k_render_sync2_resolved) : VRC_BUILD(k_render_sync2)));
#undef VRC_BUILD
        void (*kernel_fn)(const FrameArgs) = b.fn;
        name = b.name;
        hipLaunchKernelGGL(kernel_fn, dim3(grid_blocks), dim3(VRC_RENDER_BLOCK), lds, st, a);
    }
    if (launched) *launched = name;
    return hipGetLastError();
}

hipError_t launch_resolve(const void* accum, void* image, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_resolve, dim3((n + 255) / 256), dim3(256), 0, st, (const uint4*)accum, (uint32_t*)image, n);
    return hipGetLastError();
}

hipError_t launch_fill_u32(void* p, uint32_t value, uint64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_u32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, value, n);
    return hipGetLastError();
}

hipError_t launch_sanitize_nodes(void* nodes, uint64_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sanitize_nodes, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (uint2*)nodes, n);
    return hipGetLastError();
}

hipError_t launch_validate_nodes(const void* nodes, uint64_t n, uint32_t depth, uint8_t* level_dev, uint32_t* flags_dev, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(level_dev, 0xff, n, st);
    if (e == hipSuccess) e = hipMemsetAsync(level_dev, 0, 1, st);    // the root
    if (e != hipSuccess) return e;
    for (uint32_t L = 0; L < depth; ++L)
        hipLaunchKernelGGL(k_validate_level, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (const uint2*)nodes, n, depth, L,
                           level_dev, flags_dev);
    return hipGetLastError();
}

hipError_t launch_selftest_exact_arith(unsigned long long* out_dev, hipStream_t st)
{
    hipLaunchKernelGGL(k_selftest_exact_arith, dim3(1u << 16), dim3(256), 0, st, out_dev);
    return hipGetLastError();
}

hipError_t launch_pack_shard(const void* image, uint32_t W, uint32_t H, uint32_t row_block, uint32_t shard,
                             uint32_t count, uint32_t slots, void* dst, hipStream_t st)
{
    const uint64_t n = (uint64_t)row_block * W * slots;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack_shard, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st,
                       (const uint32_t*)image, W, H, row_block, shard, count, slots, (uint32_t*)dst);
    return hipGetLastError();
}

hipError_t launch_resolve_shard(void* accum, void* image, uint32_t W, uint32_t H, uint32_t row_block, uint32_t shard,
                                uint32_t count, uint32_t slots, void* dst, uint32_t reset, uint32_t* queue, hipStream_t st)
{
    const uint64_t n = (uint64_t)row_block * W * slots;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve_shard, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, (uint4*)accum, (uint32_t*)image,
                       W, H, row_block, shard, count, slots, (uint32_t*)dst, reset, queue);
    return hipGetLastError();
}

hipError_t launch_unpack_shards(const void* gathered, uint32_t W, uint32_t H, uint32_t row_block, uint32_t count,
                                uint32_t slots, void* image, hipStream_t st)
{
    const uint64_t n = (uint64_t)W * H;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpack_shards, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st,
                       (const uint32_t*)gathered, W, H, row_block, count, slots, (uint32_t*)image);
    return hipGetLastError();
}

}  // namespace vrc
