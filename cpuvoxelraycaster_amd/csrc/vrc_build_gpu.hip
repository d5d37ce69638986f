// vrc_build_gpu.hip -- LSVO construction on the GPU: the step
// immediately before the hot path.  Produces, directly in HBM, the exact array
// compileSVO emits (src/lsvo_utils.cpp:4-49 over the tree SVO::setCell builds,
// include/svo.hpp:72-114) -- bit-identical to the host builder and to the oracle.
//
// The layout is a depth-first pre-order: the k-th node visited by compileSVO_rec
// (k = 0 for the root) owns the 8 slots starting at 1 + 8k, and a node's own
// index is its parent's block + slot (lsvo_utils.cpp:8-10,25-27,37-39).  With
//   cnt(v)  = number of internal (non-leaf, non-empty) nodes in v's subtree,
//   rank(v) = pre-order index of v among internal nodes
//           = rank(parent) + 1 + sum of cnt over the siblings visited before v,
// where siblings are visited x-outer, y-middle, z-inner (:29-31) and written to
// slot z*4 + y*2 + x (:34), the whole array follows from two sweeps over dense
// per-level grids: counts bottom-up, ranks + node records top-down.  Integer /
// byte work, HBM-bound; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <new>
#include <random>
#include <vector>

#include "../../include/vrc.h"
#include "vrc_internal.h"

namespace {

#define HIP_TRYB(expr)                                                                              \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) { rc = vrc::fail(e_ == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, \
                                               "%s: %s", #expr, hipGetErrorString(e_)); goto done; } \
    } while (0)

// voxel occupancy sources ----------------------------------------------------

// main.cpp:65-74: column (x,z) solid for y in [S/2 + 1, S/2 + lim), lim = max(16, min(S, height));
// voxels outside the volume are dropped (out-of-bounds Node::sub index in the reference).
struct TerrainVox {
    const int32_t* lim;  // [x*S + z]
    uint32_t S;
    __device__ bool operator()(uint32_t x, uint32_t y, uint32_t z) const
    {
        const int32_t base = (int32_t)(S / 2u);
        const int32_t l = lim[(size_t)x * S + z];
        return (int32_t)y >= base + 1 && (int32_t)y < base + l;
    }
};

struct VolumeVox {
    const uint8_t* solid;  // [(x*S + y)*S + z]
    uint32_t S;
    __device__ bool operator()(uint32_t x, uint32_t y, uint32_t z) const
    {
        return solid[((size_t)x * S + y) * S + z] != 0;
    }
};

__global__ void k_terrain_lim(const int32_t* __restrict__ height, int32_t* __restrict__ lim, uint32_t S)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S) return;
    int32_t l = height[i];
    if (l > (int32_t)S) l = (int32_t)S;   // std::min(max_height, height)
    if (l < 16) l = 16;                   // std::max(ground_level, ...)
    lim[i] = l;
}

// level N-1 (parents of unit voxels): cnt = 1 if any of the 8 voxels is solid
template <class Vox>
__global__ void k_count_leaf_parents(Vox vox, uint32_t n /* cells per axis */, uint32_t* __restrict__ cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n * n * n) return;
    const uint32_t cz = (uint32_t)(i % n), cy = (uint32_t)((i / n) % n), cx = (uint32_t)(i / ((uint64_t)n * n));
    uint32_t any = 0;
    for (uint32_t k = 0; k < 8; ++k) any |= vox(2 * cx + (k >> 2), 2 * cy + ((k >> 1) & 1), 2 * cz + (k & 1)) ? 1u : 0u;
    cnt[i] = any;
}

// level L < N-1: cnt = 1 + sum of the children's counts if any child exists
__global__ void k_count_level(const uint32_t* __restrict__ child_cnt, uint32_t n, uint32_t* __restrict__ cnt)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n * n * n) return;
    const uint32_t cz = (uint32_t)(i % n), cy = (uint32_t)((i / n) % n), cx = (uint32_t)(i / ((uint64_t)n * n));
    const uint32_t m = 2 * n;
    uint32_t s = 0;
    for (uint32_t k = 0; k < 8; ++k)
        s += child_cnt[((uint64_t)(2 * cx + (k >> 2)) * m + (2 * cy + ((k >> 1) & 1))) * m + (2 * cz + (k & 1))];
    cnt[i] = s ? s + 1u : 0u;
}

// top-down: write this level's node records, hand rank / index to the children.
// LEAF_LEVEL: children are unit voxels (leaf_mask, lsvo_utils.cpp:40-42).
template <class Vox, bool LEAF_LEVEL>
__global__ void k_emit_level(Vox vox, uint32_t n, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ child_cnt,
                             const uint32_t* __restrict__ rank, const uint32_t* __restrict__ index,
                             uint32_t* __restrict__ child_rank, uint32_t* __restrict__ child_index,
                             uint2* __restrict__ nodes)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n * n * n) return;
    if (cnt[i] == 0u) return;                       // empty cell: its slot keeps the blank LNode()
    const uint32_t cz = (uint32_t)(i % n), cy = (uint32_t)((i / n) % n), cx = (uint32_t)(i / ((uint64_t)n * n));
    const uint32_t m = 2 * n;
    const uint32_t my_rank = rank[i], my_index = index[i];
    const uint32_t block = 1u + 8u * my_rank;       // child_pos = data.size() at visit time (:8)
    uint32_t child_mask = 0u, leaf_mask = 0u;
    uint32_t running = my_rank + 1u;
    for (uint32_t x = 0; x < 2; ++x)                // visiting order of lsvo_utils.cpp:29-31
        for (uint32_t y = 0; y < 2; ++y)
            for (uint32_t z = 0; z < 2; ++z) {
                const uint32_t sub_index = z * 4u + y * 2u + x;   // :34
                if (LEAF_LEVEL) {
                    if (vox(2 * cx + x, 2 * cy + y, 2 * cz + z)) { child_mask |= 1u << sub_index; leaf_mask |= 1u << sub_index; }
                } else {
                    const uint64_t ci = ((uint64_t)(2 * cx + x) * m + (2 * cy + y)) * m + (2 * cz + z);
                    const uint32_t c = child_cnt[ci];
                    if (c) {
                        child_mask |= 1u << sub_index;
                        child_rank[ci] = running;
                        child_index[ci] = block + sub_index;
                        running += c;
                    }
                }
            }
    // LNode{color 1, child_mask, leaf_mask, pad 0, child_offset} (lsvo_utils.hpp:5-18)
    nodes[my_index] = make_uint2(1u | (child_mask << 8) | (leaf_mask << 16), block - my_index);
}

__global__ void k_fill_blank_nodes(uint2* __restrict__ nodes, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) nodes[i] = make_uint2(1u, 0u);       // LNode(): color 1, everything else 0
}

inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + 255) / 256)); }

// `pre` (optional) enqueues the kernels that produce the occupancy source (noise, column limits); it runs inside the
// timed region.  *ms_out = device time of the build: [pre + count sweep] + [blank fill + emit sweep], two event
// pairs around the enqueued kernels; the per-level grids come from ONE allocation made before the first event,
// and the node array's allocation (its size is the count sweep's result) lies between the pairs -- neither is timed.
template <class Vox, class Pre>
int build_on_device(Vox vox, uint32_t depth, int device, int cus, vrc_scene** out, float* ms_out, Pre pre)
{
    int rc = VRC_OK;
    const uint32_t N = depth;
    std::vector<uint32_t*> cnt(N, nullptr), rank(N, nullptr), index(N, nullptr);
    uint32_t* arena = nullptr;
    void* d_nodes = nullptr;
    void* d_tex = nullptr;
    vrc_scene* s = nullptr;
    uint32_t root_cnt = 0;
    uint64_t n_nodes = 0, total_cells = 0;
    float ms_count = 0.0f, ms_emit = 0.0f;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i) HIP_TRYB(hipEventCreate(&ev[i]));
    // levels 0 .. N-2 share one allocation; the bottom level (7/8 of all cells: 1.6 GB at depth 10) gets three of its
    // own -- one multi-GB hipMalloc costs ~100 ms where the same bytes in a few pieces cost 2 ms
    for (uint32_t L = 0; L + 1 < N; ++L) total_cells += 1ull << (3 * L);
    HIP_TRYB(hipMalloc((void**)&arena, (total_cells ? total_cells : 1) * 12));
    {
        uint64_t off = 0;
        for (uint32_t L = 0; L + 1 < N; ++L) {
            const uint64_t cells = 1ull << (3 * L);
            cnt[L] = arena + off; rank[L] = arena + total_cells + off; index[L] = arena + 2 * total_cells + off;
            off += cells;
        }
        const uint64_t bottom = 1ull << (3 * (N - 1));
        HIP_TRYB(hipMalloc((void**)&cnt[N - 1], bottom * 4));
        HIP_TRYB(hipMalloc((void**)&rank[N - 1], bottom * 4));
        HIP_TRYB(hipMalloc((void**)&index[N - 1], bottom * 4));
    }
    HIP_TRYB(hipMemsetAsync(rank[0], 0, 4, nullptr));
    HIP_TRYB(hipMemsetAsync(index[0], 0, 4, nullptr));
    HIP_TRYB(hipEventRecord(ev[0], nullptr));
    HIP_TRYB(pre());
    // bottom-up counts
    {
        const uint32_t n = 1u << (N - 1);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_count_leaf_parents<Vox>), grid_for((uint64_t)n * n * n), dim3(256), 0, nullptr, vox, n, cnt[N - 1]);
        for (int L = (int)N - 2; L >= 0; --L) {
            const uint32_t nl = 1u << L;
            hipLaunchKernelGGL(k_count_level, grid_for((uint64_t)nl * nl * nl), dim3(256), 0, nullptr, cnt[L + 1], nl, cnt[L]);
        }
        HIP_TRYB(hipGetLastError());
    }
    HIP_TRYB(hipEventRecord(ev[1], nullptr));
    HIP_TRYB(hipMemcpy(&root_cnt, cnt[0], 4, hipMemcpyDeviceToHost));
    n_nodes = 1ull + 8ull * root_cnt;                 // data = { root } + 8 slots per internal node
    if (n_nodes > VRC_MAX_NODES) { rc = vrc::fail(VRC_ERR_INVALID, "scene needs more than 2^29 nodes (4 GiB)"); goto done; }
    HIP_TRYB(hipMalloc(&d_nodes, n_nodes * sizeof(vrc_lnode)));
    HIP_TRYB(hipMalloc(&d_tex, 1536));
    HIP_TRYB(hipEventRecord(ev[2], nullptr));
    hipLaunchKernelGGL(k_fill_blank_nodes, grid_for(n_nodes), dim3(256), 0, nullptr, (uint2*)d_nodes, n_nodes);
    if (root_cnt == 0) {
        // empty scene: compileSVO_rec still stores child_offset = 1 in the root (:8-10) and appends nothing
        const vrc_lnode root = {1u, 0u, 0u, 0u, 1u};
        HIP_TRYB(hipMemcpy(d_nodes, &root, sizeof(root), hipMemcpyHostToDevice));
    } else {
        for (uint32_t L = 0; L < N; ++L) {
            const uint32_t n = 1u << L;
            const uint64_t cells = (uint64_t)n * n * n;
            if (L + 1 == N)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_level<Vox, true>), grid_for(cells), dim3(256), 0, nullptr, vox, n, cnt[L],
                                   (const uint32_t*)nullptr, rank[L], index[L], (uint32_t*)nullptr, (uint32_t*)nullptr, (uint2*)d_nodes);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit_level<Vox, false>), grid_for(cells), dim3(256), 0, nullptr, vox, n, cnt[L],
                                   cnt[L + 1], rank[L], index[L], rank[L + 1], index[L + 1], (uint2*)d_nodes);
        }
        HIP_TRYB(hipGetLastError());
    }
    HIP_TRYB(hipEventRecord(ev[3], nullptr));
    HIP_TRYB(hipEventSynchronize(ev[3]));
    HIP_TRYB(hipEventElapsedTime(&ms_count, ev[0], ev[1]));
    HIP_TRYB(hipEventElapsedTime(&ms_emit, ev[2], ev[3]));
    if (ms_out) *ms_out = ms_count + ms_emit;
    HIP_TRYB(hipMemset(d_tex, 0xff, 1536));
    s = new (std::nothrow) vrc_scene();
    if (!s) { rc = vrc::fail(VRC_ERR_OOM, "out of host memory"); goto done; }
    s->device = device; s->cu_count = cus; s->d_nodes = d_nodes; s->d_tex = d_tex; s->n_nodes = n_nodes; s->depth = depth;
    d_nodes = nullptr; d_tex = nullptr;
    *out = s;
done:
    if (arena) (void)hipFree(arena);
    if (N) { (void)hipFree(cnt[N - 1]); (void)hipFree(rank[N - 1]); (void)hipFree(index[N - 1]); }
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_tex) (void)hipFree(d_tex);
    for (int i = 0; i < 4; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
    return rc;
}

}  // namespace

// ---------------------------------------------------------------------------
// N4: the terrain generator's noise (src/main.cpp:61-69 over the vendored
// FastNoise v0.4.1, SimplexFractal / FBM, FN_DECIMAL = float) on the GPU, so the
// whole chain noise -> heights -> LSVO -> frames stays in HBM.  Bit-identical to
// the real FastNoise (tests compare with heights produced by the reference's own
// FastNoise.cpp).  All float operations in the reference's order, no contraction.
//
// The noise ALGORITHM restated here (seeded permutation, 2-D simplex, FBM) is FastNoise's:
//   FastNoise -- MIT License -- Copyright (c) 2017 Jordan Peck
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation
//   the rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software,
//   and to permit persons to whom the Software is furnished to do so, subject to the following conditions: The
//   above copyright notice and this permission notice shall be included in all copies or substantial portions of
//   the Software.  THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND.
// ---------------------------------------------------------------------------
namespace {

struct NoiseTables {
    uint8_t perm[512], perm12[512];
    float F2, G2, two_G2, bounding;
};

// The permutation the reference's FastNoise builds for a seed (FastNoise::SetSeed, FastNoise.cpp:197-215): a
// Fisher-Yates pass over the identity, position `done` drawing its partner from the not yet fixed tail with
// std::mt19937_64 (one draw per position, modulo the tail length); both tables are stored twice so that
// perm[a + perm[b]] needs no wrap.  Plus the 2-D simplex skew constants (:1270-1272) and the FBM normaliser
// (CalculateFractalBounding, :217-227: gain 0.5, 3 octaves).  Host side.
void noise_tables(int seed, NoiseTables& t)
{
    std::mt19937_64 draw(seed);
    uint8_t order[256];
    for (unsigned v = 0; v < 256u; ++v) order[v] = (uint8_t)v;
    for (unsigned done = 0; done < 256u; ++done) {
        const unsigned pick = done + (unsigned)((int)(draw() % (256u - done)));
        const uint8_t fixed = order[pick];
        order[pick] = order[done];
        order[done] = fixed;
    }
    for (unsigned v = 0; v < 256u; ++v) {
        t.perm[v] = t.perm[v + 256u] = order[v];
        t.perm12[v] = t.perm12[v + 256u] = (uint8_t)(order[v] % 12u);
    }
    const volatile float root3 = 1.7320508075688772935274463415059f;
    t.F2 = 0.5f * (root3 - 1.0f);
    t.G2 = (3.0f - root3) / 6.0f;
    t.two_G2 = 2.0f * t.G2;
    float octave_gain = 0.5f, total = 1.0f;                // 1 + 0.5 + 0.25
    for (int octave = 1; octave < 3; ++octave) { total += octave_gain; octave_gain *= 0.5f; }
    t.bounding = 1.0f / total;
}

__device__ __forceinline__ int fn_fast_floor(float f) { return f >= 0 ? (int)f : (int)f - 1; }   // FastNoise.cpp:184

// One simplex corner's term: falloff (0.5 - |d|^2)^4 times the gradient picked by the doubly indexed permutation
// (Index2D_12, FastNoise.cpp:244-247; GradCoord2D, :312-317).  The 12 gradients (GRAD_X / GRAD_Y, :37-48, two
// tables there) are decoded from the index instead: entries 0-3 are the four diagonals (+-1, +-1), 4-7 are
// (+-1, 0) and 8-11 are (0, +-1), the sign alternating with the index.  A corner outside the falloff radius
// contributes exactly +0.
__device__ __forceinline__ float fn_corner2(const uint8_t* perm, const uint8_t* perm12, uint32_t offset, int cell_x, int cell_y,
                                            float dx, float dy)
{
    float fall = 0.5f - dx * dx - dy * dy;
    if (fall < 0) return 0.0f;
    const uint32_t g = perm12[(cell_x & 0xff) + perm[(cell_y & 0xff) + offset]];
    const float odd = (g & 1u) ? -1.0f : 1.0f;
    const float gx = g >= 8u ? 0.0f : odd;
    const float gy = g >= 8u ? odd : (g >= 4u ? 0.0f : ((g & 2u) ? -1.0f : 1.0f));
    fall *= fall;
    return fall * fall * (dx * gx + dy * gy);
}

// 2-D simplex noise of one octave (SingleSimplex, FastNoise.cpp:1274-1335): skew the point onto the lattice of
// unit cells, un-skew the cell origin, and sum the terms of the triangle's three corners -- the origin, the
// middle corner (one step along the larger of the two offsets), and the far corner (1, 1).
__device__ float fn_simplex2(const uint8_t* perm, const uint8_t* perm12, uint32_t offset, float x, float y, float F2, float G2, float two_G2)
{
    const float skew = (x + y) * F2;
    const int cell_x = fn_fast_floor(x + skew), cell_y = fn_fast_floor(y + skew);
    const float unskew = (float)(cell_x + cell_y) * G2;
    const float ox = x - ((float)cell_x - unskew), oy = y - ((float)cell_y - unskew);   // offset from the cell origin
    const int step_x = ox > oy ? 1 : 0, step_y = 1 - step_x;
    float sum = fn_corner2(perm, perm12, offset, cell_x, cell_y, ox, oy);
    sum = sum + fn_corner2(perm, perm12, offset, cell_x + step_x, cell_y + step_y, ox - (float)step_x + G2, oy - (float)step_y + G2);
    sum = sum + fn_corner2(perm, perm12, offset, cell_x + 1, cell_y + 1, ox - 1.0f + two_G2, oy - 1.0f + two_G2);
    return 70.0f * sum;
}

__global__ void k_terrain_heights(NoiseTables tab, uint32_t S, int32_t* __restrict__ height)
{
    __shared__ uint8_t perm[512], perm12[512];
    for (uint32_t i = threadIdx.x; i < 512u; i += blockDim.x) { perm[i] = tab.perm[i]; perm12[i] = tab.perm12[i]; }
    __syncthreads();
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * S) return;
    const uint32_t x = idx / S, z = idx % S;
    // GetNoise(0.75 x, 0.75 z): frequency 0.01, FBM over 3 octaves, lacunarity 2, gain 0.5 (:410-447, :1175-1207)
    float fx = (0.75f * (float)x) * 0.01f, fy = (0.75f * (float)z) * 0.01f;
    float sum = fn_simplex2(perm, perm12, perm[0], fx, fy, tab.F2, tab.G2, tab.two_G2);
    float amp = 1.0f;
    for (int o = 1; o < 3; ++o) {
        fx *= 2.0f; fy *= 2.0f;
        amp *= 0.5f;
        sum += fn_simplex2(perm, perm12, perm[o], fx, fy, tab.F2, tab.G2, tab.two_G2) * amp;
    }
    const float noise = sum * tab.bounding;
    height[idx] = (int32_t)(64.0f * noise + 32.0f);                              // main.cpp:69
}

}  // namespace

extern "C" int vrc_terrain_heights(int32_t seed, uint32_t size, int device, int32_t* height_host)
{
    if (!height_host || size == 0 || size > 4096) return vrc::fail(VRC_ERR_INVALID, "vrc_terrain_heights: bad argument");
    int rc = vrc::require_device(device, nullptr);
    if (rc) return rc;
    NoiseTables tab;
    noise_tables(seed, tab);
    int32_t* d_h = nullptr;
    hipError_t e = hipMalloc((void**)&d_h, (size_t)size * size * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_terrain_heights, grid_for((uint64_t)size * size), dim3(256), 0, nullptr, tab, size, d_h);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(height_host, d_h, (size_t)size * size * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_h);
    if (e != hipSuccess) return vrc::fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_terrain_heights: %s", hipGetErrorString(e));
    return VRC_OK;
}

// main.cpp:59-88 end to end on the device: noise -> column limits -> LSVO.
extern "C" int vrc_scene_build_fastnoise_terrain(int32_t seed, uint32_t depth, int device, vrc_scene** out, float* build_ms)
{
    if (!out) return vrc::fail(VRC_ERR_INVALID, "vrc_scene_build_fastnoise_terrain: null argument");
    if (depth < 2 || depth > 10) return vrc::fail(VRC_ERR_INVALID, "vrc_scene_build_fastnoise_terrain: depth %u not in [2,10]", depth);
    int cus = 0;
    int rc = vrc::require_device(device, &cus);
    if (rc) return rc;
    const uint32_t S = 1u << depth;
    NoiseTables tab;
    noise_tables(seed, tab);
    int32_t *d_h = nullptr, *d_lim = nullptr;
    hipError_t e = hipMalloc((void**)&d_h, (size_t)S * S * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_lim, (size_t)S * S * 4);
    if (e != hipSuccess) rc = vrc::fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_scene_build_fastnoise_terrain: %s", hipGetErrorString(e));
    else rc = build_on_device(TerrainVox{d_lim, S}, depth, device, cus, out, build_ms, [&]() {
        // inside the timed region: *build_ms covers noise -> heights -> column limits -> LSVO
        hipLaunchKernelGGL(k_terrain_heights, grid_for((uint64_t)S * S), dim3(256), 0, nullptr, tab, S, d_h);
        hipLaunchKernelGGL(k_terrain_lim, grid_for((uint64_t)S * S), dim3(256), 0, nullptr, d_h, d_lim, S);
        return hipGetLastError();
    });
    (void)hipFree(d_h);
    (void)hipFree(d_lim);
    return rc;
}

extern "C" int vrc_scene_build_terrain(const int32_t* height, uint32_t depth, int device, vrc_scene** out, float* build_ms)
{
    if (!height || !out) return vrc::fail(VRC_ERR_INVALID, "vrc_scene_build_terrain: null argument");
    if (depth < 2 || depth > 10) return vrc::fail(VRC_ERR_INVALID, "vrc_scene_build_terrain: depth %u not in [2,10]", depth);
    int cus = 0;
    int rc = vrc::require_device(device, &cus);
    if (rc) return rc;
    const uint32_t S = 1u << depth;
    int32_t *d_h = nullptr, *d_lim = nullptr;
    hipError_t e = hipMalloc((void**)&d_h, (size_t)S * S * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_lim, (size_t)S * S * 4);
    if (e == hipSuccess) e = hipMemcpy(d_h, height, (size_t)S * S * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = vrc::fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_scene_build_terrain: %s", hipGetErrorString(e));
    else rc = build_on_device(TerrainVox{d_lim, S}, depth, device, cus, out, build_ms, [&]() {
        hipLaunchKernelGGL(k_terrain_lim, grid_for((uint64_t)S * S), dim3(256), 0, nullptr, d_h, d_lim, S);
        return hipGetLastError();
    });
    (void)hipFree(d_h);
    (void)hipFree(d_lim);
    return rc;
}

extern "C" int vrc_scene_build_volume(const uint8_t* solid, uint32_t depth, int device, vrc_scene** out, float* build_ms)
{
    if (!solid || !out) return vrc::fail(VRC_ERR_INVALID, "vrc_scene_build_volume: null argument");
    if (depth < 2 || depth > 10) return vrc::fail(VRC_ERR_INVALID, "vrc_scene_build_volume: depth %u not in [2,10]", depth);
    int cus = 0;
    int rc = vrc::require_device(device, &cus);
    if (rc) return rc;
    const uint64_t S = 1ull << depth;
    uint8_t* d_solid = nullptr;
    hipError_t e = hipMalloc((void**)&d_solid, S * S * S);
    if (e == hipSuccess) e = hipMemcpy(d_solid, solid, S * S * S, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = vrc::fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_scene_build_volume: %s", hipGetErrorString(e));
    else rc = build_on_device(VolumeVox{d_solid, (uint32_t)S}, depth, device, cus, out, build_ms, []() { return hipSuccess; });
    (void)hipFree(d_solid);
    return rc;
}

extern "C" int vrc_scene_download_nodes(const vrc_scene* s, vrc_lnode* dst)
{
    if (!s || !dst) return vrc::fail(VRC_ERR_INVALID, "vrc_scene_download_nodes: null argument");
    hipError_t e = hipSetDevice(s->device);
    if (e == hipSuccess) e = hipMemcpy(dst, s->d_nodes, s->n_nodes * sizeof(vrc_lnode), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return vrc::fail(VRC_ERR_HIP, "vrc_scene_download_nodes: %s", hipGetErrorString(e));
    return VRC_OK;
}
