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

template <class Vox>
int build_on_device(Vox vox, uint32_t depth, int device, int cus, vrc_scene** out, float* ms_out)
{
    int rc = VRC_OK;
    const uint32_t N = depth;
    std::vector<uint32_t*> cnt(N, nullptr), rank(N, nullptr), index(N, nullptr);
    void* d_nodes = nullptr;
    void* d_tex = nullptr;
    vrc_scene* s = nullptr;
    uint32_t root_cnt = 0;
    uint64_t n_nodes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRYB(hipEventCreate(&e0));
    HIP_TRYB(hipEventCreate(&e1));
    HIP_TRYB(hipEventRecord(e0, nullptr));
    for (uint32_t L = 0; L < N; ++L) {
        const uint64_t cells = 1ull << (3 * L);
        HIP_TRYB(hipMalloc((void**)&cnt[L], cells * 4));
        HIP_TRYB(hipMalloc((void**)&rank[L], cells * 4));
        HIP_TRYB(hipMalloc((void**)&index[L], cells * 4));
    }
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
    HIP_TRYB(hipMemcpy(&root_cnt, cnt[0], 4, hipMemcpyDeviceToHost));
    n_nodes = 1ull + 8ull * root_cnt;                 // data = { root } + 8 slots per internal node
    if (n_nodes > VRC_MAX_NODES) { rc = vrc::fail(VRC_ERR_INVALID, "scene needs more than 2^29 nodes (4 GiB)"); goto done; }
    HIP_TRYB(hipMalloc(&d_nodes, n_nodes * sizeof(vrc_lnode)));
    hipLaunchKernelGGL(k_fill_blank_nodes, grid_for(n_nodes), dim3(256), 0, nullptr, (uint2*)d_nodes, n_nodes);
    if (root_cnt == 0) {
        // empty scene: compileSVO_rec still stores child_offset = 1 in the root (:8-10) and appends nothing
        const vrc_lnode root = {1u, 0u, 0u, 0u, 1u};
        HIP_TRYB(hipMemcpy(d_nodes, &root, sizeof(root), hipMemcpyHostToDevice));
    } else {
        HIP_TRYB(hipMemset(rank[0], 0, 4));
        HIP_TRYB(hipMemset(index[0], 0, 4));
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
    HIP_TRYB(hipEventRecord(e1, nullptr));
    HIP_TRYB(hipEventSynchronize(e1));
    if (ms_out) HIP_TRYB(hipEventElapsedTime(ms_out, e0, e1));
    HIP_TRYB(hipMalloc(&d_tex, 1536));
    HIP_TRYB(hipMemset(d_tex, 0xff, 1536));
    s = (vrc_scene*)calloc(1, sizeof(vrc_scene));
    if (!s) { rc = vrc::fail(VRC_ERR_OOM, "out of host memory"); goto done; }
    s->device = device; s->cu_count = cus; s->d_nodes = d_nodes; s->d_tex = d_tex; s->n_nodes = n_nodes; s->depth = depth;
    d_nodes = nullptr; d_tex = nullptr;
    *out = s;
done:
    for (uint32_t L = 0; L < N; ++L) { (void)hipFree(cnt[L]); (void)hipFree(rank[L]); (void)hipFree(index[L]); }
    if (d_nodes) (void)hipFree(d_nodes);
    if (d_tex) (void)hipFree(d_tex);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

}  // namespace

// ---------------------------------------------------------------------------
// N4: the terrain generator's noise (src/main.cpp:61-69 over the vendored
// FastNoise v0.4.1, SimplexFractal / FBM, FN_DECIMAL = float) on the GPU, so the
// whole chain noise -> heights -> LSVO -> frames stays in HBM.  Bit-identical to
// the real FastNoise (tests compare with heights produced by the reference's own
// FastNoise.cpp).  All float operations in the reference's order, no contraction.
// ---------------------------------------------------------------------------
namespace {

struct NoiseTables {
    uint8_t perm[512], perm12[512];
    float F2, G2, two_G2, bounding;
};

// FastNoise::SetSeed (FastNoise.cpp:197-215) + the constants of :1270-1272 and
// CalculateFractalBounding (:217-227, gain 0.5, 3 octaves).  Host side.
void noise_tables(int seed, NoiseTables& t)
{
    std::mt19937_64 gen(seed);
    for (int i = 0; i < 256; i++) t.perm[i] = (uint8_t)i;
    for (int j = 0; j < 256; j++) {
        const int rng = (int)(gen() % (256 - j));
        const int k = rng + j;
        const int l = t.perm[j];
        t.perm[j] = t.perm[j + 256] = t.perm[k];
        t.perm[k] = (uint8_t)l;
        t.perm12[j] = t.perm12[j + 256] = (uint8_t)(t.perm[j] % 12);
    }
    const volatile float SQRT3 = 1.7320508075688772935274463415059f;
    t.F2 = 0.5f * (SQRT3 - 1.0f);
    t.G2 = (3.0f - SQRT3) / 6.0f;
    t.two_G2 = 2.0f * t.G2;
    float amp = 0.5f, amp_fractal = 1.0f;
    for (int i = 1; i < 3; i++) { amp_fractal += amp; amp *= 0.5f; }
    t.bounding = 1.0f / amp_fractal;
}

__device__ __forceinline__ int fn_fast_floor(float f) { return f >= 0 ? (int)f : (int)f - 1; }   // FastNoise.cpp:184

__device__ __forceinline__ float fn_grad2(const uint8_t* perm, const uint8_t* perm12, uint32_t offset, int x, int y, float xd, float yd)
{
    // Index2D_12 (:244-247) + GradCoord2D (:312-317); GRAD_X / GRAD_Y (:37-48) as selects
    const uint32_t lut = perm12[(x & 0xff) + perm[(y & 0xff) + offset]];
    const float gx = lut >= 8u ? 0.0f : ((lut & 1u) ? -1.0f : 1.0f);
    const float gy = lut < 4u ? ((lut & 2u) ? -1.0f : 1.0f) : (lut < 8u ? 0.0f : ((lut & 1u) ? -1.0f : 1.0f));
    return xd * gx + yd * gy;
}

__device__ float fn_simplex2(const uint8_t* perm, const uint8_t* perm12, uint32_t offset, float x, float y, float F2, float G2, float two_G2)
{
    // SingleSimplex (:1274-1335)
    float t = (x + y) * F2;
    const int i = fn_fast_floor(x + t), j = fn_fast_floor(y + t);
    t = (float)(i + j) * G2;
    const float X0 = (float)i - t, Y0 = (float)j - t;
    const float x0 = x - X0, y0 = y - Y0;
    const int i1 = x0 > y0 ? 1 : 0, j1 = x0 > y0 ? 0 : 1;
    const float x1 = x0 - (float)i1 + G2, y1 = y0 - (float)j1 + G2;
    const float x2 = x0 - 1.0f + two_G2, y2 = y0 - 1.0f + two_G2;
    float n0 = 0.0f, n1 = 0.0f, n2 = 0.0f;
    t = 0.5f - x0 * x0 - y0 * y0;
    if (!(t < 0)) { t *= t; n0 = t * t * fn_grad2(perm, perm12, offset, i, j, x0, y0); }
    t = 0.5f - x1 * x1 - y1 * y1;
    if (!(t < 0)) { t *= t; n1 = t * t * fn_grad2(perm, perm12, offset, i + i1, j + j1, x1, y1); }
    t = 0.5f - x2 * x2 - y2 * y2;
    if (!(t < 0)) { t *= t; n2 = t * t * fn_grad2(perm, perm12, offset, i + 1, j + 1, x2, y2); }
    return 70.0f * (n0 + n1 + n2);
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
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_terrain_heights, grid_for((uint64_t)S * S), dim3(256), 0, nullptr, tab, S, d_h);
        hipLaunchKernelGGL(k_terrain_lim, grid_for((uint64_t)S * S), dim3(256), 0, nullptr, d_h, d_lim, S);
        e = hipGetLastError();
    }
    if (e != hipSuccess) rc = vrc::fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_scene_build_fastnoise_terrain: %s", hipGetErrorString(e));
    else rc = build_on_device(TerrainVox{d_lim, S}, depth, device, cus, out, build_ms);
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
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_terrain_lim, grid_for((uint64_t)S * S), dim3(256), 0, nullptr, d_h, d_lim, S);
        e = hipGetLastError();
    }
    if (e != hipSuccess) rc = vrc::fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_scene_build_terrain: %s", hipGetErrorString(e));
    else rc = build_on_device(TerrainVox{d_lim, S}, depth, device, cus, out, build_ms);
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
    else rc = build_on_device(VolumeVox{d_solid, (uint32_t)S}, depth, device, cus, out, build_ms);
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
