// vrc_api.cpp -- the C ABI declared in include/vrc.h (device memory, streams,
// launches).  Compiled with hipcc into libvrc_hip.so together with
// vrc_kernels.hip and vrc_builder.cpp.  No CPU compute fallback: every entry
// point that casts rays needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/vrc.h"
#include <vector>
#include "vrc_internal.h"

#include <chrono>
#include <thread>

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
thread_local char g_err[512] = "";
}

int vrc::fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
using vrc::fail;

namespace {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail(VRC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Scheduling knobs of the frame kernels.  Every renderer carries its own copy (two renderers may use different
// kernels, from different threads); the process-wide defaults below only seed new renderers and are read and written
// under g_tuning_mu.
struct Tuning {
    uint32_t blocks_per_cu = 0;         // 0 = the library's choice of build (waves per SIMD), see sync_waves_for
    bool blocks_per_cu_set = false;
    uint32_t sample_chunk = 0;   // 0 = automatic
    uint32_t tail_units_per_wave = 4;   // automatic chunking: units of half the chunk for the last tiles (0 = off); tools/chunk_time.py
    bool reuse_invariant = false;       // pinhole camera: walk a unit's sample-invariant rays once (never a process default)
    bool walk_from_root = false;        // measurement switch: no ray starts below the root
    uint32_t lane_samples = 0;          // lane <-> (pixel, sample) map of the stage-synchronous kernel: 0 = automatic, 1, 4
    bool quad_walks = true;             // pinhole kernels: the sample-invariant walks quadrant by quadrant where a launch allows it
};
Tuning g_tuning;
std::mutex g_tuning_mu;

constexpr uint32_t VRC_MAX_SPP = 65536;          // per call; the u32 accumulators hold 255 * 16.8 M samples in total

int apply_sample_chunk(Tuning& t, uint32_t samples_per_unit)
{
    // values above 0xffff0000 set the tail policy of the automatic mode instead (experiments): low 16 bits = units per wave
    if (samples_per_unit >= 0xffff0000u) { t.tail_units_per_wave = samples_per_unit & 0xffffu; return VRC_OK; }
    if (samples_per_unit > VRC_MAX_SPP) return vrc::fail(VRC_ERR_INVALID, "sample chunk %u > %u", samples_per_unit, VRC_MAX_SPP);
    t.sample_chunk = samples_per_unit;
    return VRC_OK;
}
int apply_tuning(Tuning& t, uint32_t blocks_per_cu)
{
    if (blocks_per_cu > 8) return vrc::fail(VRC_ERR_INVALID, "vrc_set_tuning: blocks_per_cu %u > 8", blocks_per_cu);
    t.blocks_per_cu = blocks_per_cu;
    t.blocks_per_cu_set = blocks_per_cu != 0;
    return VRC_OK;
}

}  // namespace

struct vrc_grid {
    int device;
    void* d_cells;
    int32_t X, Y, Z;
};

struct vrc_renderer {
    const vrc_scene* scene = nullptr;   // must outlive every vrc_render_frame call; the other calls only need r->device
    int device = 0;
    uint32_t width = 0, height = 0;
    void* d_image = nullptr;  // RGBA8
    void* image_target = nullptr;   // where sharded frames are resolved to instead of d_image (vrc_renderer_set_image_target): a peer's framebuffer
    void* d_accum = nullptr;  // 4 x u32 per pixel
    void* d_stats = nullptr;  // VRC_STATS_BYTES of counter slots, then VRC_QUEUE_BYTES of work-queue heads
    vrc_hit* d_prim = nullptr;
    // two sets of work-queue heads: a stage-synchronous launch takes its units from one set and zeroes the other for the
    // launch after it; queue_zero[s] = set s is known to hold zeros when the next launch reaches it
    bool queue_zero[2] = {false, false};
    uint32_t* d_tile_done = nullptr;   // one arrival counter per 8 x 8 tile (fused resolve), zero between frames
    Tuning tuning;      // snapshot of the process defaults at creation; vrc_renderer_set_* change it
    const char* last_kernel = "";   // symbol of the frame kernel the last vrc_render_frame* launched
};

extern "C" const char* vrc_last_error(void) { return g_err; }

extern "C" int vrc_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(VRC_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int vrc::require_device(int device, int* cu_count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(VRC_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(VRC_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    if (cu_count) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        *cu_count = prop.multiProcessorCount;
    }
    return VRC_OK;
}

using vrc::require_device;

extern "C" int vrc_stream_create(int device, void** stream)
{
    if (!stream) return fail(VRC_ERR_INVALID, "vrc_stream_create: null argument");
    *stream = nullptr;
    if (int rc = require_device(device, nullptr)) return rc;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = st;
    return VRC_OK;
}
extern "C" int vrc_stream_destroy(int device, void* stream)
{
    if (!stream) return VRC_OK;
    if (int rc = require_device(device, nullptr)) return rc;
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return VRC_OK;
}
extern "C" int vrc_stream_synchronize(int device, void* stream)
{
    if (int rc = require_device(device, nullptr)) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_set_sample_chunk(uint32_t samples_per_unit)
{
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    return apply_sample_chunk(g_tuning, samples_per_unit);
}
extern "C" int vrc_set_tuning(uint32_t blocks_per_cu)
{
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    return apply_tuning(g_tuning, blocks_per_cu);
}
extern "C" int vrc_renderer_set_sample_chunk(vrc_renderer* r, uint32_t samples_per_unit)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    return apply_sample_chunk(r->tuning, samples_per_unit);
}
extern "C" int vrc_renderer_set_invariant_ray_reuse(vrc_renderer* r, uint32_t on)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    if (on > 1) return fail(VRC_ERR_INVALID, "vrc_renderer_set_invariant_ray_reuse: on = %u (0 or 1)", on);
    r->tuning.reuse_invariant = on != 0;
    return VRC_OK;
}
extern "C" int vrc_renderer_set_walk_from_root(vrc_renderer* r, uint32_t on)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    if (on > 1) return fail(VRC_ERR_INVALID, "vrc_renderer_set_walk_from_root: on = %u (0 or 1)", on);
    r->tuning.walk_from_root = on != 0;
    return VRC_OK;
}
extern "C" int vrc_set_lane_samples(uint32_t samples)
{
    if (samples != 0 && samples != 1 && samples != 4) return fail(VRC_ERR_INVALID, "vrc_set_lane_samples: %u (0, 1 or 4)", samples);
    std::lock_guard<std::mutex> lk(g_tuning_mu);
    g_tuning.lane_samples = samples;
    return VRC_OK;
}
extern "C" int vrc_renderer_set_lane_samples(vrc_renderer* r, uint32_t samples)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    if (samples != 0 && samples != 1 && samples != 4) return fail(VRC_ERR_INVALID, "vrc_renderer_set_lane_samples: %u (0, 1 or 4)", samples);
    r->tuning.lane_samples = samples;
    return VRC_OK;
}
extern "C" int vrc_renderer_set_quad_walks(vrc_renderer* r, uint32_t on)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    if (on > 1) return fail(VRC_ERR_INVALID, "vrc_renderer_set_quad_walks: on = %u (0 or 1)", on);
    r->tuning.quad_walks = on != 0;
    return VRC_OK;
}
extern "C" const char* vrc_renderer_last_kernel(const vrc_renderer* r) { return r ? r->last_kernel : ""; }
extern "C" int vrc_renderer_set_tuning(vrc_renderer* r, uint32_t blocks_per_cu)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    return apply_tuning(r->tuning, blocks_per_cu);
}

// ---------------------------------------------------------------------------
// scene
// ---------------------------------------------------------------------------

extern "C" int vrc_scene_create(const vrc_lnode* lnodes, uint64_t n_nodes, uint32_t depth, int device, vrc_scene** out)
{
    if (!lnodes || !out || n_nodes == 0) return fail(VRC_ERR_INVALID, "vrc_scene_create: null / empty input");
    if (depth < 2 || depth > VRC_MAX_DEPTH) return fail(VRC_ERR_INVALID, "vrc_scene_create: depth %u not in [2,%d]", depth, VRC_MAX_DEPTH);
    if (n_nodes > VRC_MAX_NODES) return fail(VRC_ERR_INVALID, "vrc_scene_create: more than 2^29 nodes (4 GiB) are not addressable by the walk");
    int cus = 0;
    int rc = require_device(device, &cus);
    if (rc) return rc;
    vrc_scene* s = new (std::nothrow) vrc_scene();
    if (!s) return fail(VRC_ERR_OOM, "out of host memory");
    s->device = device; s->cu_count = cus; s->n_nodes = n_nodes; s->depth = depth;
    uint8_t* d_level = nullptr;      // n_nodes bytes of level marks, then one u32 of flags (4-byte aligned)
    const uint64_t flags_off = (n_nodes + 3ull) & ~3ull;
    uint32_t flags = 0;
    hipError_t e = hipMalloc(&s->d_nodes, n_nodes * sizeof(vrc_lnode));
    if (e == hipSuccess) e = hipMalloc(&s->d_tex, 1536);
    if (e == hipSuccess) e = hipMalloc((void**)&d_level, flags_off + 4);
    if (e == hipSuccess) e = hipMemcpy(s->d_nodes, lnodes, n_nodes * sizeof(vrc_lnode), hipMemcpyHostToDevice);
    // stray leaf bits (leaf without child: never read by the walk) are cleared in the device copy, see k_sanitize_nodes
    if (e == hipSuccess) e = vrc::launch_sanitize_nodes(s->d_nodes, n_nodes, nullptr);
    // The walk indexes nodes[parent + child_offset + slot] and its LDS stack by level without bounds checks (and drops
    // lsvo.hpp:72's lower loop bound on the strength of the tree being `depth` levels deep): verify both here, once.
    if (e == hipSuccess) e = hipMemsetAsync(d_level + flags_off, 0, 4, nullptr);
    if (e == hipSuccess) e = vrc::launch_validate_nodes(s->d_nodes, n_nodes, depth, d_level, (uint32_t*)(d_level + flags_off), nullptr);
    if (e == hipSuccess) e = hipMemcpy(&flags, d_level + flags_off, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemset(s->d_tex, 0xff, 1536);  // sf::Color::White until textures are set
    if (d_level) (void)hipFree(d_level);
    if (e != hipSuccess) {
        vrc::scene_free(s);
        return fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_scene_create: %s", hipGetErrorString(e));
    }
    if (flags) {
        vrc::scene_free(s);
        return fail(VRC_ERR_INVALID, "vrc_scene_create: malformed LNode array:%s%s%s",
                    (flags & 1u) ? " a child block reaches past the end of the array;" : "",
                    (flags & 2u) ? " a non-leaf child below the unit-voxel level (tree deeper than `depth`);" : "",
                    (flags & 4u) ? " a node is reachable at two different levels;" : "");
    }
    *out = s;
    return VRC_OK;
}

void vrc::scene_free(vrc_scene* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stage_stream) { (void)hipStreamSynchronize(s->stage_stream); (void)hipStreamDestroy(s->stage_stream); }
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_tex) (void)hipFree(s->d_tex);
    delete s;
}

extern "C" int vrc_scene_set_textures(vrc_scene* s, const uint8_t top_rgb[768], const uint8_t side_rgb[768])
{
    if (!s || !top_rgb || !side_rgb) return fail(VRC_ERR_INVALID, "vrc_scene_set_textures: null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(s->d_tex, top_rgb, 768, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy((uint8_t*)s->d_tex + 768, side_rgb, 768, hipMemcpyHostToDevice));
    return VRC_OK;
}

extern "C" int vrc_scene_destroy(vrc_scene* s)
{
    vrc::scene_free(s);
    return VRC_OK;
}

extern "C" uint64_t vrc_scene_node_count(const vrc_scene* s) { return s ? s->n_nodes : 0; }
extern "C" uint32_t vrc_scene_depth(const vrc_scene* s) { return s ? s->depth : 0; }

// ---------------------------------------------------------------------------
// per-ray operator
// ---------------------------------------------------------------------------

namespace {

// Device staging for the host-memory form of the batch operators: one grow-only block per device, kept between calls
// (a hipMalloc / hipFree pair costs more than casting ten thousand rays, and hipFree synchronises the whole device),
// at most VRC_STAGE_CACHE_MAX bytes; a call that finds the block in use by another thread, or needs more than that,
// allocates privately.
constexpr size_t VRC_STAGE_CACHE_MAX = 1ull << 30;
struct StageCache {
    std::mutex mu;
    uint8_t* ptr = nullptr;
    size_t cap = 0;
};
StageCache g_stage_cache[16];

// Stage host ray buffers through device memory around `launch`.
template <class Launch>
int staged_cast(int device, uint64_t n, const float* org, const float* dir, const float* coef, const float* bias, vrc_hit* out,
                hipStream_t st, Launch launch)
{
    // one block: hits (48 B, 16-byte aligned records first) | origins | directions | coef | bias
    const size_t need = n * (sizeof(vrc_hit) + 12 + 12 + 4 + 4);
    StageCache* cache = (device >= 0 && device < 16 && need <= VRC_STAGE_CACHE_MAX) ? &g_stage_cache[device] : nullptr;
    std::unique_lock<std::mutex> lk;
    if (cache) {
        lk = std::unique_lock<std::mutex>(cache->mu, std::try_to_lock);
        if (!lk.owns_lock()) cache = nullptr;
    }
    float *d_org = nullptr, *d_dir = nullptr, *d_coef = nullptr, *d_bias = nullptr;
    vrc_hit* d_out = nullptr;
    uint8_t* arena = nullptr;
    hipError_t e = hipSuccess;
    if (cache) {
        if (cache->cap < need) {
            if (cache->ptr) (void)hipFree(cache->ptr);
            cache->ptr = nullptr; cache->cap = 0;
            e = hipMalloc((void**)&cache->ptr, need);
            if (e == hipSuccess) cache->cap = need;
        }
        arena = cache->ptr;
    } else {
        e = hipMalloc((void**)&arena, need);
    }
    if (e == hipSuccess) {
        d_out = (vrc_hit*)arena;
        d_org = (float*)(arena + n * sizeof(vrc_hit));
        d_dir = d_org + 3 * n;
        if (coef) d_coef = d_dir + 3 * n;
        if (bias) d_bias = d_dir + 4 * n;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(d_org, org, n * 12, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dir, dir, n * 12, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && coef) e = hipMemcpyAsync(d_coef, coef, n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && bias) e = hipMemcpyAsync(d_bias, bias, n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch(d_org, d_dir, d_coef, d_bias, d_out);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(vrc_hit), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (!cache && arena) (void)hipFree(arena);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "cast_rays: %s", hipGetErrorString(e));
    return VRC_OK;
}

}  // namespace

extern "C" int vrc_cast_rays(const vrc_scene* s, uint64_t n, const float* org_xyz, const float* dir_xyz, const float* coef,
                             const float* bias, vrc_hit* out, int mem, void* stream)
{
    if (!s) return fail(VRC_ERR_INVALID, "vrc_cast_rays: null scene");
    if (n == 0) return VRC_OK;
    if (!org_xyz || !dir_xyz || !out) return fail(VRC_ERR_INVALID, "vrc_cast_rays: null buffer");
    if (n > 0xffffffffull * 256ull) return fail(VRC_ERR_INVALID, "vrc_cast_rays: too many rays for one launch");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    if (mem == VRC_MEM_DEVICE) {
        HIP_TRY(vrc::launch_cast_rays(s->d_nodes, (int)s->depth, n, org_xyz, dir_xyz, coef, bias, out, st));
        return VRC_OK;
    }
    if (mem != VRC_MEM_HOST) return fail(VRC_ERR_INVALID, "vrc_cast_rays: bad mem kind %d", mem);
    return staged_cast(s->device, n, org_xyz, dir_xyz, coef, bias, out, st,
                       [&](float* o, float* d, float* c, float* b, vrc_hit* h) {
                           return vrc::launch_cast_rays(s->d_nodes, (int)s->depth, n, o, d, c, b, h, st);
                       });
}

extern "C" int vrc_cast_ray_chains(const vrc_scene* s, uint64_t n, const float* org_a_xyz, const float* dir_a_xyz, const float* org_b_xyz,
                                   const float* dir_b_xyz, float coef_b, vrc_hit* out_a, vrc_hit* out_b, uint32_t* not_executed, void* stream)
{
    if (!s) return fail(VRC_ERR_INVALID, "vrc_cast_ray_chains: null scene");
    if (n == 0) return VRC_OK;
    if (!org_a_xyz || !dir_a_xyz || !org_b_xyz || !dir_b_xyz || !out_a || !out_b) return fail(VRC_ERR_INVALID, "vrc_cast_ray_chains: null buffer");
    if (!(coef_b >= 0.0f && coef_b <= 0.5f)) return fail(VRC_ERR_INVALID, "vrc_cast_ray_chains: coef_b = %g outside [0, 0.5] (the start below the root is proven for those)", (double)coef_b);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(vrc::launch_cast_ray_chains(s->d_nodes, (int)s->depth, n, org_a_xyz, dir_a_xyz, org_b_xyz, dir_b_xyz, coef_b, out_a, out_b,
                                        not_executed, (hipStream_t)stream));
    return VRC_OK;
}

// Camera::getClosestPoint (camera_controller.hpp:56-60, once per frame, main.cpp:115): the ray goes through the scene's
// pinned slot -- the kernel reads it from host memory and writes the HitPoint back there; the call is a 32-byte store,
// one launch on the scene's own stream, a wait for that stream and a 48-byte load.
extern "C" int vrc_cast_ray(const vrc_scene* cs, const float org[3], const float dir[3], float ray_size_coef,
                            float ray_size_bias, vrc_hit* out)
{
    if (!cs || !org || !dir || !out) return fail(VRC_ERR_INVALID, "vrc_cast_ray: null argument");
    vrc_scene* s = const_cast<vrc_scene*>(cs);
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> lk(s->stage_mu);
    if (!s->h_stage) {
        void* p = nullptr;
        HIP_TRY(hipHostMalloc(&p, 128, hipHostMallocMapped));
        hipStream_t st = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) { (void)hipHostFree(p); return fail(VRC_ERR_HIP, "vrc_cast_ray: %s", hipGetErrorString(e)); }
        s->h_stage = p; s->stage_stream = st;
    }
    float* in = (float*)s->h_stage;                       // org[3] dir[3] coef bias | (64) vrc_hit
    vrc_hit* res = (vrc_hit*)((uint8_t*)s->h_stage + 64);
    in[0] = org[0]; in[1] = org[1]; in[2] = org[2]; in[3] = dir[0]; in[4] = dir[1]; in[5] = dir[2];
    in[6] = ray_size_coef; in[7] = ray_size_bias;
    void* dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dev, s->h_stage, 0));
    float* din = (float*)dev;
    HIP_TRY(vrc::launch_cast_rays(s->d_nodes, (int)s->depth, 1, din, din + 3, din + 6, din + 7, (vrc_hit*)((uint8_t*)dev + 64), s->stage_stream));
    HIP_TRY(hipStreamSynchronize(s->stage_stream));
    *out = *res;
    return VRC_OK;
}

// ---------------------------------------------------------------------------
// dense grid
// ---------------------------------------------------------------------------

extern "C" int vrc_grid_create(const uint8_t* cells, int32_t X, int32_t Y, int32_t Z, int device, vrc_grid** out)
{
    if (!cells || !out || X <= 0 || Y <= 0 || Z <= 0) return fail(VRC_ERR_INVALID, "vrc_grid_create: bad argument");
    int rc = require_device(device, nullptr);
    if (rc) return rc;
    vrc_grid* g = (vrc_grid*)calloc(1, sizeof(vrc_grid));
    if (!g) return fail(VRC_ERR_OOM, "out of host memory");
    g->device = device; g->X = X; g->Y = Y; g->Z = Z;
    const size_t bytes = (size_t)X * Y * Z;
    hipError_t e = hipMalloc(&g->d_cells, bytes);
    if (e == hipSuccess) e = hipMemcpy(g->d_cells, cells, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (g->d_cells) (void)hipFree(g->d_cells);
        free(g);
        return fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_grid_create: %s", hipGetErrorString(e));
    }
    *out = g;
    return VRC_OK;
}

extern "C" int vrc_grid_destroy(vrc_grid* g)
{
    if (!g) return VRC_OK;
    (void)hipSetDevice(g->device);
    (void)hipFree(g->d_cells);
    free(g);
    return VRC_OK;
}

extern "C" int vrc_grid_cast_rays(const vrc_grid* g, uint64_t n, const float* org_xyz, const float* dir_xyz, vrc_hit* out,
                                  int mem, void* stream)
{
    if (!g) return fail(VRC_ERR_INVALID, "vrc_grid_cast_rays: null grid");
    if (n == 0) return VRC_OK;
    if (!org_xyz || !dir_xyz || !out) return fail(VRC_ERR_INVALID, "vrc_grid_cast_rays: null buffer");
    HIP_TRY(hipSetDevice(g->device));
    hipStream_t st = (hipStream_t)stream;
    if (mem == VRC_MEM_DEVICE) {
        HIP_TRY(vrc::launch_grid_cast((const uint8_t*)g->d_cells, g->X, g->Y, g->Z, n, org_xyz, dir_xyz, out, st));
        return VRC_OK;
    }
    if (mem != VRC_MEM_HOST) return fail(VRC_ERR_INVALID, "vrc_grid_cast_rays: bad mem kind %d", mem);
    return staged_cast(g->device, n, org_xyz, dir_xyz, nullptr, nullptr, out, st,
                       [&](float* o, float* d, float*, float*, vrc_hit* h) {
                           return vrc::launch_grid_cast((const uint8_t*)g->d_cells, g->X, g->Y, g->Z, n, o, d, h, st);
                       });
}

// ---------------------------------------------------------------------------
// renderer
// ---------------------------------------------------------------------------

extern "C" int vrc_renderer_create(const vrc_scene* s, uint32_t width, uint32_t height, vrc_renderer** out)
{
    if (!s || !out || width == 0 || height == 0) return fail(VRC_ERR_INVALID, "vrc_renderer_create: bad argument");
    if ((uint64_t)width * height > 0x7fffffffull) return fail(VRC_ERR_INVALID, "vrc_renderer_create: frame too large");
    HIP_TRY(hipSetDevice(s->device));
    vrc_renderer* r = new (std::nothrow) vrc_renderer();
    if (!r) return fail(VRC_ERR_OOM, "out of host memory");
    r->scene = s; r->device = s->device; r->width = width; r->height = height;
    { std::lock_guard<std::mutex> lk(g_tuning_mu); r->tuning = g_tuning; }
    const uint64_t n = (uint64_t)width * height;
    hipError_t e = hipMalloc(&r->d_image, n * 4);
    if (e == hipSuccess) e = hipMalloc(&r->d_accum, n * 16);
    const uint64_t n_tiles = (uint64_t)((width + 3u) / 4u) * ((height + 3u) / 4u);   // of the finest lane map (4 x 4 pixels)
    if (e == hipSuccess) e = hipMalloc(&r->d_stats, vrc::VRC_STATS_BYTES + 2 * vrc::VRC_QUEUE_BYTES);
    if (e == hipSuccess) e = hipMalloc((void**)&r->d_tile_done, n_tiles * 4);
    if (e == hipSuccess) e = hipMemset(r->d_accum, 0, n * 16);
    if (e == hipSuccess) e = hipMemset(r->d_stats, 0, vrc::VRC_STATS_BYTES + 2 * vrc::VRC_QUEUE_BYTES);
    if (e == hipSuccess) e = hipMemset(r->d_tile_done, 0, n_tiles * 4);
    r->queue_zero[0] = r->queue_zero[1] = true;
    if (e == hipSuccess) e = vrc::launch_fill_u32(r->d_image, 0xff000000u, n, nullptr);  // sf::Image::create: opaque black
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        if (r->d_image) (void)hipFree(r->d_image);
        if (r->d_accum) (void)hipFree(r->d_accum);
        if (r->d_stats) (void)hipFree(r->d_stats);
        if (r->d_tile_done) (void)hipFree(r->d_tile_done);
        delete r;
        return fail(e == hipErrorOutOfMemory ? VRC_ERR_OOM : VRC_ERR_HIP, "vrc_renderer_create: %s", hipGetErrorString(e));
    }
    *out = r;
    return VRC_OK;
}

extern "C" int vrc_renderer_destroy(vrc_renderer* r)
{
    if (!r) return VRC_OK;
    (void)hipSetDevice(r->device);   // the scene may already be gone
    (void)hipFree(r->d_image);
    (void)hipFree(r->d_accum);
    (void)hipFree(r->d_stats);
    (void)hipFree(r->d_tile_done);
    delete r;
    return VRC_OK;
}

void vrc::renderer_info(const vrc_renderer* r, int* device, uint32_t* width, uint32_t* height)
{
    *device = r->device; *width = r->width; *height = r->height;
}

extern "C" int vrc_renderer_set_primary_capture(vrc_renderer* r, vrc_hit* prim_dev)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    r->d_prim = prim_dev;
    return VRC_OK;
}

namespace {
// fused: resolve + pack + reset in the frame kernel (vrc_render_frame_resolved); dst: packed shard rows or NULL
int render_impl(vrc_renderer* r, const vrc_camera* cam, const vrc_frame_params* p, bool fused, void* dst, void* stream);
}

extern "C" int vrc_render_frame(vrc_renderer* r, const vrc_camera* cam, const vrc_frame_params* p, void* stream)
{
    return render_impl(r, cam, p, false, nullptr, stream);
}

// vrc_render_frame + vrc_resolve_shard(row_block, shard_index, shard_count, dst, reset = 1) as ONE launch where the frame
// kernel can do it (stage-synchronous kernel, sample mode, no checkerboard), as those two calls otherwise.
extern "C" int vrc_render_frame_resolved(vrc_renderer* r, const vrc_camera* cam, const vrc_frame_params* p, void* dst_dev, void* stream)
{
    if (!r || !cam || !p) return fail(VRC_ERR_INVALID, "vrc_render_frame_resolved: null argument");
    if (!p->use_samples) return fail(VRC_ERR_INVALID, "vrc_render_frame_resolved: needs use_samples (there is nothing to resolve otherwise)");
    if (p->checker_parity < 0) return render_impl(r, cam, p, true, dst_dev, stream);
    int rc = render_impl(r, cam, p, false, nullptr, stream);
    if (rc) return rc;
    const bool sharded = p->row_block && p->shard_count > 1;
    return vrc_resolve_shard(r, sharded ? p->row_block : 0u, sharded ? p->shard_index : 0u, sharded ? p->shard_count : 1u, dst_dev, 1, stream);
}

namespace {
int render_impl(vrc_renderer* r, const vrc_camera* cam, const vrc_frame_params* p, bool fused, void* dst, void* stream)
{
    if (!r || !cam || !p) return fail(VRC_ERR_INVALID, "vrc_render_frame: null argument");
    if (p->gi_bounces > 2) return fail(VRC_ERR_INVALID, "vrc_render_frame: gi_bounces %u > 2 not supported", p->gi_bounces);
    if (p->checker_parity < -1 || p->checker_parity > 1) return fail(VRC_ERR_INVALID, "vrc_render_frame: checker_parity must be -1, 0 or 1");
    if (p->row_block && p->shard_count > 1) {
        if (p->row_block % 8u) return fail(VRC_ERR_INVALID, "vrc_render_frame: row_block must be a multiple of 8");
        if (p->shard_index >= p->shard_count) return fail(VRC_ERR_INVALID, "vrc_render_frame: shard_index >= shard_count");
    }
    if (p->spp > VRC_MAX_SPP) return fail(VRC_ERR_INVALID, "vrc_render_frame: spp %u > %u per call", p->spp, VRC_MAX_SPP);
    const vrc_scene* s = r->scene;
    const Tuning tuning = r->tuning;
    HIP_TRY(hipSetDevice(r->device));
    hipStream_t st = (hipStream_t)stream;
    vrc::FrameArgs a;
    memset(&a, 0, sizeof(a));
    a.nodes = (const uint2*)s->d_nodes;
    a.tex = (const uint8_t*)s->d_tex;
    // a sample-mode frame of a shard may be resolved straight into another renderer's framebuffer (direct peer writes)
    a.image = (uint8_t*)((fused && r->image_target) ? r->image_target : r->d_image);
    a.accum = (uint32_t*)r->d_accum;
    a.prim = r->d_prim;
    a.stats = (uint64_t*)r->d_stats;
    a.queue = (uint32_t*)((uint8_t*)r->d_stats + vrc::VRC_STATS_BYTES);
    a.depth = s->depth; a.width = r->width; a.height = r->height;
    a.cam = *cam;
    a.p = *p;
    if (a.p.spp == 0) a.p.spp = 1;
    // rows this shard owns, in compact row space
    uint32_t rows = r->height;
    if (p->row_block && p->shard_count > 1) {
        const uint32_t nblocks = (r->height + p->row_block - 1) / p->row_block;
        const uint32_t mine = nblocks > p->shard_index ? (nblocks - p->shard_index + p->shard_count - 1) / p->shard_count : 0;
        rows = mine * p->row_block;
    } else {
        a.p.row_block = 0; a.p.shard_index = 0; a.p.shard_count = 1;
    }
    // checkerboard frames on the stage-synchronous kernel: 16 x 8 pixel tiles, 64 selected pixels each
    a.checker_wide = p->checker_parity >= 0 ? 1u : 0u;
    // the lane <-> (pixel, sample) map (vrc_renderer_set_lane_samples): four samples abreast where the kernel has a build for it
    // and the frame's samples divide by four -- the accumulators make the order of a pixel's samples immaterial, the 0.4 / 0.6
    // blend of the non-sample mode (raycaster.hpp:79-85) does not
    const bool can_s4 = a.p.use_samples && a.p.spp % 4u == 0u && p->checker_parity < 0 &&
                        a.p.gi_bounces <= 1u && !tuning.walk_from_root && !tuning.reuse_invariant &&
                        (tuning.sample_chunk == 0u || tuning.sample_chunk % 4u == 0u);
    // The library's choice (lane_samples 0), measured on C3 / C4 and their 1/2 .. 1/8 shards (profiles/r04/ab_lane_map.txt,
    // shard_inflight_lane_map.txt): four abreast for a launch that has the chip to itself -- 3-6 % off a frame's latency: four
    // times the units, a quarter as long, no accumulator atomics -- and the pixel tiles for whole-spp units, which a caller asks
    // for when frames overlap (there the 8 x 8 map wins by 5 %: both maps issue the same number of VALU instructions,
    // profiles/r04/pmcq_ns{1,4}.txt, and four abreast has four times the queue pops and unit prologues for a wave to sit out,
    // which a chip kept full by overlapping launches cannot hide -- DESIGN.md section 9).
    const bool caller_whole_spp = tuning.sample_chunk != 0u && tuning.sample_chunk >= a.p.spp;
    a.lane_samples = (can_s4 && (tuning.lane_samples == 4u || (tuning.lane_samples == 0u && !caller_whole_spp))) ? 4u : 1u;
    const uint32_t tw = a.lane_samples == 4u ? 4u : 8u;
    const uint32_t tiles_per_row = a.checker_wide ? (r->width + 15u) / 16u : (r->width + tw - 1u) / tw;
    const uint64_t items = (uint64_t)tiles_per_row * ((rows + tw - 1u) / tw) * 64ull;
    if (items > 0xfffffff0ull) return fail(VRC_ERR_INVALID, "vrc_render_frame: frame too large");
    a.n_items = (uint32_t)items;
    if (a.n_items == 0) return VRC_OK;
    int queue_used = -1;
    {
        uint32_t* sets[2] = {a.queue, a.queue + vrc::VRC_QUEUE_BYTES / 4};
        // take a set that is known to be zero (memset one if neither is: after an error exit); the launch zeroes the other
        // set, so the next launch finds its queue ready without a memset or a kernel in between
        const int use = r->queue_zero[0] ? 0 : (r->queue_zero[1] ? 1 : 0);
        if (!r->queue_zero[use]) HIP_TRY(hipMemsetAsync(sets[use], 0, vrc::VRC_QUEUE_BYTES, st));
        a.queue = sets[use];
        a.queue_other = sets[1 - use];
        queue_used = use;
        // Until the launch is known to have been enqueued neither set counts as zero: the other set is only zeroed BY this
        // launch (its block 0), so an error exit below must not leave it marked ready -- the next frame would take heads that
        // still hold the previous frame's consumed counts, render nothing and resolve a stale image.
        r->queue_zero[0] = r->queue_zero[1] = false;
    }
    a.fused_resolve = fused ? 1u : 0u;
    a.reuse_invariant = tuning.reuse_invariant ? 1u : 0u;
    a.walk_from_root = tuning.walk_from_root ? 1u : 0u;
    a.tile_done = r->d_tile_done;
    a.resolve_dst = (uint32_t*)dst;
    // Quadrant walks (render_sync_body's QUAD; vrc_renderer_set_quad_walks, on by default): the pinhole kernels on the 8 x 8 map
    // when every work unit has a multiple of four samples (the walks of a pixel's sample-invariant rays are laid out four
    // abreast), without invariant-ray reuse (one walk per unit: nothing to lay out) and without the primary-hit capture (which
    // records per-lane complexities); the tree must have 8 levels or more (a walk's final state waits in stack rows 3..7).
    // Decided BEFORE the occupancy and the unit policy, which follow the build that is launched (its builds sit at their own
    // occupancy); the one condition that needs the policy's result -- every unit a multiple of four samples -- is checked after
    // it, and a launch that fails it is planned again for the plain build.
    const bool quad_candidate = vrc::quad_available() && tuning.quad_walks && a.lane_samples == 1u && a.p.use_samples && a.p.spp % 4u == 0u &&
                                !a.reuse_invariant && !a.prim && !a.walk_from_root && !a.checker_wide && vrc::camera_is_pinhole(a.cam) && s->depth >= 8u;
    uint64_t want = 0, cap = 0;
    auto plan = [&](bool quad) -> int {
        want = (items + VRC_RENDER_BLOCK - 1) / VRC_RENDER_BLOCK;
        a.sample_chunk = a.sample_chunk_tail = a.tail_tiles = 0;
        uint32_t bpc = 0;
        {
            const uint32_t lds = 2u * s->depth * 1024u + 1536u + 128u + 544u + 32u + (quad ? vrc::quad_lds_bytes() : 0u);  // two stacks (primary path, secondary walks), tables, counters, camera paths
            const uint32_t fit = 163840u / lds;
            // which build of the kernel (waves per SIMD = workgroups per CU): by the kind of launch, or the caller's choice
            const bool whole_spp = a.p.use_samples && a.p.spp > 1 && tuning.sample_chunk >= a.p.spp;
            uint32_t waves = vrc::sync_waves_for(vrc::camera_is_pinhole(a.cam), a.p.gi_bounces <= 1, a.p.use_gi != 0, whole_spp,
                                                 (uint64_t)a.n_items * a.p.spp, tuning.blocks_per_cu_set ? tuning.blocks_per_cu : 0u);
            if (a.lane_samples == 4u) waves = vrc::sync_s4_waves(vrc::camera_is_pinhole(a.cam));
            if (quad) waves = vrc::quad_waves();
            if (waves > fit) waves = quad ? fit : vrc::sync_max_blocks_per_cu();   // deep trees: the stacks of 8 workgroups do not fit a CU's LDS
            a.waves = waves;
            bpc = waves;
            if (tuning.blocks_per_cu_set && tuning.blocks_per_cu < bpc) bpc = tuning.blocks_per_cu;
            if (bpc > fit) bpc = fit;
        }
        cap = (uint64_t)s->cu_count * bpc;
        if (a.p.use_samples && a.p.spp > 1) {
            // Units should be short against the launch (its end waits for the last unit of every wave, and the oldest
            // wave of a SIMD runs ~3.6x faster than the youngest) yet not so small that the accumulator atomics and
            // queue traffic show: the largest chunk that still gives ~48 units per wave of a full grid, else 2 samples
            // per unit, else (small multi-GPU shards) 1.  Measured: C3 1.91 -> 1.83 ms, C5 26.3 -> 26.0 ms
            // (tools/chunk_time.py).
            const uint64_t tiles = a.n_items / 64, waves = cap * (VRC_RENDER_BLOCK / 64);
            uint32_t c = tuning.sample_chunk ? tuning.sample_chunk : a.p.spp;
            if (c > a.p.spp) c = a.p.spp;
            if (!tuning.sample_chunk && a.lane_samples == 4u) {
                // four samples abreast: a unit's samples come in fours (a tile has 16 pixels, so there are four times the units)
                while (c % 8u == 0u && tiles * (a.p.spp / c) < 48 * waves) c /= 2;
            } else if (!tuning.sample_chunk) {
                while (c > 2 && tiles * ((a.p.spp + c - 1) / c) < 48 * waves) c = (c + 1) / 2;
                if (c == 2 && tiles * ((a.p.spp + 1) / 2) < 8 * waves) c = 1;
            }
            a.sample_chunk = c < a.p.spp ? c : 0;
            // shorter units for the tiles handed out last (about four per wave): halves the spread of the waves' end times
            uint64_t units = tiles * ((a.p.spp + c - 1) / c);
            if (!tuning.sample_chunk && c >= 2 && tuning.tail_units_per_wave && (a.lane_samples == 1u || c % 8u == 0u)) {
                const uint32_t ct = c / 2, cpt_tail = (a.p.spp + ct - 1) / ct;
                uint64_t tt = (uint64_t)tuning.tail_units_per_wave * waves / cpt_tail;
                if (tt > tiles) tt = tiles;
                a.sample_chunk_tail = ct;
                a.tail_tiles = (uint32_t)tt;
                units = (tiles - tt) * ((a.p.spp + c - 1) / c) + tt * cpt_tail;
            }
#ifdef VRC_EXP_UNITS   // experiment builds only (tools/build_variant.py): "head chunk,tail chunk,tail units per wave" from the environment
            if (const char* ev = getenv("VRC_EXP_UNITS")) {
                unsigned ec = 0, ect = 0, etpw = 0;
                if (sscanf(ev, "%u,%u,%u", &ec, &ect, &etpw) == 3 && ec >= 1 && ect >= 1 && a.lane_samples == 1u) {
                    c = ec > a.p.spp ? a.p.spp : ec;
                    a.sample_chunk = c < a.p.spp ? c : 0;
                    const uint32_t cpt_tail = (a.p.spp + ect - 1) / ect;
                    uint64_t tt = (uint64_t)etpw * waves / cpt_tail;
                    if (tt > tiles) tt = tiles;
                    a.sample_chunk_tail = ect; a.tail_tiles = (uint32_t)tt;
                    units = (tiles - tt) * ((a.p.spp + c - 1) / c) + tt * cpt_tail;
                }
            }
#endif
            // the kernel numbers work units in 32 bits
            if (units > 0xfffffff0ull) return fail(VRC_ERR_INVALID, "vrc_render_frame: %llu work units (tiles x sample chunks) do not fit 32 bits; "
                                                   "use fewer samples per call or a larger sample chunk", (unsigned long long)units);
            // one wave per unit until the chip is full: a shard of few tiles still spreads over all CUs
            want = (units + VRC_RENDER_BLOCK / 64 - 1) / (VRC_RENDER_BLOCK / 64);
        }
        return VRC_OK;
    };
    if (int rc = plan(quad_candidate)) return rc;
    if (quad_candidate) {
        const uint32_t c_head = a.sample_chunk ? a.sample_chunk : a.p.spp, c_tail = a.sample_chunk_tail ? a.sample_chunk_tail : c_head;
        if (c_head % 4u == 0u && c_tail % 4u == 0u) a.quad_walks = 1u;
        else if (int rc = plan(false)) return rc;
    }
    const uint64_t cap_launch = cap;
    const uint32_t grid = (uint32_t)(want < cap_launch ? want : cap_launch);
    HIP_TRY(vrc::launch_render(a, grid, st, &r->last_kernel));
    if (queue_used >= 0) r->queue_zero[1 - queue_used] = true;           // zeroed by the launch that is now in the stream
    return VRC_OK;
}
}  // namespace

extern "C" int vrc_samples_to_image(vrc_renderer* r, void* stream)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(vrc::launch_resolve(r->d_accum, r->d_image, r->width * r->height, (hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_reset_samples(vrc_renderer* r, void* stream)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipMemsetAsync(r->d_accum, 0, (uint64_t)r->width * r->height * 16, (hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_clear_image(vrc_renderer* r, void* stream)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(vrc::launch_fill_u32(r->d_image, 0xff000000u, (uint64_t)r->width * r->height, (hipStream_t)stream));
    return VRC_OK;
}

// ---- direct peer writes ("... or direct peer writes into the root's framebuffer") ----
// The presenting rank exports its renderer's framebuffer; every other rank opens it and makes it the target of its own
// renderer, whose frame kernel then writes this rank's rows of the frame where they belong: no pack, no collective, no
// unpack.  Ordering across processes is by interprocess events (one per frame slot and direction).
static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(vrc_ipc_handle), "vrc_ipc_handle too small for hipIpcMemHandle_t");

extern "C" int vrc_ipc_export_image(vrc_renderer* r, vrc_ipc_handle* out)
{
    if (!r || !out) return fail(VRC_ERR_INVALID, "vrc_ipc_export_image: null argument");
    HIP_TRY(hipSetDevice(r->device));
    memset(out, 0, sizeof(*out));
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, r->d_image));
    memcpy(out, &h, sizeof(h));
    return VRC_OK;
}

extern "C" int vrc_ipc_open_image(int device, const vrc_ipc_handle* handle, void** image_dev)
{
    if (!handle || !image_dev) return fail(VRC_ERR_INVALID, "vrc_ipc_open_image: null argument");
    int rc = vrc::require_device(device, nullptr);
    if (rc) return rc;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    HIP_TRY(hipIpcOpenMemHandle(image_dev, h, hipIpcMemLazyEnablePeerAccess));
    return VRC_OK;
}

extern "C" int vrc_ipc_close_image(int device, void* image_dev)
{
    if (!image_dev) return VRC_OK;
    int rc = vrc::require_device(device, nullptr);
    if (rc) return rc;
    HIP_TRY(hipIpcCloseMemHandle(image_dev));
    return VRC_OK;
}

extern "C" int vrc_renderer_set_image_target(vrc_renderer* r, void* image_dev)
{
    if (!r) return fail(VRC_ERR_INVALID, "null renderer");
    r->image_target = image_dev;
    return VRC_OK;
}

// Frame flags shared by the processes of a node: 32-bit counters in a POSIX shared-memory segment that every process maps
// and registers with its HIP runtime, written and waited for IN STREAM ORDER (hipStreamWriteValue32 / hipStreamWaitValue32,
// greater-or-equal): "rank k's rows of frame n are in framebuffer s", "framebuffer s has been consumed up to frame n".  A
// wait names a VALUE, not an earlier call, so the processes need no host messages to keep their calls in order (interprocess
// HIP events would: a wait refers to the last record the waiting process has seen -- and ROCm 7.2's lose count after 32
// records per event).
struct vrc_ipc_flags {
    int device = 0;
    uint32_t count = 0;
    size_t bytes = 0;
    uint32_t* host = nullptr;     // the mapping
    uint32_t* dev = nullptr;      // the same words as the device sees them
    bool owner = false;
    bool unlinked = false;        // the creator has removed the name already (vrc_ipc_flags_unlink)
    bool drain_failed = false;    // a stream of this process did not drain within the cap after the exchange was given up
    char name[96] = {0};
};
// the segment: `count` flags, then four words of header {magic, owner's pid, count, given up}
constexpr uint32_t VRC_FLAGS_MAGIC = 0x56524346u;   // "VRCF"
enum { FLAGS_HDR_MAGIC = 0, FLAGS_HDR_OWNER = 1, FLAGS_HDR_COUNT = 2, FLAGS_HDR_GIVEN_UP = 3, FLAGS_HDR_WORDS = 4 };

static bool process_gone(int32_t pid)
{
    if (pid <= 0) return false;
    if (kill((pid_t)pid, 0) != 0 && errno == ESRCH) return true;
    // a child that exited but has not been reaped still has a pid: its state in /proc/<pid>/stat is Z
    char path[64], buf[512];
    snprintf(path, sizeof(path), "/proc/%d/stat", (int)pid);
    FILE* fp = fopen(path, "r");
    if (!fp) return false;
    const size_t n = fread(buf, 1, sizeof(buf) - 1, fp);
    fclose(fp);
    buf[n] = 0;
    const char* rp = strrchr(buf, ')');                        // "pid (comm) S ..."
    return rp && rp[1] == ' ' && (rp[2] == 'Z' || rp[2] == 'X');
}

extern "C" int vrc_ipc_flags_open(const char* name, uint32_t count, int device, int create, vrc_ipc_flags** out)
{
    if (!name || !out || !count || count > (1u << 20)) return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: bad argument");
    if (name[0] != '/' || strlen(name) >= sizeof(vrc_ipc_flags::name)) return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: name must be \"/something\" (shm_open)");
    int rc = vrc::require_device(device, nullptr);
    if (rc) return rc;
    const size_t page = 4096, bytes = (((size_t)(count + FLAGS_HDR_WORDS) * 4u) + page - 1) / page * page;
    int fd = create ? shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600) : shm_open(name, O_RDWR, 0600);
    if (fd < 0 && create && errno == EEXIST) {
        // a segment of that name exists: replace it only when the process that made it is gone (a run that died) -- unlinking
        // one that is in use would leave its processes waiting on memory nobody else maps
        int32_t owner_pid = 0;
        const int old = shm_open(name, O_RDONLY, 0600);
        if (old >= 0) {
            struct stat sb;
            if (fstat(old, &sb) == 0 && sb.st_size >= (off_t)(FLAGS_HDR_WORDS * 4)) {
                // the header sits behind the flags; its count word says where
                void* m0 = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_SHARED, old, 0);
                if (m0 != MAP_FAILED) {
                    const uint32_t* w = (const uint32_t*)m0;
                    const size_t words = (size_t)sb.st_size / 4u;
                    for (size_t c = 0; c + FLAGS_HDR_WORDS <= words; ++c)        // find {magic, pid, count == c}
                        if (w[c + FLAGS_HDR_MAGIC] == VRC_FLAGS_MAGIC && w[c + FLAGS_HDR_COUNT] == (uint32_t)c) { owner_pid = (int32_t)w[c + FLAGS_HDR_OWNER]; break; }
                    munmap(m0, (size_t)sb.st_size);
                }
            }
            close(old);
        }
        if (owner_pid > 0 && owner_pid != (int32_t)getpid() && !process_gone(owner_pid))
            return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: %s is in use by process %d (give concurrent runs different names)", name, (int)owner_pid);
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    }
    if (fd < 0) return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: shm_open(%s): %s", name, strerror(errno));
    if (create && ftruncate(fd, (off_t)bytes) != 0) {           // a new segment reads as zeros
        const int e = errno; close(fd); shm_unlink(name);
        return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: ftruncate: %s", strerror(e));
    }
    if (!create) {                                               // never map past what is there: that is a SIGBUS on first touch
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size < (off_t)bytes) {
            close(fd);
            return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: %s holds %lld bytes, %u flags need %zu (not created yet, or made for another count)",
                        name, (long long)sb.st_size, count, bytes);
        }
    }
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    const int em = errno;
    close(fd);
    if (m == MAP_FAILED) { if (create) shm_unlink(name); return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: mmap: %s", strerror(em)); }
    hipError_t e = hipHostRegister(m, bytes, hipHostRegisterMapped);
    void* d = nullptr;
    if (e == hipSuccess) e = hipHostGetDevicePointer(&d, m, 0);
    if (e != hipSuccess) {
        munmap(m, bytes);
        if (create) shm_unlink(name);
        return fail(VRC_ERR_HIP, "vrc_ipc_flags_open: registering the segment: %s", hipGetErrorString(e));
    }
    uint32_t* hdr = (uint32_t*)m + count;
    if (create) {
        hdr[FLAGS_HDR_OWNER] = (uint32_t)getpid(); hdr[FLAGS_HDR_COUNT] = count; hdr[FLAGS_HDR_GIVEN_UP] = 0u;
        __atomic_store_n(hdr + FLAGS_HDR_MAGIC, VRC_FLAGS_MAGIC, __ATOMIC_RELEASE);
    } else if (__atomic_load_n(hdr + FLAGS_HDR_MAGIC, __ATOMIC_ACQUIRE) != VRC_FLAGS_MAGIC || hdr[FLAGS_HDR_COUNT] != count) {
        (void)hipHostUnregister(m);
        munmap(m, bytes);
        return fail(VRC_ERR_INVALID, "vrc_ipc_flags_open: %s was not made for %u flags", name, count);
    }
    vrc_ipc_flags* f = new vrc_ipc_flags;
    f->device = device; f->count = count; f->bytes = bytes; f->host = (uint32_t*)m; f->dev = (uint32_t*)d; f->owner = create != 0;
    snprintf(f->name, sizeof(f->name), "%s", name);
    *out = f;
    return VRC_OK;
}

static bool flags_given_up(const vrc_ipc_flags* f) { return __atomic_load_n(f->host + f->count + FLAGS_HDR_GIVEN_UP, __ATOMIC_ACQUIRE) != 0u; }

// the watchdog of a stream that waits for flags (include/vrc.h).  timeout_ms is an INACTIVITY limit: the clock starts again
// whenever any flag of the segment changes (a frame of some rank completed), so a healthy exchange that is still draining a
// long queue is never declared dead -- only one on which nothing has moved for timeout_ms.
extern "C" int vrc_ipc_stream_wait(vrc_ipc_flags* f, void* stream, const int32_t* pids, uint32_t n_pids, uint32_t timeout_ms)
{
    if (!f || (n_pids && !pids)) return fail(VRC_ERR_INVALID, "vrc_ipc_stream_wait: bad argument");
    HIP_TRY(hipSetDevice(f->device));
    auto flags_digest = [f]() {                                    // changes whenever a flag does (flags only ever grow)
        uint64_t d = 0;
        for (uint32_t i = 0; i < f->count; ++i) d += __atomic_load_n(f->host + i, __ATOMIC_RELAXED);
        return d;
    };
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_progress = t_begin;
    uint64_t digest = flags_digest();
    const char* why = nullptr;
    int32_t who = 0;
    uint32_t polls = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery((hipStream_t)stream);
        if (q == hipSuccess) return flags_given_up(f) ? fail(VRC_ERR_PEER, "vrc_ipc_stream_wait: the exchange was given up (a peer died or timed out)") : VRC_OK;
        if (q != hipErrorNotReady) return fail(VRC_ERR_HIP, "vrc_ipc_stream_wait: hipStreamQuery: %s", hipGetErrorString(q));
        if (flags_given_up(f)) { why = "another process gave the exchange up"; break; }
        const auto now = std::chrono::steady_clock::now();
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(now - t_begin).count();
        // the caller usually sits in a timed region: the first 3 ms are polled without sleeping (a stream that is nearly
        // drained ends within a poll, not within a sleep), then 50 us naps, 500 us ones after 100 ms.  The peers and the
        // flags are looked at every 64th poll (every few milliseconds once the naps have begun), as before
        if ((++polls & 63u) == 0u) {
            const uint64_t d = flags_digest();
            if (d != digest) { digest = d; t_progress = now; }
            for (uint32_t k = 0; k < n_pids && !why; ++k)
                if (process_gone(pids[k])) { why = "a peer process is gone"; who = pids[k]; }
            if (why) break;
            const auto idle_ms = std::chrono::duration_cast<std::chrono::milliseconds>(now - t_progress).count();
            if (timeout_ms && idle_ms >= (long long)timeout_ms) { why = "no flag moved within the timeout"; break; }
        }
        if (us >= 3000) std::this_thread::sleep_for(std::chrono::microseconds(us < 100000 ? 50 : 500));
    }
    // give up for everyone: mark the segment, then release every wait on its flags (>= comparisons: the largest value passes all).
    // The release has to be HELD while the stream drains: the stream-ordered flag writes still queued behind the waits (this
    // process's and the peers') put ordinary frame numbers back into the flags, and a wait that comes after such a write would
    // block again -- so the flags are re-asserted until this stream is empty (every process's own watchdog does the same for its
    // stream; bounded, in case the device itself is gone).
    __atomic_store_n(f->host + f->count + FLAGS_HDR_GIVEN_UP, 1u, __ATOMIC_RELEASE);
    const auto t1 = std::chrono::steady_clock::now();
    // (bounded: 20 s for the first stream of this process that does not drain -- the device itself may be gone, or busy tearing
    // the dead process down --, 2 s for every one after it, so that a process with several streams still leaves within half a minute)
    const long long cap_ms = f->drain_failed ? 2000 : 20000;
    bool drained = false;
    for (;;) {
        for (uint32_t i = 0; i < f->count; ++i) __atomic_store_n(f->host + i, 0xffffffffu, __ATOMIC_RELEASE);
        if (hipStreamQuery((hipStream_t)stream) != hipErrorNotReady) { drained = true; break; }
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t1).count() >= cap_ms) break;
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    if (!drained) f->drain_failed = true;
    // once more behind the last operation of this stream: a flag write that was still queued may have landed after the last
    // store above and put an ordinary frame number back, on which another process's later wait would block again
    for (uint32_t i = 0; i < f->count; ++i) __atomic_store_n(f->host + i, 0xffffffffu, __ATOMIC_RELEASE);
    return fail(VRC_ERR_PEER, "vrc_ipc_stream_wait: %s (pid %d); every wait on %s was released, frames after this are not valid", why, (int)who, f->name);
}

// Once every process of the run has opened the segment its NAME is no longer needed: the creator removes it (the mappings stay
// valid until the last process unmaps), so a run that is killed later leaves nothing behind in /dev/shm.
extern "C" int vrc_ipc_flags_unlink(vrc_ipc_flags* f)
{
    if (!f) return fail(VRC_ERR_INVALID, "vrc_ipc_flags_unlink: null argument");
    if (!f->owner) return fail(VRC_ERR_INVALID, "vrc_ipc_flags_unlink: only the process that created %s removes its name", f->name);
    if (!f->unlinked && shm_unlink(f->name) != 0 && errno != ENOENT)
        return fail(VRC_ERR_INVALID, "vrc_ipc_flags_unlink: shm_unlink(%s): %s", f->name, strerror(errno));
    f->unlinked = true;
    return VRC_OK;
}

extern "C" int vrc_ipc_flags_close(vrc_ipc_flags* f)
{
    if (!f) return VRC_OK;
    (void)hipSetDevice(f->device);
    (void)hipHostUnregister(f->host);
    munmap(f->host, f->bytes);
    if (f->owner && !f->unlinked) shm_unlink(f->name);
    delete f;
    return VRC_OK;
}

extern "C" int vrc_stream_write_flag(vrc_ipc_flags* f, uint32_t index, uint32_t value, void* stream)
{
    if (!f || index >= f->count) return fail(VRC_ERR_INVALID, "vrc_stream_write_flag: bad argument");
    if (flags_given_up(f)) return fail(VRC_ERR_PEER, "vrc_stream_write_flag: the exchange on %s was given up", f->name);
    HIP_TRY(hipSetDevice(f->device));
    HIP_TRY(hipStreamWriteValue32((hipStream_t)stream, f->dev + index, value, 0));
    return VRC_OK;
}

extern "C" int vrc_stream_wait_flag(vrc_ipc_flags* f, uint32_t index, uint32_t value, void* stream)
{
    if (!f || index >= f->count) return fail(VRC_ERR_INVALID, "vrc_stream_wait_flag: bad argument");
    if (flags_given_up(f)) return fail(VRC_ERR_PEER, "vrc_stream_wait_flag: the exchange on %s was given up", f->name);
    HIP_TRY(hipSetDevice(f->device));
    HIP_TRY(hipStreamWaitValue32((hipStream_t)stream, f->dev + index, value, hipStreamWaitValueGte, 0xffffffffu));
    return VRC_OK;
}

extern "C" int vrc_ipc_flag_set(vrc_ipc_flags* f, uint32_t index, uint32_t value)
{
    if (!f || index >= f->count) return fail(VRC_ERR_INVALID, "vrc_ipc_flag_set: bad argument");
    __atomic_store_n(f->host + index, value, __ATOMIC_RELEASE);
    return VRC_OK;
}

extern "C" uint32_t vrc_ipc_flag_value(const vrc_ipc_flags* f, uint32_t index)
{
    return (f && index < f->count) ? __atomic_load_n(f->host + index, __ATOMIC_ACQUIRE) : 0u;
}

extern "C" void* vrc_image_device_ptr(vrc_renderer* r) { return r ? r->d_image : nullptr; }
extern "C" void* vrc_accum_device_ptr(vrc_renderer* r) { return r ? r->d_accum : nullptr; }

extern "C" int vrc_read_image(vrc_renderer* r, uint8_t* rgba_host, void* stream)
{
    if (!r || !rgba_host) return fail(VRC_ERR_INVALID, "vrc_read_image: null argument");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipMemcpyAsync(rgba_host, r->d_image, (uint64_t)r->width * r->height * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_write_image(vrc_renderer* r, const uint8_t* rgba_host, void* stream)
{
    if (!r || !rgba_host) return fail(VRC_ERR_INVALID, "vrc_write_image: null argument");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipMemcpyAsync(r->d_image, rgba_host, (uint64_t)r->width * r->height * 4, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_read_accum(vrc_renderer* r, uint32_t* accum_host, void* stream)
{
    if (!r || !accum_host) return fail(VRC_ERR_INVALID, "vrc_read_accum: null argument");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipMemcpyAsync(accum_host, r->d_accum, (uint64_t)r->width * r->height * 16, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_get_stats(vrc_renderer* r, vrc_frame_stats* out, int reset, void* stream)
{
    if (!r || !out) return fail(VRC_ERR_INVALID, "vrc_get_stats: null argument");
    HIP_TRY(hipSetDevice(r->device));
    std::vector<uint64_t> slots(vrc::VRC_STAT_SLOTS * 8u);
    HIP_TRY(hipMemcpyAsync(slots.data(), r->d_stats, vrc::VRC_STATS_BYTES, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    uint64_t h[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < vrc::VRC_STAT_SLOTS; ++i)
        for (int k = 0; k < 5; ++k) h[k] += slots[8u * i + k];
    out->rays = h[0]; out->sum_complexity = h[1]; out->primary_hits = h[2]; out->pixels = h[3];
    out->iterations_not_executed = h[4];
    if (reset) {
        HIP_TRY(hipMemsetAsync(r->d_stats, 0, vrc::VRC_STATS_BYTES, (hipStream_t)stream));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    return VRC_OK;
}

// ---------------------------------------------------------------------------
// multi-GPU shard plumbing
// ---------------------------------------------------------------------------

static uint32_t shard_slots(uint32_t height, uint32_t row_block, uint32_t shard_count)
{
    const uint32_t nblocks = (height + row_block - 1) / row_block;
    return (nblocks + shard_count - 1) / shard_count;
}

extern "C" uint64_t vrc_shard_bytes(uint32_t width, uint32_t height, uint32_t row_block, uint32_t shard_count)
{
    if (!row_block || !shard_count) return 0;
    return (uint64_t)shard_slots(height, row_block, shard_count) * row_block * width * 4ull;
}

extern "C" int vrc_pack_shard(vrc_renderer* r, uint32_t row_block, uint32_t shard_index, uint32_t shard_count, void* dst_dev,
                              void* stream)
{
    if (!r || !dst_dev || !row_block || !shard_count || shard_index >= shard_count)
        return fail(VRC_ERR_INVALID, "vrc_pack_shard: bad argument");
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(vrc::launch_pack_shard(r->d_image, r->width, r->height, row_block, shard_index, shard_count,
                                   shard_slots(r->height, row_block, shard_count), dst_dev, (hipStream_t)stream));
    return VRC_OK;
}

extern "C" int vrc_resolve_shard(vrc_renderer* r, uint32_t row_block, uint32_t shard_index, uint32_t shard_count, void* dst_dev,
                                 int reset, void* stream)
{
    if (!r || !shard_count || shard_index >= shard_count) return fail(VRC_ERR_INVALID, "vrc_resolve_shard: bad argument");
    if (shard_count == 1 && row_block == 0) row_block = r->height;     // the whole frame as one block
    if (!row_block) return fail(VRC_ERR_INVALID, "vrc_resolve_shard: row_block is 0");
    HIP_TRY(hipSetDevice(r->device));
    uint32_t* queue = (uint32_t*)((uint8_t*)r->d_stats + vrc::VRC_STATS_BYTES);
    HIP_TRY(vrc::launch_resolve_shard(r->d_accum, r->image_target ? r->image_target : r->d_image, r->width, r->height, row_block, shard_index, shard_count,
                                      shard_slots(r->height, row_block, shard_count), dst_dev, reset ? 1u : 0u, queue,
                                      (hipStream_t)stream));
    if (reset) r->queue_zero[0] = true;     // k_resolve_shard zeroes the first set's heads in the same pass
    return VRC_OK;
}

extern "C" int vrc_unpack_shards(const void* gathered_dev, uint32_t width, uint32_t height, uint32_t row_block,
                                 uint32_t shard_count, void* image_dev, void* stream)
{
    if (!gathered_dev || !image_dev || !row_block || !shard_count || !width || !height)
        return fail(VRC_ERR_INVALID, "vrc_unpack_shards: bad argument");
    {   // no renderer here: launch on the device that owns the destination frame
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, image_dev) == hipSuccess) HIP_TRY(hipSetDevice(attr.device));
        else (void)hipGetLastError();
    }
    HIP_TRY(vrc::launch_unpack_shards(gathered_dev, width, height, row_block, shard_count,
                                      shard_slots(height, row_block, shard_count), image_dev, (hipStream_t)stream));
    return VRC_OK;
}

// The kernels replace some IEEE divisions / square roots by short sequences that are proven equal on the ranges they
// are used on; this runs that proof on the device: every float bit pattern of those ranges (~4.3 x 10^9 evaluations, a
// fraction of a second).  mismatches[0..3]: reciprocal, square root, 1 / sqrt composition, get_rand -- all must be 0.
extern "C" int vrc_selftest_exact_arith(int device, uint64_t mismatches[4])
{
    if (!mismatches) return fail(VRC_ERR_INVALID, "vrc_selftest_exact_arith: null argument");
    int rc = require_device(device, nullptr);
    if (rc) return rc;
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 32));
    hipError_t e = hipMemset(d, 0, 32);
    if (e == hipSuccess) e = vrc::launch_selftest_exact_arith(d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(mismatches, d, 32, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(VRC_ERR_HIP, "vrc_selftest_exact_arith: %s", hipGetErrorString(e));
    return VRC_OK;
}

// ---------------------------------------------------------------------------
// host helper: generateRotationMatrix (utils.cpp:94-100) = mat3(ry * rx) with
// rx = rotate(I, -angle.x, Y), ry = rotate(I, -angle.y, X); glm::rotate is the
// axis-angle form.  Columns m[0], m[1], m[2].
// ---------------------------------------------------------------------------

namespace {
struct M4 { float c[4][4]; };
M4 m4_identity()
{
    M4 m;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m.c[i][j] = (i == j) ? 1.0f : 0.0f;
    return m;
}
M4 m4_rotate(const M4& m, float angle, float ax, float ay, float az)
{
    const float c = std::cos(angle), s = std::sin(angle);
    const float inv = 1.0f / std::sqrt((ax * ax + ay * ay) + az * az);
    ax *= inv; ay *= inv; az *= inv;
    const float tx = (1.0f - c) * ax, ty = (1.0f - c) * ay, tz = (1.0f - c) * az;
    float R[3][3];
    R[0][0] = c + tx * ax;      R[0][1] = tx * ay + s * az; R[0][2] = tx * az - s * ay;
    R[1][0] = ty * ax - s * az; R[1][1] = c + ty * ay;      R[1][2] = ty * az + s * ax;
    R[2][0] = tz * ax + s * ay; R[2][1] = tz * ay - s * ax; R[2][2] = c + tz * az;
    M4 out;
    for (int i = 0; i < 3; ++i)
        for (int r = 0; r < 4; ++r) out.c[i][r] = (m.c[0][r] * R[i][0] + m.c[1][r] * R[i][1]) + m.c[2][r] * R[i][2];
    for (int r = 0; r < 4; ++r) out.c[3][r] = m.c[3][r];
    return out;
}
M4 m4_mul(const M4& A, const M4& B)
{
    M4 out;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r)
            out.c[j][r] = ((A.c[0][r] * B.c[j][0] + A.c[1][r] * B.c[j][1]) + A.c[2][r] * B.c[j][2]) + A.c[3][r] * B.c[j][3];
    return out;
}
}  // namespace

extern "C" void vrc_make_rotation(float angle_x, float angle_y, float rot[9])
{
    const M4 rx = m4_rotate(m4_identity(), -angle_x, 0.0f, 1.0f, 0.0f);
    const M4 ry = m4_rotate(m4_identity(), -angle_y, 1.0f, 0.0f, 0.0f);
    const M4 m = m4_mul(ry, rx);
    for (int j = 0; j < 3; ++j)
        for (int r = 0; r < 3; ++r) rot[j * 3 + r] = m.c[j][r];
}
