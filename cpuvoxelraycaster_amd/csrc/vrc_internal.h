// vrc_internal.h -- shared between vrc_kernels.hip (device + launchers) and
// vrc_api.cpp (C ABI).  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "../../include/vrc.h"

#define VRC_RENDER_BLOCK 256

// opaque handles of include/vrc.h
struct vrc_scene {
    int device = 0;
    int cu_count = 0;
    void* d_nodes = nullptr;
    void* d_tex = nullptr;  // 1536 bytes
    uint64_t n_nodes = 0;
    uint32_t depth = 0;
    // single-ray slot of vrc_cast_ray (Camera::getClosestPoint, once per frame): 128 bytes of pinned host memory the
    // kernel reads the ray from and writes the HitPoint to, and a stream of its own -- no allocation, no copy and no
    // device-wide synchronisation per call.  Created on first use, guarded by the mutex (a scene is shared by threads).
    std::mutex stage_mu;
    void* h_stage = nullptr;
    hipStream_t stage_stream = nullptr;
};

namespace vrc {

// shared error plumbing (vrc_api.cpp)
int fail(int code, const char* fmt, ...);
int require_device(int device, int* cu_count);

// Counters and queue heads are spread out: thousands of waves ending within microseconds of each other on one
// 128-byte line serialise in the L2 atomic unit (~88 atomics/us per line) and stretch the end of a launch.
constexpr uint32_t VRC_STAT_SLOTS = 1024;      // 64-byte slots (8 x u64, 5 used), slot = workgroup % VRC_STAT_SLOTS
constexpr uint32_t VRC_QUEUE_HEADS = 8;
constexpr uint32_t VRC_QUEUE_STRIDE = 1024;    // u32 between heads (4 KiB: separate L2 channels)
constexpr uint32_t VRC_STATS_BYTES = VRC_STAT_SLOTS * 64u;
constexpr uint32_t VRC_QUEUE_BYTES = VRC_QUEUE_HEADS * VRC_QUEUE_STRIDE * 4u;   // one set of heads; a renderer has two

// by-value kernel argument of the frame kernels (k_render_sync*)
struct FrameArgs {
    const uint2* nodes;       // LNode[] (8 B each)
    const uint8_t* tex;       // 1536 B: top table then side table
    uint8_t* image;           // RGBA8 w*h
    uint32_t* accum;          // 4 x u32 per pixel
    vrc_hit* prim;            // optional primary-hit capture
    uint64_t* stats;          // VRC_STAT_SLOTS x {rays, sum_complexity, primary_hits, pixels, iterations not executed, -, -, -}; summed by vrc_get_stats
    uint32_t* queue;          // work-queue heads, VRC_QUEUE_STRIDE u32 apart
    uint32_t depth, width, height;
    uint32_t n_items;         // 64 x the tiles of this shard's rows (a tile = the pixels of one wave)
    uint32_t sample_chunk;    // samples per work unit (0 = all spp in one unit)
    uint32_t sample_chunk_tail, tail_tiles;   // ... and for the last `tail_tiles` tiles (0 = same chunk)
    uint32_t checker_wide;    // k_render_sync, checkerboard frames: tiles are 16 x 8 pixels, every lane takes a selected pixel
    // k_render_sync*: the heads of the renderer's OTHER queue set, zeroed by this launch for the next one (the two sets
    // alternate, so no launch needs a memset or another kernel to find its queue at zero); NULL = leave it alone
    uint32_t* queue_other;
    // fused resolve (vrc_render_frame_resolved): when a tile's last work unit has added its samples it turns the tile's
    // sums into RGBA8 -- image, and resolve_dst (packed shard rows, may be NULL) -- and leaves the accumulators at zero:
    // samples_to_image + pack_shard + reset_samples without a second kernel.  tile_done: one arrival counter per tile.
    uint32_t fused_resolve;
    // pinhole kernels (vrc_renderer_set_invariant_ray_reuse): walk a work unit's sample-invariant rays (the primary ray
    // and the shadow ray of its hit) once instead of once per sample; same image, fewer walks executed and counted
    uint32_t reuse_invariant;
    // host only, measurement switch (vrc_renderer_set_walk_from_root): launch the pinhole one-bounce kernel's from-root build
    uint32_t walk_from_root;
    // host only: which build of the stage-synchronous kernel to launch, by waves per SIMD (6, 7 or 8; see sync_waves_for)
    uint32_t waves;
    // which lane <-> (pixel, sample) map of the stage-synchronous kernel: 1 = 8 x 8 pixels, 4 = 4 x 4 pixels x 4 samples abreast
    // (render_sync_body's NS); n_items then counts 64 lanes per 4 x 4 tile
    uint32_t lane_samples;
    // pinhole kernels: the sample-invariant walks quadrant by quadrant, four samples abreast (render_sync_body's QUAD); the host
    // sets it only for launches whose every unit has a multiple of four samples, without reuse and without the capture
    uint32_t quad_walks;
    uint32_t* tile_done;
    uint32_t* resolve_dst;
    vrc_camera cam;
    vrc_frame_params p;
};

hipError_t launch_cast_rays(const void* nodes, int depth, uint64_t n, const float* org, const float* dir,
                            const float* coef, const float* bias, vrc_hit* out, hipStream_t st);
hipError_t launch_cast_ray_chains(const void* nodes, int depth, uint64_t n, const float* org_a, const float* dir_a, const float* org_b,
                                  const float* dir_b, float coef_b, vrc_hit* out_a, vrc_hit* out_b, uint32_t* not_executed, hipStream_t st);
hipError_t launch_grid_cast(const uint8_t* cells, int X, int Y, int Z, uint64_t n, const float* org, const float* dir,
                            vrc_hit* out, hipStream_t st);
uint32_t sync_max_blocks_per_cu();    // register-limited workgroups per CU of k_render_sync
uint32_t quad_lds_bytes();            // LDS a workgroup of the quadrant-walk kernels needs on top
uint32_t quad_waves();                // waves per SIMD the quadrant-walk kernels are built for
bool quad_available();                // false in the variant builds that leave the quadrant-walk kernels out (launch_render would fall back)
uint32_t sync_s4_waves(bool pinhole); // waves per SIMD the samples-abreast builds were compiled for
// The stage-synchronous kernels are built for 6 waves per SIMD (80 VGPRs), the lens one-bounce kernel also for 7 (72).
// Rounds 2 and 3 shipped 7- and 8-wave builds of the pinhole kernels too, picked by a table (profiles/r03/sweep_waves.txt:
// the 8-wave build won C3 with frames in flight by 2 %).  With walks that start below the root -- fewer iterations, two
// stacks, more state per ray -- the 6-wave build ties or beats them on every configuration, pose and shard size
// (tools/sweep_waves.sh -> profiles/r03/sweep_waves_below.txt, shard_inflight_below.txt; ms per frame at pitch -0.5):
//   pinhole + GI (C3), whole-spp units, three frames in flight  6: 0.978   7: 0.981   8: 0.981      (C5, depth 10: 6 only)
//   pinhole + GI (C3), one launch at a time (automatic chunk)   6: 1.125   7: 1.121   8: 1.162      (pitch 0: 0.770 / 0.786 / 0.814)
//   lens (DOF) + GI (C4), three frames in flight                6: 1.250   7: 1.232
//   lens (DOF) + GI (C4), one at a time                         6: 1.372   7: 1.348                 (pitch 0: 0.889 / 0.899)
//   short launch (C2: 1280 x 720, primary + shadow, 1 spp)      6: 0.073   7: 0.079   8: 0.087
// so they are gone (and with them their register spills: the 6-wave pinhole kernel has one).  sync_waves_for: 6, the lens
// kernel 7 with whole-spp units; a caller's vrc_renderer_set_tuning(blocks_per_cu) >= 7 selects the lens kernel's 7-wave build.
uint32_t sync_waves_for(bool pinhole, bool one_bounce, bool use_gi, bool whole_spp_units, uint64_t pixel_samples, uint32_t requested);
bool camera_is_pinhole(const vrc_camera& cam);   // the host's predicate for the pinhole kernels (see launch_render)
hipError_t launch_render(const FrameArgs& a, uint32_t grid_blocks, hipStream_t st, const char** launched);
hipError_t launch_resolve(const void* accum, void* image, uint32_t n, hipStream_t st);
hipError_t launch_fill_u32(void* p, uint32_t value, uint64_t n, hipStream_t st);
hipError_t launch_sanitize_nodes(void* nodes, uint64_t n, hipStream_t st);   // leaf_mask &= child_mask
// Structural check of a caller-supplied LNode array (vrc_scene_create): *flags_dev (one u32, zeroed by the caller) gets
// bit 0 = a child block reaches past the array, bit 1 = a non-leaf child below the unit-voxel level (the tree is
// deeper than `depth`), bit 2 = a node reachable at two different levels (cycle / mis-shared subtree).
// level_dev: n bytes of scratch.
hipError_t launch_validate_nodes(const void* nodes, uint64_t n, uint32_t depth, uint8_t* level_dev, uint32_t* flags_dev, hipStream_t st);
hipError_t launch_selftest_exact_arith(unsigned long long* out_dev, hipStream_t st);
void renderer_info(const vrc_renderer* r, int* device, uint32_t* width, uint32_t* height);
void scene_free(vrc_scene* s);   // releases the device / pinned resources and deletes s
hipError_t launch_pack_shard(const void* image, uint32_t W, uint32_t H, uint32_t row_block, uint32_t shard,
                             uint32_t count, uint32_t slots, void* dst, hipStream_t st);
hipError_t launch_resolve_shard(void* accum, void* image, uint32_t W, uint32_t H, uint32_t row_block, uint32_t shard,
                                uint32_t count, uint32_t slots, void* dst, uint32_t reset, uint32_t* queue, hipStream_t st);
hipError_t launch_unpack_shards(const void* gathered, uint32_t W, uint32_t H, uint32_t row_block, uint32_t count,
                                uint32_t slots, void* image, hipStream_t st);

}  // namespace vrc
