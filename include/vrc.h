/*
 * vrc.h -- C ABI of the MI355X-native voxel ray-traversal hot path
 * (drop-in for the per-ray / per-frame operators of johnBuffer/CpuVoxelRaycaster).
 *
 * Plain pointers and sizes only.  All citations are file:line under the
 * reference tree.  The library is libvrc_hip.so (cpuvoxelraycaster_amd/csrc);
 * every compute entry point runs hand-written HIP kernels for gfx950 and fails
 * with VRC_ERR_NO_DEVICE when no GPU is present -- there is no CPU fallback.
 *
 * Error convention (the reference has none to inherit: volumetric.hpp:55-61
 * has no error channel): every function returns 0 on success or a negative
 * VRC_ERR_* code; vrc_last_error() returns a thread-local message.
 *
 * Threading: a vrc_scene is immutable after creation and may be shared by
 * any number of renderers / threads (the reference calls castRay concurrently
 * from 16 workers, main.cpp:139-152).  A vrc_renderer is not re-entrant.
 *
 * Streams: `stream` arguments are hipStream_t handles passed as void*
 * (NULL = the default stream).  Calls taking a stream are asynchronous on it
 * unless stated otherwise.
 */
#ifndef VRC_H
#define VRC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRC_OK                0
#define VRC_ERR_INVALID      -1   /* bad argument */
#define VRC_ERR_NO_DEVICE    -2   /* no HIP device / HIP runtime error at init */
#define VRC_ERR_HIP          -3   /* HIP runtime call failed (see vrc_last_error) */
#define VRC_ERR_OOM          -4
#define VRC_ERR_PEER         -5   /* a peer process of a direct-peer-write exchange died or timed out (vrc_ipc_stream_wait) */

#define VRC_MEM_HOST   0
#define VRC_MEM_DEVICE 1

#define VRC_MAX_DEPTH 11          /* the loop bound of lsvo.hpp:72 binds from depth 12 on */
#define VRC_MAX_NODES (1ull << 29) /* 4 GiB of LNodes: the walk addresses a node by a 32-bit byte offset */

/* include/lsvo_utils.hpp:5-18 -- the 8-byte linear octree node the kernels
 * consume: index 0 = root; a non-empty internal node i owns the 8 consecutive
 * slots starting at i + child_offset, slot k = child (x=k&1, y=k>>1&1, z=k>>2)
 * (src/lsvo_utils.cpp:29-47). */
typedef struct vrc_lnode {
    uint8_t color;
    uint8_t child_mask;
    uint8_t leaf_mask;
    uint8_t pad;
    uint32_t child_offset;
} vrc_lnode;

/* include/volumetric.hpp:7-22 (HitPoint, 48 bytes).  `const Cell* cell` is
 * replaced by `hit` + `node`: hit = kind | child_shift<<8 | scale<<16 with
 * kind 0 = miss (cell == nullptr), 1 = unit-voxel leaf (lsvo.hpp:92-95),
 * 2 = LOD cut-off (lsvo.hpp:82-85); node = index of the LNode whose child was
 * hit.  The cell is always {Solid, Grass} (lsvo.hpp:21-23).  On a miss all
 * other fields except `complexity` are zero (uninitialised in the reference). */
typedef struct vrc_hit {
    float position[3];
    float normal[3];        /* (+-1,0,0) (0,+-2,0) (0,0,+-4), lsvo.hpp:149 */
    float voxel_coord[2];
    uint32_t hit;
    uint32_t node;
    float distance;
    uint32_t complexity;    /* loop iterations, lsvo.hpp:73 */
} vrc_hit;

typedef struct vrc_scene vrc_scene;        /* device-resident LSVO + albedo tables */
typedef struct vrc_grid vrc_grid;          /* device-resident dense Grid3D */
typedef struct vrc_renderer vrc_renderer;  /* RayCaster state: framebuffer + sample accumulators */

const char *vrc_last_error(void);
int vrc_device_count(void);                /* >= 0, or VRC_ERR_NO_DEVICE */

/* ---- scene: LSVO<N> (lsvo.hpp:12-24) ---------------------------------- */

/* Copies `n_nodes` LNodes (host memory) to `device` and keeps them resident.
 * Replaces LSVO(const SVO<N>&) + compileSVO for a pre-compiled array.  The array
 * is checked once, on the device: every non-leaf child's 8-slot block must lie
 * inside the array, the tree must not be deeper than `depth` levels (children of
 * level depth-1 nodes are unit-voxel leaves, lsvo.hpp:90-95), and no node may be
 * reachable at two different levels; a malformed or truncated array is
 * VRC_ERR_INVALID, never an out-of-bounds read in the walk. */
int vrc_scene_create(const vrc_lnode *lnodes, uint64_t n_nodes, uint32_t depth,
                     int device, vrc_scene **out);
/* 16x16 RGB tables, top-down rows, as sf::Image::getPixel sees
 * res/grass_{top,side}_16x16.bmp (raycaster.hpp:53-54). */
int vrc_scene_set_textures(vrc_scene *s, const uint8_t top_rgb[768], const uint8_t side_rgb[768]);
int vrc_scene_destroy(vrc_scene *s);
uint64_t vrc_scene_node_count(const vrc_scene *s);
uint32_t vrc_scene_depth(const vrc_scene *s);

/* Scene construction that precedes the path (main.cpp:59-88):
 * builds the exact compileSVO layout (lsvo_utils.cpp:4-49) for the terrain
 * generator of main.cpp:63-76 from height[x*size+z] (the int32 `height` of
 * main.cpp:69), without a pointer tree.  Host code.  *out is released with
 * vrc_free_host. */
int vrc_build_terrain_lsvo(const int32_t *height, uint32_t depth,
                           vrc_lnode **out, uint64_t *n_nodes);
/* Same layout for an arbitrary occupancy volume: solid[(x*size+y)*size+z] != 0. */
int vrc_build_volume_lsvo(const uint8_t *solid, uint32_t depth,
                          vrc_lnode **out, uint64_t *n_nodes);
void vrc_free_host(void *p);

/* The same construction on the GPU, straight into HBM (no host array): dense
 * per-level count / rank sweeps, bit-identical output.  height / solid are host
 * buffers laid out as above; *build_ms (optional) receives the device time of the
 * build kernels.  vrc_scene_download_nodes copies the resident array back
 * (vrc_scene_node_count entries). */
int vrc_scene_build_terrain(const int32_t *height, uint32_t depth, int device, vrc_scene **out, float *build_ms);
int vrc_scene_build_volume(const uint8_t *solid, uint32_t depth, int device, vrc_scene **out, float *build_ms);
int vrc_scene_download_nodes(const vrc_scene *s, vrc_lnode *dst);

/* The scene generator's noise on the GPU: height[x*size + z] =
 * int32(64 * noise(0.75x, 0.75z) + 32) with the reference's FastNoise settings
 * (SimplexFractal FBM, 3 octaves, frequency 0.01; seed 1337 in main.cpp:61),
 * bit-identical to lib/fastnoise.  vrc_scene_build_fastnoise_terrain runs
 * main.cpp:59-88 end to end on the device: noise -> heights -> LSVO in HBM. */
int vrc_terrain_heights(int32_t seed, uint32_t size, int device, int32_t *height_host);
int vrc_scene_build_fastnoise_terrain(int32_t seed, uint32_t depth, int device, vrc_scene **out, float *build_ms);

/* ---- per-ray operator: Volumetric::castRay (volumetric.hpp:58, lsvo.hpp:33) */

/* Batch form of HitPoint castRay(position, direction, ray_size_coef,
 * ray_size_bias).  org_xyz / dir_xyz: n x 3 floats; coef / bias: n floats or
 * NULL (= 0).  `mem` says where ALL ray and output buffers live
 * (VRC_MEM_HOST: staged through the library, synchronous -- the staging block is kept
 * per device between calls, grow-only up to 1 GiB; VRC_MEM_DEVICE: used in place,
 * asynchronous on `stream`). */
int vrc_cast_rays(const vrc_scene *s, uint64_t n,
                  const float *org_xyz, const float *dir_xyz,
                  const float *coef, const float *bias,
                  vrc_hit *out, int mem, void *stream);

/* Chains of two casts, the way RayCaster::castRay chains them (raycaster.hpp:131 -> :153; :194 ->
 * :198): ray A as vrc_cast_rays casts it (coefficient and bias 0), then ray B -- origin next to A's
 * hit -- cast with ray_size_coef = coef_b (0 <= coef_b <= 0.5; bias 0) and STARTED BELOW THE ROOT on
 * the path A's walk left, exactly as the frame kernels start their shadow / GI / GI-shadow rays
 * (csrc/vrc_device.h: start_scale_next_to[_lod]); where A missed, B starts at the root.  out_b[i]
 * equals vrc_cast_rays of ray B alone bit for bit (complexity included: the iterations a start below
 * the root leaves out are counted, lsvo.hpp:73); not_executed[i] (may be NULL) = how many those were.
 * Device buffers only (as VRC_MEM_DEVICE above); asynchronous on `stream`. */
int vrc_cast_ray_chains(const vrc_scene *s, uint64_t n,
                        const float *org_a_xyz, const float *dir_a_xyz,
                        const float *org_b_xyz, const float *dir_b_xyz, float coef_b,
                        vrc_hit *out_a, vrc_hit *out_b, uint32_t *not_executed, void *stream);

/* Single-ray form for Camera::getClosestPoint (camera_controller.hpp:56-60). Synchronous; thread-safe
 * (calls on one scene are serialised).  Uses a pinned slot and a stream owned by the scene: no
 * allocation, no device-wide synchronisation, frames in flight on other streams are not disturbed. */
int vrc_cast_ray(const vrc_scene *s, const float org[3], const float dir[3],
                 float ray_size_coef, float ray_size_bias, vrc_hit *out);

/* ---- dense grid: Grid3D<X,Y,Z> (grid_3d.hpp:10-138) ------------------- */

/* cells[(x*Y + y)*Z + z] = Cell::Type (0 = Empty). */
int vrc_grid_create(const uint8_t *cells, int32_t X, int32_t Y, int32_t Z, int device, vrc_grid **out);
int vrc_grid_destroy(vrc_grid *g);
/* Grid3D::castRay(position, direction) (grid_3d.hpp:36-132), voxel units. */
int vrc_grid_cast_rays(const vrc_grid *g, uint64_t n, const float *org_xyz, const float *dir_xyz,
                       vrc_hit *out, int mem, void *stream);

/* ---- per-frame operator: RayCaster (raycaster.hpp:43-283) ------------- */

/* Camera (camera_controller.hpp:16-61).  The Camera class itself stays on the
 * host; these are the values Camera::getRay reads. */
typedef struct vrc_camera {
    float position[3];      /* world voxel units */
    float rot[9];           /* glm::mat3 rot_mat, columns m[0], m[1], m[2] */
    float fov;
    float aperture;
    float focal_length;
} vrc_camera;

typedef struct vrc_frame_params {
    float light_position[3];  /* RayCaster::setLightPosition (raycaster.hpp:62), SVO space */
    uint32_t use_gi;          /* raycaster.hpp:274 */
    uint32_t use_samples;     /* raycaster.hpp:275 */
    uint32_t shadow_samples;  /* 0 = reference default: use_samples ? 4 : 1 (raycaster.hpp:147) */
    uint32_t gi_bounces;      /* 0/1 = reference (one indirect bounce); 2 = extension */
    int32_t checker_parity;   /* -1 = every pixel; 0/1 = checker_board_offset (main.cpp:137,143) */
    uint32_t spp;             /* renderRay-equivalents per pixel in this call (1 .. 65536) */
    uint32_t seed;            /* counter-based RNG key ... */
    uint32_t frame_index;     /* ... sample s of this call uses frame_index + s */
    uint32_t row_block;       /* multi-GPU: rows per shard block (0 = whole frame) */
    uint32_t shard_index;     /* this renderer renders row blocks b with b % shard_count == shard_index */
    uint32_t shard_count;
} vrc_frame_params;

typedef struct vrc_frame_stats {
    uint64_t rays;            /* LSVO::castRay-equivalent traversals executed */
    uint64_t sum_complexity;  /* sum of HitPoint::complexity over them (8 B each = algorithmic bytes) */
    uint64_t primary_hits;
    uint64_t pixels;          /* pixel-samples shaded */
    uint64_t iterations_not_executed;  /* the part of sum_complexity the frame kernels counted without executing: a ray next to
                                        * the previous hit (or the camera) starts below the root, at the end of the descends
                                        * that lsvo.hpp:72-111 would make from the root (same walk, same count) */
} vrc_frame_stats;

/* Streams for hosts that do not link the HIP runtime themselves: every `void *stream` argument of
 * this header is a hipStream_t (NULL = the default stream).  A host that keeps frames in flight
 *  needs one stream per frame in flight. */
int vrc_stream_create(int device, void **stream);
int vrc_stream_destroy(int device, void *stream);
int vrc_stream_synchronize(int device, void *stream);   /* blocks the calling thread until the stream's work is done */

/* RayCaster(svo, render_size) (raycaster.hpp:48-60): framebuffer cleared to
 * opaque black (sf::Image::create), accumulators zero.  The scene must outlive
 * every vrc_render_frame call on the renderer (as `const LSVO<9>& svo` must
 * outlive the RayCaster, raycaster.hpp:265); the other renderer calls and
 * vrc_renderer_destroy do not touch it.  The renderer takes a snapshot of the
 * process-wide scheduling defaults (vrc_set_*) at creation. */
int vrc_renderer_create(const vrc_scene *s, uint32_t width, uint32_t height, vrc_renderer **out);
int vrc_renderer_destroy(vrc_renderer *r);

/* One frame = what the swarm lambda does (main.cpp:139-152): for every selected
 * pixel, Camera::getRay + RayCaster::renderRay, `spp` times.  Adds to the
 * renderer's running vrc_frame_stats.  Asynchronous on `stream`. */
int vrc_render_frame(vrc_renderer *r, const vrc_camera *cam, const vrc_frame_params *p, void *stream);
/* vrc_render_frame followed by vrc_resolve_shard(r, p->row_block, p->shard_index, p->shard_count, dst_dev, reset = 1)
 * -- i.e. the swarm lambda, samples_to_image() and resetSamples() of one progressive frame (main.cpp:139-158,
 * raycaster.hpp:94-116) -- as ONE launch: the tile's last work unit turns the tile's sample sums into RGBA8 (into the
 * image and, if dst_dev != NULL, into the packed shard buffer) and leaves the accumulators at zero.  Needs
 * p->use_samples and accumulators that are zero on entry (after creation, vrc_reset_samples, vrc_resolve_shard(reset)
 * or a previous call of this function); same image, same shard rows, same counters as the two calls.  Rows of dst_dev
 * that correspond to no image row (padding of the last row block / slot) are left untouched.  Frame kernels or modes
 * that cannot fuse (checkerboard) run the two calls instead. */
int vrc_render_frame_resolved(vrc_renderer *r, const vrc_camera *cam, const vrc_frame_params *p, void *dst_dev, void *stream);
/* Optional: also record the primary-ray HitPoint of sample 0 per pixel into
 * prim_dev (device memory, width*height vrc_hit) during the next frames; NULL disables. */
int vrc_renderer_set_primary_capture(vrc_renderer *r, vrc_hit *prim_dev);

int vrc_samples_to_image(vrc_renderer *r, void *stream);  /* raycaster.hpp:94-103 */
int vrc_reset_samples(vrc_renderer *r, void *stream);     /* raycaster.hpp:105-116 */
int vrc_clear_image(vrc_renderer *r, void *stream);       /* render_image.create() again */

/* render_image (raycaster.hpp:261): RGBA8 row-major, width*height*4 bytes. */
void *vrc_image_device_ptr(vrc_renderer *r);              /* for device-side gathers (RCCL) */
void *vrc_accum_device_ptr(vrc_renderer *r);              /* uint32 r,g,b,count per pixel */
int vrc_read_image(vrc_renderer *r, uint8_t *rgba_host, void *stream);   /* synchronous */
int vrc_write_image(vrc_renderer *r, const uint8_t *rgba_host, void *stream); /* synchronous */
int vrc_read_accum(vrc_renderer *r, uint32_t *accum_host, void *stream);  /* synchronous */

/* Synchronises `stream`, returns and optionally clears the running stats. */
int vrc_get_stats(vrc_renderer *r, vrc_frame_stats *out, int reset, void *stream);

/* Direct peer writes (the alternative to a collective -- "direct peer writes into the root's framebuffer";
 * include/raycaster.hpp:84,261: the one render_image every swarm worker writes its pixels into).  The presenting process
 * exports the framebuffer of a renderer (vrc_ipc_export_image); every other process of the node opens it
 * (vrc_ipc_open_image) and makes it the target of its own renderer (vrc_renderer_set_image_target): a sharded
 * vrc_render_frame_resolved / vrc_resolve_shard then writes this shard's rows of the frame where they belong in the
 * presenter's memory -- over xGMI between GPUs -- instead of into the renderer's own image: no pack, no collective, no unpack.
 * Ordering between the processes is by FRAME FLAGS: 32-bit counters in a POSIX shared-memory segment (vrc_ipc_flags_open:
 * one process creates "/name", the others open it) that are written and waited for in stream order -- a writer sets "my
 * rows of frame n are in" behind its frame (vrc_stream_write_flag), the presenter's stream waits until the flag is >= n
 * (vrc_stream_wait_flag) before it reads the frame, and the other way round before a writer reuses a framebuffer.  A wait
 * names a value, so no host-side handshake is needed.  The image handle is 64 opaque bytes to pass by any host channel. */
typedef struct vrc_ipc_handle { unsigned char opaque[64]; } vrc_ipc_handle;
typedef struct vrc_ipc_flags vrc_ipc_flags;
int vrc_ipc_export_image(vrc_renderer *r, vrc_ipc_handle *out);
int vrc_ipc_open_image(int device, const vrc_ipc_handle *handle, void **image_dev);
int vrc_ipc_close_image(int device, void *image_dev);
/* image_dev: RGBA8 width x height of this renderer's size on any device this one can reach; NULL = the renderer's own image. */
int vrc_renderer_set_image_target(vrc_renderer *r, void *image_dev);
/* create = 1: makes the segment (zeros) and records this process as its owner; a segment of that name that already
 * exists is replaced only when its owner is gone (a run that died) -- one that is still in use is VRC_ERR_INVALID, so give
 * concurrent runs different names (a pid or a token in it).  create = 0: opens it; a segment that is smaller than `count`
 * asks for or was made for another count is VRC_ERR_INVALID, never a fault on first touch. */
int vrc_ipc_flags_open(const char *name, uint32_t count, int device, int create, vrc_ipc_flags **out);
/* Host-side store to a flag (release order; e.g. an "everyone may leave" word the processes poll with vrc_ipc_flag_value). */
int vrc_ipc_flag_set(vrc_ipc_flags *f, uint32_t index, uint32_t value);
/* A stream-ordered wait (vrc_stream_wait_flag) HAS NO TIMEOUT: if the process that should write the flag died, the stream --
 * and every hipStreamSynchronize on it -- waits for ever.  So a host never synchronises such a stream blindly; it calls
 * vrc_ipc_stream_wait: polls `stream` until everything enqueued on it has completed (VRC_OK), or until one of the peers
 * `pids` (process ids on this node, n_pids of them; may be NULL) no longer exists, or NO FLAG of the segment has changed for `timeout_ms` (an inactivity limit, not a deadline from the
 * start of the call: a healthy exchange that is still draining a long queue keeps moving its flags; 0 = no limit: only the
 * peers' disappearance ends the wait), or another process has already given up on this segment.  On those three it gives up for everyone: marks the segment and writes
 * 0xffffffff into every flag from the host, which releases every wait on them in every process, holds them there while the
 * stream drains (the flag writes still queued behind the waits would lower them again) and returns VRC_ERR_PEER; call it for
 * every stream of the process before leaving, so that no wait stays pending on the device (frames completed after that are not valid; vrc_stream_wait_flag / _write_flag on a marked segment
 * fail with VRC_ERR_PEER as well).  Start a fresh process to try again. */
int vrc_ipc_stream_wait(vrc_ipc_flags *f, void *stream, const int32_t *pids, uint32_t n_pids, uint32_t timeout_ms);
/* The creator removes the segment's NAME once every process of the run has opened it (the caller's barrier says when): the
 * mappings stay valid, and a run that is killed afterwards leaves nothing behind in /dev/shm. */
int vrc_ipc_flags_unlink(vrc_ipc_flags *f);
int vrc_ipc_flags_close(vrc_ipc_flags *f);                     /* the creator also removes the name, if it has not yet */
int vrc_stream_write_flag(vrc_ipc_flags *f, uint32_t index, uint32_t value, void *stream);
int vrc_stream_wait_flag(vrc_ipc_flags *f, uint32_t index, uint32_t value, void *stream);      /* until flag >= value */
uint32_t vrc_ipc_flag_value(const vrc_ipc_flags *f, uint32_t index);                           /* host read */

/* Multi-GPU frame sharding: compact this shard's row blocks into
 * `dst_dev` (ceil(nblocks/shard_count) blocks of row_block*width*4 bytes), and
 * the inverse scatter of an all-gathered buffer [shard][slot] into a full frame. */
uint64_t vrc_shard_bytes(uint32_t width, uint32_t height, uint32_t row_block, uint32_t shard_count);
int vrc_pack_shard(vrc_renderer *r, uint32_t row_block, uint32_t shard_index, uint32_t shard_count,
                   void *dst_dev, void *stream);
/* vrc_samples_to_image + vrc_pack_shard (+ vrc_reset_samples when `reset` != 0) for this shard's row blocks in
 * one pass: resolves the accumulators of the shard's pixels into the image and, when dst_dev != NULL, into the
 * packed shard buffer; with `reset` the same pixels' accumulators are zeroed for the next frame (a following
 * vrc_render_frame must be ordered after this call, as it would after vrc_reset_samples).
 * shard_count == 1 with row_block == 0 covers the whole frame (dst_dev layout == image layout). */
int vrc_resolve_shard(vrc_renderer *r, uint32_t row_block, uint32_t shard_index, uint32_t shard_count,
                      void *dst_dev, int reset, void *stream);
/* Launches on the device that owns image_dev. */
int vrc_unpack_shards(const void *gathered_dev, uint32_t width, uint32_t height, uint32_t row_block,
                      uint32_t shard_count, void *image_dev, void *stream);

/* ---- post-process / present chain that follows the path (main.cpp:160-182) ---------- */

/* Device-resident restatement of what main.cpp does with render_image every frame, so that the frame stays in
 * HBM between the renderer and whatever shows it (no PCIe read-back):
 *   render_tex   = render_image x Color(255 * (1 - old))                        (sf::BlendMultiply, :163-172)
 *   denoised_tex = denoised_tex x Color(255 * old) + render_tex, saturating     (BlendMultiply, BlendAdd, :161-162,175-177)
 *   window       = denoised_tex scaled to out_width x out_height, nearest       (final_sprite.setScale(1 / 0.75), :179-182)
 * with old = old_value_conservation (main.cpp:161: use_samples ? 0 : 0.1), and an optional per-channel median around
 * the sampled texel: median = 0 none, 3 = the 3x3 network of res/median_3.frag, 5 = the 5x5 network of
 * res/median.frag as shipped (taps one texel apart, clamped to the edge).  UNORM8 products are rounded to nearest,
 * round(a * b / 255) -- fixed-function GL blending is not specified bit-exactly, so against a real GL device the two
 * blend steps carry a tolerance of +-1 LSB each; upscale and median are exact.  The presenter owns denoised_tex
 * (zero at creation, persistent across frames) and the window image. */
typedef struct vrc_presenter vrc_presenter;
int vrc_presenter_create(int device, uint32_t width, uint32_t height, uint32_t out_width, uint32_t out_height,
                         vrc_presenter **out);
int vrc_presenter_destroy(vrc_presenter *p);
/* One frame of the chain on the renderer's framebuffer (same device, same size).  Asynchronous on `stream`. */
int vrc_present(vrc_presenter *p, vrc_renderer *r, float old_value_conservation, uint32_t median, void *stream);
/* The same on any device-resident RGBA8 image of the presenter's render size (e.g. a gathered multi-GPU frame). */
int vrc_present_image(vrc_presenter *p, const void *image_dev, float old_value_conservation, uint32_t median, void *stream);
int vrc_presenter_clear(vrc_presenter *p, void *stream);          /* denoised_tex = 0 */
void *vrc_presenter_window_ptr(vrc_presenter *p);                 /* RGBA8 out_width x out_height, device */
void *vrc_presenter_denoised_ptr(vrc_presenter *p);               /* RGBA8 width x height, device */
/* Copies the window image and / or denoised_tex to host memory (either may be NULL).  Synchronous.  `stream` must be the
 * stream of the vrc_present call whose result is wanted (or one ordered after it): streams made by vrc_stream_create are
 * non-blocking, so a read on the NULL stream does not wait for a present issued on one of them. */
int vrc_presenter_read(vrc_presenter *p, uint8_t *window_rgba_host, uint8_t *denoised_rgba_host, void *stream);

/* Self-test of the arithmetic shortcuts: the kernels compute -1 / |d| (lsvo.hpp:47), glm::normalize's 1 / sqrt and
 * getRand's x / 100 (utils.cpp:77-81) by v_rcp_f32 / v_rsq_f32 + one FMA correction step where that is proven equal to
 * the correctly rounded IEEE result, and by the IEEE expansion elsewhere.  This runs the proof on `device`: every float
 * bit pattern of the ranges concerned against the IEEE operations.  mismatches[0..3] = reciprocal, square root,
 * 1 / sqrt, get_rand; all zero on a conforming device. */
int vrc_selftest_exact_arith(int device, uint64_t mismatches[4]);

/* Host helper: generateRotationMatrix (utils.cpp:94-100) for Camera::setViewAngle. */
void vrc_make_rotation(float angle_x, float angle_y, float rot[9]);

/* Scheduling knobs.  vrc_set_* change the process-wide DEFAULTS that new renderers copy at
 * creation (thread-safe); vrc_renderer_set_* change one renderer (a renderer is not re-entrant,
 * so call them between its frames).  Results never depend on them.
 * blocks_per_cu: resident 256-thread workgroups of the frame kernel per CU (= waves per SIMD),
 * 0 = the library's choice, at most 8 -- see vrc_renderer_last_kernel below for the builds. */
int vrc_set_tuning(uint32_t blocks_per_cu);
/* Sample mode: samples per work unit (tile x sample chunk).  0 =
 * automatic: the largest chunk that still gives every wave a few dozen units (down to 1 sample
 * per unit for small multi-GPU shards), and half that chunk for the tiles handed out last, so
 * that units get shorter towards the end of a launch; a pixel's samples split over several
 * units are accumulated with integer atomics (same sums).  Values >= 0xffff0000 set the
 * automatic mode's tail policy instead: low 16 bits = units per wave with the halved chunk. */
int vrc_set_sample_chunk(uint32_t samples_per_unit);
/* The stage-synchronous kernel is built for 6 resident workgroups per CU (= waves per SIMD) and, for
 * cameras with a lens, also for 7; the library picks by the kind of launch (csrc/vrc_internal.h:
 * sync_waves_for), and blocks_per_cu >= 7 picks the 7-wave build where there is one (fewer than 6:
 * the 6-wave build on fewer workgroups).
 * vrc_renderer_last_kernel: the symbol of the frame kernel the renderer's last frame launched
 * (what a profile of the run lists), "" before the first frame. */
const char *vrc_renderer_last_kernel(const vrc_renderer *r);
int vrc_renderer_set_tuning(vrc_renderer *r, uint32_t blocks_per_cu);
int vrc_renderer_set_sample_chunk(vrc_renderer *r, uint32_t samples_per_unit);
/* Beyond the reference (off by default).  With aperture exactly +0 -- the reference's default
 * camera -- a pixel's primary ray and the shadow ray(s) of its hit are the same for every
 * sample, and RayCaster::renderRay (raycaster.hpp:119-167) walks them again each time.  on = 1
 * lets the stage-synchronous kernel walk them ONCE per work unit (tile x sample chunk: once per
 * pixel and frame with vrc_renderer_set_sample_chunk(r, spp)) and share the result between the
 * unit's samples; the GI rays stay per sample.  Image, accumulators and primary capture are bit
 * for bit the same; vrc_frame_stats.rays / sum_complexity then count the walks EXECUTED, i.e.
 * fewer than the reference's.  No effect with aperture != 0. */
int vrc_renderer_set_invariant_ray_reuse(vrc_renderer *r, uint32_t on);
/* Measurement / A-B switch (off by default): on = 1 makes the frame kernel start EVERY ray at the
 * root, as lsvo.hpp:60-72 does, instead of below it next to the previous hit / the camera
 *  -- for every kind of frame (pinhole or lens, one bounce or the 2-bounce
 * extension; the builds of the 8 x 8 lane map at 6 waves per SIMD, vrc_renderer_last_kernel ends in
 * "_from_root").  Same image, same vrc_frame_stats.rays / sum_complexity either way;
 * iterations_not_executed is 0 with it.  (bench.py reports the frame time with it as
 * extra.every_ray_from_the_root.) */
int vrc_renderer_set_walk_from_root(vrc_renderer *r, uint32_t on);
/* The lane <-> (pixel, sample) map of the stage-synchronous kernel.  The reference's own map of
 * pixels to workers is the static 4 x 4 area grid of main.cpp:140-143; results do not depend on
 * it there or here.  samples = 1: a wave takes 8 x 8 pixels and loops over the samples of its work
 * unit.  samples = 4: a wave takes 4 x 4 pixels with four samples of each side by side (the four
 * lanes of a pixel share its primary hit: a stage's longest ray is the longest of 16 neighbouring
 * pixels', not of 64) -- for sample-mode frames (use_samples) with spp a multiple of 4, one GI
 * bounce and no checkerboard; other frames use map 1 whatever is set.  samples = 0 (default): the
 * library's choice -- four abreast for a launch that has the chip to itself (3-6 % off a frame's
 * latency), the pixel tiles for whole-spp work units (vrc_renderer_set_sample_chunk(r, spp): frames
 * in flight, where they are 5 % faster).  Image, accumulators, captures and vrc_frame_stats are the
 * same either way. */
int vrc_renderer_set_lane_samples(vrc_renderer *r, uint32_t samples);
int vrc_set_lane_samples(uint32_t samples);                 /* the process default new renderers copy */
/* Quadrant walks (round 5; on by default).  With a pinhole camera a pixel's primary ray and the
 * shadow ray of its hit are the same for every sample, and the reference walks them once per sample
 * (raycaster.hpp:131,153).  on = 1: where a launch allows it the pinhole kernels lay those walks out
 * as one quadrant of the 8 x 8 tile at a time, 16 pixels with their four samples side by side in the
 * wave, instead of 64 pixels with one sample -- every ray is still walked, each by its own lane, but a
 * stage lasts as long as the longest ray of 16 neighbouring pixels instead of 64 (C3: 6.5 % fewer
 * instructions, 5.5 % less time; 16 spp: 11 %).  Applies to sample-mode frames on the 8 x 8 lane map whose work
 * units all have a multiple of four samples (spp and the sample chunk multiples of four), without
 * invariant-ray reuse and primary-hit capture, on trees of 8 levels or more; every other launch runs
 * the plain kernels (vrc_renderer_last_kernel ends in "_q" when it applied).  on = 0: never.  Image,
 * accumulators and vrc_frame_stats are the same either way. */
int vrc_renderer_set_quad_walks(vrc_renderer *r, uint32_t on);

#ifdef __cplusplus
}
#endif
#endif
