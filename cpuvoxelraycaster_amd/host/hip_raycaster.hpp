// hip_raycaster.hpp -- C++ host adapter over the C ABI (include/vrc.h) that
// presents the reference's operator interface for the hot path:
//
//   vrc_host::HipLSVO       <- LSVO<N> : Volumetric   (include/lsvo.hpp:10-33,
//                              include/volumetric.hpp:55-61): castRay / setCell
//   vrc_host::HipRayCaster  <- RayCaster (include/raycaster.hpp:43-283):
//                              setLightPosition, renderRay's per-frame batch
//                              (renderFrame), samples_to_image, resetSamples,
//                              public flags use_gi / use_samples, render_image
//
// Header-only, C++14, no GLM / SFML needed.  When the reference's own headers
// are on the include path, define VRC_WITH_REFERENCE_HEADERS before including
// this file to also get `HipVolumetric`, a real `Volumetric` subclass that can
// be handed to Camera::getClosestPoint and friends unchanged.
#pragma once

#include <cstdint>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vrc.h"

namespace vrc_host {

struct Vec3 { float x, y, z; };   // layout-compatible with glm::vec3
struct Vec2 { float x, y; };

// include/cell.hpp:3-24 -- every LSVO hit reports this one cell (lsvo.hpp:21-23)
struct Cell {
    enum Type { Empty, Solid, Mirror };
    enum Texture { None, Grass, Red, White };
    Type type = Solid;
    Texture texture = Grass;
};

// include/volumetric.hpp:7-22
struct HitPoint {
    Vec3 position{0, 0, 0};
    Vec3 normal{0, 0, 0};
    Vec2 voxel_coord{0, 0};
    const Cell* cell = nullptr;   // nullptr = miss
    float distance = 0.0f;
    uint32_t complexity = 0u;
};

inline void check(int rc, const char* what)
{
    if (rc != VRC_OK) throw std::runtime_error(std::string(what) + ": " + vrc_last_error());
}

// The albedo tables as the reference gets them: sf::Image::loadFromFile("res/grass_top_16x16.bmp") and getPixel(x, y)
// (raycaster.hpp:53-54,239) -- an uncompressed 24- or 32-bpp BMP decoded to RGB rows, top row first.  Returns
// width * height * 3 bytes (768 for the reference's 16 x 16 files); throws on anything else.
inline std::vector<uint8_t> loadBMP(const std::string& path, uint32_t* width = nullptr, uint32_t* height = nullptr)
{
    std::ifstream f(path, std::ios::binary);
    std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto u32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
    auto u16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') throw std::runtime_error("loadBMP: not a BMP file: " + path);
    const uint32_t off = u32(10), bpp = u16(28), comp = u32(30);
    const int32_t w = (int32_t)u32(18), hs = (int32_t)u32(22);
    const uint32_t h = (uint32_t)(hs < 0 ? -hs : hs);
    if (w <= 0 || w > 16384 || h == 0 || h > 16384 || (bpp != 24 && bpp != 32) || (comp != 0 && !(comp == 3 && bpp == 32)))
        throw std::runtime_error("loadBMP: only uncompressed 24 / 32 bpp is supported: " + path);
    const size_t stride = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
    if (d.size() < off + stride * h) throw std::runtime_error("loadBMP: truncated file: " + path);
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    for (uint32_t row = 0; row < h; ++row) {
        const uint32_t y = hs < 0 ? row : h - 1 - row;          // positive height: the file's first row is the bottom one
        const uint8_t* line = d.data() + off + stride * row;
        for (int32_t x = 0; x < w; ++x) {
            const uint8_t* px = line + (size_t)x * (bpp / 8);    // B, G, R [, A]
            uint8_t* o = rgb.data() + ((size_t)y * w + x) * 3;
            o[0] = px[2]; o[1] = px[1]; o[2] = px[0];
        }
    }
    if (width) *width = (uint32_t)w;
    if (height) *height = h;
    return rgb;
}

// LSVO<N> (lsvo.hpp:10): immutable, device resident; castRay is const and may be
// called from any number of threads, like the reference's (main.cpp:139-152).
class HipLSVO {
public:
    // LSVO(const SVO<N>&) + compileSVO replaced by a pre-compiled LNode array
    HipLSVO(const vrc_lnode* lnodes, uint64_t n_nodes, uint32_t depth, int device = 0)
    {
        check(vrc_scene_create(lnodes, n_nodes, depth, device, &scene_), "vrc_scene_create");
    }
    // main.cpp:59-88 on the device: FastNoise heights (seed as main.cpp:61) -> voxels -> LSVO, straight into HBM
    static std::unique_ptr<HipLSVO> fromFastNoiseTerrain(int32_t seed, uint32_t depth, int device = 0, float* build_ms = nullptr)
    {
        vrc_scene* s = nullptr;
        check(vrc_scene_build_fastnoise_terrain(seed, depth, device, &s, build_ms), "vrc_scene_build_fastnoise_terrain");
        return std::unique_ptr<HipLSVO>(new HipLSVO(s));
    }
    ~HipLSVO() { vrc_scene_destroy(scene_); }
    HipLSVO(const HipLSVO&) = delete;
    HipLSVO& operator=(const HipLSVO&) = delete;

    // lsvo.hpp:26 -- a no-op in the reference too
    void setCell(Cell::Type, Cell::Texture, uint32_t, uint32_t, uint32_t) {}

    // lsvo.hpp:33
    HitPoint castRay(const Vec3& position, Vec3 d, const float ray_size_coef = 0.0f, const float ray_size_bias = 0.0f) const
    {
        vrc_hit h;
        const float o[3] = {position.x, position.y, position.z}, dir[3] = {d.x, d.y, d.z};
        check(vrc_cast_ray(scene_, o, dir, ray_size_coef, ray_size_bias, &h), "vrc_cast_ray");
        return convert(h);
    }

    // batch form: what a GPU-backed Volumetric needs to be useful
    std::vector<HitPoint> castRays(const std::vector<Vec3>& positions, const std::vector<Vec3>& directions,
                                   float ray_size_coef = 0.0f, float ray_size_bias = 0.0f) const
    {
        if (positions.size() != directions.size()) throw std::invalid_argument("castRays: size mismatch");
        const uint64_t n = positions.size();
        std::vector<vrc_hit> raw(n);
        std::vector<float> coef(n, ray_size_coef), bias(n, ray_size_bias);
        check(vrc_cast_rays(scene_, n, &positions[0].x, &directions[0].x, coef.data(), bias.data(), raw.data(),
                            VRC_MEM_HOST, nullptr), "vrc_cast_rays");
        std::vector<HitPoint> out(n);
        for (uint64_t i = 0; i < n; ++i) out[i] = convert(raw[i]);
        return out;
    }

    // pairs of casts as RayCaster::castRay chains them (raycaster.hpp:131 -> :153; :194 -> :198), device buffers: ray A, then
    // ray B next to A's hit, started below the root as the frame kernels start their secondary rays (vrc_cast_ray_chains);
    // hits_b[i] equals a cast of ray B alone.  Asynchronous on `stream`.
    void castRayChainsDevice(uint64_t n, const float* org_a_dev, const float* dir_a_dev, const float* org_b_dev, const float* dir_b_dev,
                             float ray_size_coef_b, vrc_hit* hits_a_dev, vrc_hit* hits_b_dev, uint32_t* not_executed_dev = nullptr,
                             void* stream = nullptr) const
    {
        check(vrc_cast_ray_chains(scene_, n, org_a_dev, dir_a_dev, org_b_dev, dir_b_dev, ray_size_coef_b, hits_a_dev, hits_b_dev,
                                  not_executed_dev, stream), "vrc_cast_ray_chains");
    }

    // RayCaster's constructor loads the two 16 x 16 tables relative to the working directory (raycaster.hpp:53-54);
    // here the caller names the files
    void loadTextures(const std::string& top_bmp, const std::string& side_bmp)
    {
        uint32_t w = 0, h = 0;
        const std::vector<uint8_t> top = loadBMP(top_bmp, &w, &h);
        if (w != 16 || h != 16) throw std::runtime_error("loadTextures: " + top_bmp + " is not 16 x 16");
        const std::vector<uint8_t> side = loadBMP(side_bmp, &w, &h);
        if (w != 16 || h != 16) throw std::runtime_error("loadTextures: " + side_bmp + " is not 16 x 16");
        check(vrc_scene_set_textures(scene_, top.data(), side.data()), "vrc_scene_set_textures");
    }

    vrc_scene* handle() const { return scene_; }
    uint32_t depth() const { return vrc_scene_depth(scene_); }
    Cell cell;   // lsvo.hpp:289

private:
    HitPoint convert(const vrc_hit& h) const
    {
        HitPoint p;
        p.position = {h.position[0], h.position[1], h.position[2]};
        p.normal = {h.normal[0], h.normal[1], h.normal[2]};
        p.voxel_coord = {h.voxel_coord[0], h.voxel_coord[1]};
        p.cell = h.hit ? &cell : nullptr;
        p.distance = h.distance;
        p.complexity = h.complexity;
        return p;
    }
    explicit HipLSVO(vrc_scene* adopted) : scene_(adopted) {}
    vrc_scene* scene_ = nullptr;
};

// Camera values Camera::getRay reads (camera_controller.hpp:16-49); the Camera /
// FlyController classes themselves stay untouched on the host.
struct CameraState {
    Vec3 position{0, 0, 0};
    Vec2 view_angle{0, 0};
    float aperture = 0.0f, focal_length = 1.0f, fov = 1.0f;
    vrc_camera to_abi() const
    {
        vrc_camera c;
        c.position[0] = position.x; c.position[1] = position.y; c.position[2] = position.z;
        vrc_make_rotation(view_angle.x, view_angle.y, c.rot);   // generateRotationMatrix, utils.cpp:94-100
        c.fov = fov; c.aperture = aperture; c.focal_length = focal_length;
        return c;
    }
};

// RayCaster (raycaster.hpp:43-283).  render_image and the Sample accumulators
// live in HBM; renderFrame replaces the swarm lambda of main.cpp:139-152.
class HipRayCaster {
public:
    HipRayCaster(const HipLSVO& svo, uint32_t width, uint32_t height) : width_(width), height_(height)
    {
        check(vrc_renderer_create(svo.handle(), width, height, &r_), "vrc_renderer_create");
    }
    ~HipRayCaster() { vrc_renderer_destroy(r_); }
    HipRayCaster(const HipRayCaster&) = delete;
    HipRayCaster& operator=(const HipRayCaster&) = delete;

    void setLightPosition(const Vec3& position) { light_ = position; }   // raycaster.hpp:62

    // One frame: Camera::getRay + renderRay for every selected pixel (main.cpp:139-152).
    // checker_board_offset = -1 renders every pixel; 0 / 1 as main.cpp:137,143.
    // Multi-GPU: this process renders the row blocks b (of `row_block` rows) with b % shard_count == shard_index.
    void setShard(uint32_t row_block, uint32_t shard_index, uint32_t shard_count)
    {
        row_block_ = row_block; shard_index_ = shard_index; shard_count_ = shard_count ? shard_count : 1;
    }

    void renderFrame(const CameraState& camera, int32_t checker_board_offset = -1, uint32_t spp = 1, void* stream = nullptr)
    {
        vrc_frame_params p{};
        p.row_block = row_block_; p.shard_index = shard_index_; p.shard_count = shard_count_;
        p.light_position[0] = light_.x; p.light_position[1] = light_.y; p.light_position[2] = light_.z;
        p.use_gi = use_gi; p.use_samples = use_samples;
        p.shadow_samples = 0; p.gi_bounces = 1;
        p.checker_parity = checker_board_offset; p.spp = spp;
        p.seed = seed; p.frame_index = frame_index_;
        const vrc_camera c = camera.to_abi();
        check(vrc_render_frame(r_, &c, &p, stream), "vrc_render_frame");
        frame_index_ += spp;
    }

    // One progressive frame in one launch: renderFrame (sample mode, every pixel) + samples_to_image + resetSamples, the
    // resolved rows also written to `shard_dev` (device, may be null) as the multi-GPU exchange expects them.
    void renderFrameResolved(const CameraState& camera, uint32_t spp, void* shard_dev = nullptr, void* stream = nullptr)
    {
        vrc_frame_params p{};
        p.row_block = row_block_; p.shard_index = shard_index_; p.shard_count = shard_count_;
        p.light_position[0] = light_.x; p.light_position[1] = light_.y; p.light_position[2] = light_.z;
        p.use_gi = use_gi; p.use_samples = 1;
        p.shadow_samples = 0; p.gi_bounces = 1;
        p.checker_parity = -1; p.spp = spp;
        p.seed = seed; p.frame_index = frame_index_;
        const vrc_camera c = camera.to_abi();
        check(vrc_render_frame_resolved(r_, &c, &p, shard_dev, stream), "vrc_render_frame_resolved");
        frame_index_ += spp;
    }

    // Frames kept in flight: a work unit = all `spp` samples of a tile ...
    void setSampleChunk(uint32_t samples_per_unit) { check(vrc_renderer_set_sample_chunk(r_, samples_per_unit), "vrc_renderer_set_sample_chunk"); }
    // ... and, beyond the reference, the sample-invariant primary / shadow walks of a pinhole camera done once per unit
    void setInvariantRayReuse(bool on) { check(vrc_renderer_set_invariant_ray_reuse(r_, on ? 1u : 0u), "vrc_renderer_set_invariant_ray_reuse"); }
    void setLaneSamples(uint32_t samples) { check(vrc_renderer_set_lane_samples(r_, samples), "vrc_renderer_set_lane_samples"); }
    void setQuadWalks(bool on) { check(vrc_renderer_set_quad_walks(r_, on ? 1u : 0u), "vrc_renderer_set_quad_walks"); }
    void setWalkFromRoot(bool on) { check(vrc_renderer_set_walk_from_root(r_, on ? 1u : 0u), "vrc_renderer_set_walk_from_root"); }

    void samples_to_image(void* stream = nullptr) { check(vrc_samples_to_image(r_, stream), "vrc_samples_to_image"); }   // raycaster.hpp:94
    void resetSamples(void* stream = nullptr) { check(vrc_reset_samples(r_, stream), "vrc_reset_samples"); }            // raycaster.hpp:105
    // samples_to_image for this shard's rows, written into `shard_dev` (device, vrc_shard_bytes) as the all-gather
    // expects it; `reset` also clears those rows' accumulators (= resetSamples) in the same pass
    void resolveShard(void* shard_dev, bool reset, void* stream = nullptr)
    {
        check(vrc_resolve_shard(r_, row_block_, shard_index_, shard_count_, shard_dev, reset ? 1 : 0, stream), "vrc_resolve_shard");
    }

    // render_image (raycaster.hpp:261) copied to host RGBA8, row-major
    std::vector<uint8_t> render_image() const
    {
        std::vector<uint8_t> img((size_t)width_ * height_ * 4);
        check(vrc_read_image(r_, img.data(), nullptr), "vrc_read_image");
        return img;
    }

    vrc_frame_stats stats(bool reset = false)
    {
        vrc_frame_stats s;
        check(vrc_get_stats(r_, &s, reset ? 1 : 0, nullptr), "vrc_get_stats");
        return s;
    }

    // Direct peer writes: the presenting process exports its framebuffer, the others resolve
    // their shards of the frame straight into it
    vrc_ipc_handle exportImage() const { vrc_ipc_handle h; check(vrc_ipc_export_image(r_, &h), "vrc_ipc_export_image"); return h; }
    void setImageTarget(void* image_dev) { check(vrc_renderer_set_image_target(r_, image_dev), "vrc_renderer_set_image_target"); }

    vrc_renderer* handle() const { return r_; }
    // sample counter of the next frame (the RNG key: frame_index + sample); renderFrame* advance it by spp
    void setFrameIndex(uint32_t frame_index) { frame_index_ = frame_index; }
    uint32_t frameIndex() const { return frame_index_; }
    std::vector<uint8_t> render_image(void* stream) const     // the same copy, ordered after (and waiting for) `stream` only
    {
        std::vector<uint8_t> img((size_t)width_ * height_ * 4);
        check(vrc_read_image(r_, img.data(), stream), "vrc_read_image");
        return img;
    }
    vrc_frame_stats stats(bool reset, void* stream)
    {
        vrc_frame_stats s;
        check(vrc_get_stats(r_, &s, reset ? 1 : 0, stream), "vrc_get_stats");
        return s;
    }

    bool use_gi = false;        // raycaster.hpp:274
    bool use_samples = false;   // raycaster.hpp:275
    uint32_t seed = 0x9E3779B9u;

private:
    vrc_renderer* r_ = nullptr;
    uint32_t width_, height_;
    uint32_t frame_index_ = 0;
    uint32_t row_block_ = 0, shard_index_ = 0, shard_count_ = 1;
    Vec3 light_{0, 0, 0};
};

// Progressive frames kept in flight -- for hosts that do not need frame i before they issue
// frame i + 1: an offline render, a replay (readReplay below), a multi-GPU shard.  `frames_in_flight` renderers and
// streams take turns; a work unit is a tile's whole sample set and the resolve is fused into the frame kernel, which is the
// fastest form once launches overlap.  submit() returns at once; image(slot) waits for that slot's frame only.  Frame n
// is the frame a single HipRayCaster would render n-th with the same spp (frame_index = n * spp), bit for bit.
class HipFramePipeline {
public:
    HipFramePipeline(const HipLSVO& svo, uint32_t width, uint32_t height, uint32_t spp, uint32_t frames_in_flight = 3, int device = 0)
        : device_(device), spp_(spp ? spp : 1)
    {
        if (frames_in_flight == 0) frames_in_flight = 1;
        for (uint32_t i = 0; i < frames_in_flight; ++i) {
            slots_.emplace_back(new HipRayCaster(svo, width, height));
            slots_.back()->setSampleChunk(spp_);
            void* st = nullptr;
            check(vrc_stream_create(device, &st), "vrc_stream_create");
            streams_.push_back(st);
        }
    }
    ~HipFramePipeline()
    {
        for (void* st : streams_) { vrc_stream_synchronize(device_, st); vrc_stream_destroy(device_, st); }
    }
    HipFramePipeline(const HipFramePipeline&) = delete;
    HipFramePipeline& operator=(const HipFramePipeline&) = delete;

    void setLightPosition(const Vec3& position) { for (auto& r : slots_) r->setLightPosition(position); }
    void setUseGI(bool on) { for (auto& r : slots_) r->use_gi = on; }
    void setSeed(uint32_t seed) { for (auto& r : slots_) r->seed = seed; }
    void setInvariantRayReuse(bool on) { for (auto& r : slots_) r->setInvariantRayReuse(on); }
    void setShard(uint32_t row_block, uint32_t shard_index, uint32_t shard_count)
    {
        for (auto& r : slots_) r->setShard(row_block, shard_index, shard_count);
    }
    uint32_t framesInFlight() const { return (uint32_t)slots_.size(); }

    // Issues frame number `submitted()` and returns its slot.  The slot's previous frame (F submits ago) is overwritten:
    // read it first.  `shard_dev`: where the resolved rows also go (device; multi-GPU exchange buffer), may be null.
    uint32_t submit(const CameraState& camera, void* shard_dev = nullptr)
    {
        const uint32_t slot = (uint32_t)(n_ % slots_.size());
        slots_[slot]->setFrameIndex((uint32_t)(n_ * spp_));
        slots_[slot]->renderFrameResolved(camera, spp_, shard_dev, streams_[slot]);
        ++n_;
        return slot;
    }
    uint64_t submitted() const { return n_; }
    uint32_t nextSlot() const { return (uint32_t)(n_ % slots_.size()); }     // the slot (and stream) the next submit() uses
    void wait(uint32_t slot) { check(vrc_stream_synchronize(device_, streams_[slot]), "vrc_stream_synchronize"); }
    void waitAll() { for (uint32_t i = 0; i < slots_.size(); ++i) wait(i); }
    // waitAll() for streams that wait for peers' frame flags (direct peer writes): never a blind synchronize -- a stream-ordered
    // wait has no timeout.  Polls the streams, the peers' process ids and a deadline (vrc_ipc_stream_wait); when a peer is gone
    // every wait on the flags is released in every process and this throws (VRC_ERR_PEER).
    void waitAllWatched(vrc_ipc_flags* flags, const std::vector<int32_t>& peer_pids, uint32_t timeout_ms = 120000)
    {
        int first = 0;
        std::string why;
        for (void* st : streams_) {
            // after a failure the remaining streams are still drained (each call holds the release while its stream empties): a
            // process must not leave with a wait pending on the device
            const int rc = vrc_ipc_stream_wait(flags, st, peer_pids.data(), (uint32_t)peer_pids.size(), timeout_ms);
            if (rc != 0 && first == 0) { first = rc; why = vrc_last_error(); }
        }
        if (first != 0) throw std::runtime_error("vrc_ipc_stream_wait: " + why);
    }
    std::vector<uint8_t> image(uint32_t slot) { return slots_[slot]->render_image(streams_[slot]); }   // waits for that slot
    void* image_device_ptr(uint32_t slot) const { return vrc_image_device_ptr(slots_[slot]->handle()); }
    void* stream(uint32_t slot) const { return streams_[slot]; }
    HipRayCaster& raycaster(uint32_t slot) { return *slots_[slot]; }
    // counters of the frames rendered on `slot` since the last reset (waits for that slot)
    vrc_frame_stats stats(uint32_t slot, bool reset = false) { return slots_[slot]->stats(reset, streams_[slot]); }

private:
    int device_;
    uint32_t spp_;
    uint64_t n_ = 0;
    std::vector<std::unique_ptr<HipRayCaster>> slots_;
    std::vector<void*> streams_;
};

// The blend / upscale chain main.cpp runs on render_image after every frame (main.cpp:160-182: render_tex,
// denoised_tex, final_sprite), device resident.  present() replaces lines 160-181; the window image stays in HBM
// (window_device_ptr) or is copied out (window()).
class HipPresenter {
public:
    HipPresenter(uint32_t render_width, uint32_t render_height, uint32_t win_width, uint32_t win_height, int device = 0)
        : out_w_(win_width), out_h_(win_height)
    {
        check(vrc_presenter_create(device, render_width, render_height, win_width, win_height, &p_), "vrc_presenter_create");
    }
    ~HipPresenter() { vrc_presenter_destroy(p_); }
    HipPresenter(const HipPresenter&) = delete;
    HipPresenter& operator=(const HipPresenter&) = delete;

    // old_value_conservation as main.cpp:161 computes it from raycaster.use_samples
    void present(HipRayCaster& raycaster, uint32_t median = 0, void* stream = nullptr)
    {
        const float old_value_conservation = raycaster.use_samples ? 0.0f : 0.1f;
        check(vrc_present(p_, raycaster.handle(), old_value_conservation, median, stream), "vrc_present");
        last_stream_ = stream;
    }
    void* window_device_ptr() const { return vrc_presenter_window_ptr(p_); }
    // Copies the window image out, ordered after the last present(): the read runs on THAT call's stream (frame streams are
    // non-blocking, the NULL stream does not wait for them).
    std::vector<uint8_t> window() const
    {
        std::vector<uint8_t> img((size_t)out_w_ * out_h_ * 4);
        check(vrc_presenter_read(p_, img.data(), nullptr, last_stream_), "vrc_presenter_read");
        return img;
    }

private:
    vrc_presenter* p_ = nullptr;
    uint32_t out_w_, out_h_;
    void* last_stream_ = nullptr;
};

// Camera-path replay files.  The format is the one include/replay.hpp:8-35 reads: whitespace-separated numbers, six per
// tick (time, position x y z, view angles x y); a tick counts only when all six parse, reading ends at the first word that
// is not a number, a missing file gives no ticks (pinned against the reference's reader: tests/golden/replay_cases.json).
// Here a tick is what the renderer consumes -- a time and a camera pose; the lens values, which the file does not carry,
// come from the caller.  The reference has the reader but no caller; this one feeds deterministic multi-frame runs
// (tests/cpp/replay_main.cpp; tools/replay_bench.py speaks the same format).
struct ReplayTick {
    float time = 0.0f;
    CameraState pose;
};

inline std::vector<ReplayTick> readReplay(const std::string& path, float aperture = 0.0f, float focal_length = 1.0f, float fov = 1.0f)
{
    std::vector<ReplayTick> ticks;
    std::ifstream in(path);
    for (;;) {
        float v[6];
        int got = 0;
        while (got < 6 && (in >> v[got])) ++got;
        if (got < 6) break;
        ReplayTick t;
        t.time = v[0];
        t.pose.position = {v[1], v[2], v[3]};
        t.pose.view_angle = {v[4], v[5]};
        t.pose.aperture = aperture; t.pose.focal_length = focal_length; t.pose.fov = fov;
        ticks.push_back(t);
    }
    return ticks;
}

}  // namespace vrc_host

#ifdef VRC_WITH_REFERENCE_HEADERS
// Drop-in Volumetric (needs the reference's include/ and GLM on the include path).
#include "volumetric.hpp"
class HipVolumetric : public Volumetric {
public:
    HipVolumetric(const vrc_lnode* lnodes, uint64_t n_nodes, uint32_t depth, int device = 0) : impl_(lnodes, n_nodes, depth, device)
    {
        cell_.type = Cell::Solid; cell_.texture = Cell::Grass;
    }
    HitPoint castRay(const glm::vec3& position, glm::vec3 direction, const float coef, const float bias) const override
    {
        const vrc_host::HitPoint h = impl_.castRay({position.x, position.y, position.z}, {direction.x, direction.y, direction.z}, coef, bias);
        HitPoint out;
        out.position = glm::vec3(h.position.x, h.position.y, h.position.z);
        out.normal = glm::vec3(h.normal.x, h.normal.y, h.normal.z);
        out.voxel_coord = glm::vec2(h.voxel_coord.x, h.voxel_coord.y);
        out.cell = h.cell ? &cell_ : nullptr;
        out.distance = h.distance;
        out.complexity = h.complexity;
        return out;
    }
    void setCell(Cell::Type, Cell::Texture, uint32_t, uint32_t, uint32_t) override {}
    const vrc_host::HipLSVO& impl() const { return impl_; }
private:
    vrc_host::HipLSVO impl_;
    Cell cell_;
};
#endif
