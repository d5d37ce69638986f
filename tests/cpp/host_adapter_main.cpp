// Exercises the C++ host adapter (cpuvoxelraycaster_amd/host/hip_raycaster.hpp) the
// way the reference's main.cpp would use it: HipLSVO::castRay for the autofocus
// ray (main.cpp:115), HipRayCaster::renderFrame for a frame (main.cpp:139-158).
// Prints values the pytest wrapper compares with the oracle.
//   usage: host_adapter_main <lnodes.bin> <depth> <tex_top.rgb> <tex_side.rgb> <W> <H>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../cpuvoxelraycaster_amd/host/hip_raycaster.hpp"

static std::vector<char> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
    if (argc != 7) return 2;
    const std::vector<char> nodes = slurp(argv[1]);
    const uint32_t depth = (uint32_t)atoi(argv[2]);
    const std::vector<char> top = slurp(argv[3]), side = slurp(argv[4]);
    const uint32_t W = (uint32_t)atoi(argv[5]), H = (uint32_t)atoi(argv[6]);
    try {
        vrc_host::HipLSVO lsvo((const vrc_lnode*)nodes.data(), nodes.size() / sizeof(vrc_lnode), depth, 0);
        vrc_host::check(vrc_scene_set_textures(lsvo.handle(), (const uint8_t*)top.data(), (const uint8_t*)side.data()), "textures");
        // Camera::getClosestPoint (camera_controller.hpp:56-60): position * (1/size) + 1, camera_vec
        const float size = (float)(1u << depth);
        vrc_host::CameraState cam;
        cam.position = {size / 2, size / 2 - 56.0f, size / 2};
        cam.view_angle = {0.0f, -0.5f};
        const vrc_camera abi = cam.to_abi();
        const vrc_host::Vec3 origin = {cam.position.x / size + 1.0f, cam.position.y / size + 1.0f, cam.position.z / size + 1.0f};
        const vrc_host::Vec3 camera_vec = {abi.rot[2], abi.rot[5], abi.rot[8]};   // (0,0,1) * rot_mat
        const vrc_host::HitPoint hp = lsvo.castRay(origin, camera_vec, 0.0f, 0.0f);
        std::printf("autofocus hit=%d distance=%a complexity=%u normal=%a,%a,%a\n", hp.cell != nullptr, hp.distance, hp.complexity,
                    hp.normal.x, hp.normal.y, hp.normal.z);
        cam.focal_length = hp.cell ? hp.distance * size : 100.0f;                   // main.cpp:116-121
        vrc_host::HipRayCaster rc(lsvo, W, H);
        rc.setLightPosition({-200.0f / 512.0f + 1.0f, -1000.0f / 512.0f + 1.0f, -300.0f / 512.0f + 1.0f});
        rc.use_gi = true; rc.use_samples = true;
        rc.renderFrame(cam, -1, 2);
        rc.samples_to_image();
        const std::vector<uint8_t> img = rc.render_image();
        unsigned long long sum = 0, fnv = 1469598103934665603ull;
        for (uint8_t b : img) { sum += b; fnv = (fnv ^ b) * 1099511628211ull; }
        const vrc_frame_stats st = rc.stats();
        std::printf("frame sum=%llu fnv=%016llx rays=%llu steps=%llu\n", sum, fnv, (unsigned long long)st.rays,
                    (unsigned long long)st.sum_complexity);
        // the same frame as three row-block shards (what three ranks would render), composed as an all-gather would
        {
            const uint32_t RB = 8, N = 3;
            const uint64_t nbytes = vrc_shard_bytes(W, H, RB, N);
            void *gathered = nullptr, *full = nullptr;
            if (hipMalloc(&gathered, nbytes * N) != hipSuccess || hipMalloc(&full, (size_t)W * H * 4) != hipSuccess) return 3;
            for (uint32_t k = 0; k < N; ++k) {
                vrc_host::HipRayCaster part(lsvo, W, H);
                part.setLightPosition({-200.0f / 512.0f + 1.0f, -1000.0f / 512.0f + 1.0f, -300.0f / 512.0f + 1.0f});
                part.use_gi = true; part.use_samples = true;
                part.setShard(RB, k, N);
                part.renderFrame(cam, -1, 2);
                part.resolveShard((char*)gathered + nbytes * k, true);
            }
            vrc_host::check(vrc_unpack_shards(gathered, W, H, RB, N, full, nullptr), "vrc_unpack_shards");
            std::vector<uint8_t> composed((size_t)W * H * 4);
            if (hipMemcpy(composed.data(), full, composed.size(), hipMemcpyDeviceToHost) != hipSuccess) return 3;
            std::printf("sharded equal=%d\n", composed == img ? 1 : 0);
            // the same three shards, each through ONE launch (renderFrameResolved = render + resolve + reset fused)
            for (uint32_t k = 0; k < N; ++k) {
                vrc_host::HipRayCaster part(lsvo, W, H);
                part.setLightPosition({-200.0f / 512.0f + 1.0f, -1000.0f / 512.0f + 1.0f, -300.0f / 512.0f + 1.0f});
                part.use_gi = true; part.use_samples = true;
                part.setShard(RB, k, N);
                part.renderFrameResolved(cam, 2, (char*)gathered + nbytes * k);
            }
            vrc_host::check(vrc_unpack_shards(gathered, W, H, RB, N, full, nullptr), "vrc_unpack_shards");
            if (hipMemcpy(composed.data(), full, composed.size(), hipMemcpyDeviceToHost) != hipSuccess) return 3;
            std::printf("sharded fused equal=%d\n", composed == img ? 1 : 0);
            (void)hipFree(gathered); (void)hipFree(full);
        }
    } catch (const std::exception& e) {
        std::printf("error %s\n", e.what());
        return 1;
    }
    return 0;
}
