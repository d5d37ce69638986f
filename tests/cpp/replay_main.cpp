// Camera-path replay through the C++ host adapter: the reference's reader
// (include/replay.hpp:18-33, restated as vrc_host::ReplayElements::loadFromFile) feeds
// HipRayCaster::renderFrame tick by tick -- what a headless main.cpp would do with the replay file.
// Prints one line per frame for the pytest wrapper to compare with the oracle.
//   usage: replay_main <lnodes.bin> <depth> <tex_top.rgb> <tex_side.rgb> <W> <H> <spp> <file.replay>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "../../cpuvoxelraycaster_amd/host/hip_raycaster.hpp"

static std::vector<char> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
    if (argc != 9) return 2;
    const std::vector<char> nodes = slurp(argv[1]);
    const uint32_t depth = (uint32_t)atoi(argv[2]);
    const std::vector<char> top = slurp(argv[3]), side = slurp(argv[4]);
    const uint32_t W = (uint32_t)atoi(argv[5]), H = (uint32_t)atoi(argv[6]), spp = (uint32_t)atoi(argv[7]);
    try {
        const std::list<vrc_host::ReplayElements> ticks = vrc_host::ReplayElements::loadFromFile(argv[8]);
        std::printf("%zu ticks loaded\n", ticks.size());                            // replay.hpp:30
        vrc_host::HipLSVO lsvo((const vrc_lnode*)nodes.data(), nodes.size() / sizeof(vrc_lnode), depth, 0);
        vrc_host::check(vrc_scene_set_textures(lsvo.handle(), (const uint8_t*)top.data(), (const uint8_t*)side.data()), "textures");
        vrc_host::HipRayCaster rc(lsvo, W, H);
        const float size = (float)(1u << depth);
        rc.setLightPosition({-200.0f / size + 1.0f, -1000.0f / size + 1.0f, -300.0f / size + 1.0f});   // main.cpp:124-126 at 512^3
        rc.use_gi = true; rc.use_samples = true;
        unsigned frame = 0;
        for (const vrc_host::ReplayElements& e : ticks) {
            const vrc_host::CameraState cam = e.camera();
            rc.resetSamples();
            rc.renderFrame(cam, -1, spp);
            rc.samples_to_image();
            const std::vector<uint8_t> img = rc.render_image();
            unsigned long long fnv = 1469598103934665603ull;
            for (uint8_t b : img) fnv = (fnv ^ b) * 1099511628211ull;
            const vrc_frame_stats st = rc.stats(true);
            std::printf("frame %u t=%a fnv=%016llx rays=%llu steps=%llu hits=%llu\n", frame++, e.timestamp, fnv,
                        (unsigned long long)st.rays, (unsigned long long)st.sum_complexity, (unsigned long long)st.primary_hits);
        }
    } catch (const std::exception& e) {
        std::printf("error %s\n", e.what());
        return 1;
    }
    return 0;
}
