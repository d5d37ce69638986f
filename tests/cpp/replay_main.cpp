// Camera-path replay through the C++ host adapter: the reference's reader
// (include/replay.hpp:18-33, restated as vrc_host::readReplay) feeds
// HipRayCaster::renderFrame tick by tick -- what a headless main.cpp would do with the replay file.
// Prints one line per frame for the pytest wrapper to compare with the oracle.  With a ninth argument F > 0 the ticks go
// through vrc_host::HipFramePipeline with F frames in flight instead (same lines, frame for frame), and `reuse` as a
// tenth turns the invariant-ray reuse on (same images; fewer walks counted).
//   usage: replay_main <lnodes.bin> <depth> <tex_top.rgb> <tex_side.rgb> <W> <H> <spp> <file.replay> [F [reuse]]
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
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
    if (argc < 9 || argc > 11) return 2;
    const uint32_t in_flight = argc >= 10 ? (uint32_t)atoi(argv[9]) : 0u;
    const bool reuse = argc >= 11 && !std::strcmp(argv[10], "reuse");
    const std::vector<char> nodes = slurp(argv[1]);
    const uint32_t depth = (uint32_t)atoi(argv[2]);
    const std::vector<char> top = slurp(argv[3]), side = slurp(argv[4]);
    const uint32_t W = (uint32_t)atoi(argv[5]), H = (uint32_t)atoi(argv[6]), spp = (uint32_t)atoi(argv[7]);
    try {
        const std::vector<vrc_host::ReplayTick> ticks = vrc_host::readReplay(argv[8]);
        std::printf("%zu ticks loaded\n", ticks.size());                            // replay.hpp:30
        vrc_host::HipLSVO lsvo((const vrc_lnode*)nodes.data(), nodes.size() / sizeof(vrc_lnode), depth, 0);
        vrc_host::check(vrc_scene_set_textures(lsvo.handle(), (const uint8_t*)top.data(), (const uint8_t*)side.data()), "textures");
        const float size = (float)(1u << depth);
        const vrc_host::Vec3 light = {-200.0f / size + 1.0f, -1000.0f / size + 1.0f, -300.0f / size + 1.0f};   // main.cpp:124-126 at 512^3
        const auto t0 = std::chrono::steady_clock::now();
        auto report = [&](unsigned frame, float timestamp, const std::vector<uint8_t>& img, const vrc_frame_stats& st) {
            unsigned long long fnv = 1469598103934665603ull;
            for (uint8_t b : img) fnv = (fnv ^ b) * 1099511628211ull;
            std::printf("frame %u t=%a fnv=%016llx rays=%llu steps=%llu hits=%llu\n", frame, timestamp, fnv,
                        (unsigned long long)st.rays, (unsigned long long)st.sum_complexity, (unsigned long long)st.primary_hits);
        };
        if (in_flight) {
            vrc_host::HipFramePipeline pipe(lsvo, W, H, spp, in_flight);
            pipe.setLightPosition(light);
            pipe.setUseGI(true);
            pipe.setInvariantRayReuse(reuse);
            const std::vector<vrc_host::ReplayTick>& tv = ticks;
            const size_t F = pipe.framesInFlight();
            for (size_t i = 0; i < tv.size() + F; ++i) {
                if (i >= F) {                                        // frame i - F sits in the slot frame i is about to take
                    const uint32_t slot = (uint32_t)((i - F) % F);
                    const std::vector<uint8_t> img = pipe.image(slot);
                    report((unsigned)(i - F), tv[i - F].time, img, pipe.stats(slot, true));
                }
                if (i < tv.size()) pipe.submit(tv[i].pose);
            }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("pipeline frames_in_flight=%zu ms_per_frame=%.4f\n", F, ms / (double)tv.size());
            return 0;
        }
        vrc_host::HipRayCaster rc(lsvo, W, H);
        rc.setLightPosition(light);
        rc.use_gi = true; rc.use_samples = true;
        unsigned frame = 0;
        for (const vrc_host::ReplayTick& e : ticks) {
            const vrc_host::CameraState& cam = e.pose;
            rc.resetSamples();
            rc.renderFrame(cam, -1, spp);
            rc.samples_to_image();
            const std::vector<uint8_t> img = rc.render_image();
            report(frame++, e.time, img, rc.stats(true));
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("sequential ms_per_frame=%.4f\n", ms / (double)(ticks.empty() ? 1 : ticks.size()));
    } catch (const std::exception& e) {
        std::printf("error %s\n", e.what());
        return 1;
    }
    return 0;
}
