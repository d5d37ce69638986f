// Direct peer writes from a C++ host (SURVEY 8e "or direct peer writes into the root's framebuffer"; include/vrc.h: vrc_ipc_*):
// WORLD processes (forked here, before anything touches the GPU) share GPU 0; each renders its 8-row blocks of every frame
// through vrc_host::HipFramePipeline (three frames in flight, whole-spp units, resolve fused into the frame kernel); the
// processes of rank > 0 resolve straight into rank 0's framebuffers, which rank 0 exported (vrc_ipc_export_image) and they
// opened (vrc_ipc_open_image, HipRayCaster::setImageTarget).  Ordering: frame flags in shared memory, written and waited for in
// stream order (vrc_stream_write_flag / vrc_stream_wait_flag).  Rank 0 copies every completed frame out on the frame's stream
// and compares all of them with the unsharded render.  The same code runs with one process per GPU of a node.
//   peer_write_main <world> <depth> <width> <height> <spp> <frames> <top.rgb> <side.rgb>
#include <hip/hip_runtime_api.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

#include "../../cpuvoxelraycaster_amd/host/hip_raycaster.hpp"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

static std::vector<char> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static bool write_all(int fd, const void* p, size_t n) { return write(fd, p, n) == (ssize_t)n; }
static bool read_all(int fd, void* p, size_t n)
{
    char* c = (char*)p;
    while (n) { const ssize_t r = read(fd, c, n); if (r <= 0) return false; c += r; n -= (size_t)r; }
    return true;
}

int main(int argc, char** argv)
{
    if (argc != 9) { std::fprintf(stderr, "usage: %s <world> <depth> <W> <H> <spp> <frames> <top.rgb> <side.rgb>\n", argv[0]); return 2; }
    const int world = std::atoi(argv[1]);
    const uint32_t depth = (uint32_t)std::atoi(argv[2]), W = (uint32_t)std::atoi(argv[3]), H = (uint32_t)std::atoi(argv[4]);
    const uint32_t spp = (uint32_t)std::atoi(argv[5]), frames = (uint32_t)std::atoi(argv[6]);
    const uint32_t F = 3, ROW_BLOCK = 8;
    if (world < 1 || world > 8) return 2;
    const int die_rank = std::getenv("PW_DIE_RANK") ? std::atoi(std::getenv("PW_DIE_RANK")) : -1;
    const int die_frame = std::getenv("PW_DIE_AT_FRAME") ? std::atoi(std::getenv("PW_DIE_AT_FRAME")) : -1;

    // ---- ranks: fork BEFORE the first HIP call; one pipe from rank 0 to every other rank carries the set-up message
    int rank = 0;
    std::vector<int> to_child;
    int from_parent = -1;
    char flags_name[64];
    std::snprintf(flags_name, sizeof(flags_name), "/vrc_cpp_peer_%d", (int)getpid());
    std::vector<pid_t> kids;
    for (int k = 1; k < world; ++k) {
        int fd[2];
        if (pipe(fd) != 0) return 2;
        const pid_t pid = fork();
        if (pid < 0) return 2;
        if (pid == 0) { rank = k; from_parent = fd[0]; close(fd[1]); for (int o : to_child) close(o); to_child.clear(); break; }
        kids.push_back(pid);
        to_child.push_back(fd[1]);
        close(fd[0]);
    }

    int rc = 0;
    try {
        const std::unique_ptr<vrc_host::HipLSVO> lsvo = vrc_host::HipLSVO::fromFastNoiseTerrain(1337, depth, 0);
        const std::vector<char> top = slurp(argv[7]), side = slurp(argv[8]);
        if (top.size() != 768 || side.size() != 768) { std::fprintf(stderr, "albedo tables must be 768 bytes\n"); return 2; }
        vrc_host::check(vrc_scene_set_textures(lsvo->handle(), (const uint8_t*)top.data(), (const uint8_t*)side.data()), "textures");
        const float size = (float)(1u << depth);
        const vrc_host::Vec3 light = {-200.0f / size + 1.0f, -1000.0f / size + 1.0f, -300.0f / size + 1.0f};   // main.cpp:124-126
        vrc_host::CameraState cam;
        const float clear = 56.0f * (size > 512.0f ? size / 512.0f : 1.0f);
        cam.position = {size * 0.5f, size * 0.5f - clear, size * 0.5f};
        cam.view_angle = {0.0f, -0.5f};
        vrc_host::HipFramePipeline pipe(*lsvo, W, H, spp, F);
        pipe.setLightPosition(light);
        pipe.setUseGI(true);
        pipe.setShard(ROW_BLOCK, (uint32_t)rank, (uint32_t)world);

        // ---- set-up: rank 0 creates the flags and exports its F framebuffers; the others open both
        const uint32_t n_flags = F * (uint32_t)world + F + 1;   // written[slot][rank], then consumed[slot], then "everyone may leave"
        const uint32_t all_done = n_flags - 1;
        auto written = [&](uint32_t slot, int k) { return slot * (uint32_t)world + (uint32_t)k; };
        auto consumed = [&](uint32_t slot) { return F * (uint32_t)world + slot; };
        vrc_ipc_flags* flags = nullptr;
        std::vector<void*> targets;
        if (rank == 0) {
            vrc_host::check(vrc_ipc_flags_open(flags_name, n_flags, 0, 1, &flags), "vrc_ipc_flags_open (create)");
            std::vector<vrc_ipc_handle> handles(F);
            for (uint32_t s = 0; s < F; ++s) handles[s] = pipe.raycaster(s).exportImage();
            for (int fd : to_child)
                if (!write_all(fd, flags_name, sizeof(flags_name)) || !write_all(fd, handles.data(), F * sizeof(vrc_ipc_handle))) return 5;
        } else {
            std::vector<vrc_ipc_handle> handles(F);
            if (!read_all(from_parent, flags_name, sizeof(flags_name)) || !read_all(from_parent, handles.data(), F * sizeof(vrc_ipc_handle))) return 5;
            vrc_host::check(vrc_ipc_flags_open(flags_name, n_flags, 0, 0, &flags), "vrc_ipc_flags_open");
            for (uint32_t s = 0; s < F; ++s) {
                void* t = nullptr;
                vrc_host::check(vrc_ipc_open_image(0, &handles[s], &t), "vrc_ipc_open_image");
                targets.push_back(t);
                pipe.raycaster(s).setImageTarget(t);
            }
        }

        std::vector<void*> keep;                                // rank 0: a copy of every completed frame
        if (rank == 0) { keep.resize(frames); for (uint32_t i = 0; i < frames; ++i) HIPCHECK(hipMalloc(&keep[i], (size_t)W * H * 4)); }

        // the watchdog's peers: rank 0 watches its children, a child its parent and the siblings forked before it
        std::vector<int32_t> peer_pids;
        if (rank == 0) for (pid_t k : kids) peer_pids.push_back((int32_t)k);
        else { peer_pids.push_back((int32_t)getppid()); for (pid_t k : kids) peer_pids.push_back((int32_t)k); }
        const auto t0 = std::chrono::steady_clock::now();
        try {
        for (uint32_t i = 0; i < frames; ++i) {
            const uint32_t slot = i % F;
            void* st = pipe.stream(slot);
            if (rank != 0 && i >= F)                            // rank 0 has consumed the frame this framebuffer holds
                vrc_host::check(vrc_stream_wait_flag(flags, consumed(slot), i - F + 1, st), "wait consumed");
            if (die_rank == rank && die_frame == (int)i) _exit(5);      // the failure test: vanish with frames in flight, flags unwritten
            pipe.submit(cam);
            if (rank != 0) {
                vrc_host::check(vrc_stream_write_flag(flags, written(slot, rank), i + 1, st), "write written");
            } else {
                for (int k = 1; k < world; ++k) vrc_host::check(vrc_stream_wait_flag(flags, written(slot, k), i + 1, st), "wait written");
                HIPCHECK(hipMemcpyAsync(keep[i], pipe.image_device_ptr(slot), (size_t)W * H * 4, hipMemcpyDeviceToDevice, (hipStream_t)st));
                vrc_host::check(vrc_stream_write_flag(flags, consumed(slot), i + 1, st), "write consumed");
            }
        }
        // the watchdog instead of a blind synchronize
        pipe.waitAllWatched(flags, peer_pids, 30000);
        } catch (const std::exception&) {
            // a peer is gone -- noticed by this rank's watchdog, or by another rank's while this one was still issuing frames
            // (vrc_stream_wait_flag / _write_flag then fail at once): drain what is queued with the release held, then leave
            try { pipe.waitAllWatched(flags, peer_pids, 10000); } catch (const std::exception&) {}
            throw;
        }
        if (rank == 0) vrc_host::check(vrc_ipc_flag_set(flags, all_done, 1), "vrc_ipc_flag_set");   // every frame is complete: the writers may leave
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

        if (rank == 0) {
            int equal = 1;
            vrc_host::HipRayCaster ref(*lsvo, W, H);
            ref.setLightPosition(light);
            ref.use_gi = true; ref.use_samples = true;
            std::vector<uint8_t> got((size_t)W * H * 4);
            for (uint32_t i = 0; i < frames; ++i) {
                ref.setFrameIndex(i * spp);
                ref.renderFrameResolved(cam, spp);
                const std::vector<uint8_t> want = ref.render_image();
                HIPCHECK(hipMemcpy(got.data(), keep[i], got.size(), hipMemcpyDeviceToHost));
                if (got != want) { equal = 0; std::printf("frame %u differs from the unsharded render\n", i); }
            }
            for (int status, k = 0; k < (int)kids.size(); ++k) { waitpid(kids[(size_t)k], &status, 0); if (!WIFEXITED(status) || WEXITSTATUS(status)) equal = 0; }
            std::printf("{\"host\": \"c++\", \"exchange\": \"direct peer writes\", \"world\": %d, \"frames\": %u, \"frames_in_flight\": %u, \"size\": [%u, %u], "
                        "\"spp\": %u, \"ms_per_frame\": %.4f, \"composed_frames_equal_unsharded\": %s}\n", world, frames, F, W, H, spp, ms / frames,
                        equal ? "true" : "false");
            for (void* p : keep) (void)hipFree(p);
            rc = equal ? 0 : 1;
        } else {
            // a writer that has finished does not leave before the presenter has every frame: to the other ranks' watchdogs a
            // process that is gone is a process that died (the flag is also released when the exchange is given up)
            while (vrc_ipc_flag_value(flags, all_done) == 0 && kill(getppid(), 0) == 0) usleep(200);
            for (uint32_t s = 0; s < F; ++s) pipe.raycaster(s).setImageTarget(nullptr);
            for (void* t : targets) vrc_ipc_close_image(0, t);
        }
        vrc_ipc_flags_close(flags);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
        rc = 6;
        if (rank == 0) for (pid_t k : kids) { int status; (void)waitpid(k, &status, 0); }     // their watchdogs end them too
        _exit(rc);                                          // (no destructors: they would synchronise streams of a dead exchange)
    }
    if (rank != 0) _exit(rc);
    return rc;
}
