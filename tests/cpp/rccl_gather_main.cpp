// The multi-GPU exchange step from a C++ host (north_star: "Host code stays C++ ... RCCL gather of the RGBA8 framebuffer"):
// one process per GPU, each renders its 8-row blocks of every frame through vrc_host::HipFramePipeline (frames in flight,
// whole-spp units, resolve fused into the frame kernel, rows packed into the shard buffer), then ONE RCCL call per frame on
// the frame's own stream meets the shards --
//   all   ncclAllGather: every rank ends up with the [rank][slot] buffer and unpacks the frame,
//   root  grouped ncclSend / ncclRecv: rank 0, the presenting rank, receives every shard straight into its place
// -- and vrc_unpack_shards scatters them into the frame.  Rank 0 checks the last frames against the unsharded render.
//
// Launched N-way by environment (RANK, WORLD_SIZE, LOCAL_RANK), the ncclUniqueId travels through a file:
//   rccl_gather_main <id-file> <all|root> <depth> <width> <height> <spp> <frames> <top.rgb> <side.rgb>
// Built by tests/test_gpu_rccl_cpp.py:  g++ -std=c++14 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -lrccl -lamdhip64 -l:libvrc_hip.so
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

#include "../../cpuvoxelraycaster_amd/host/hip_raycaster.hpp"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)
#define NCCLCHECK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 4; } } while (0)

static int env_int(const char* name, int def) { const char* v = std::getenv(name); return v ? std::atoi(v) : def; }

static std::vector<char> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
    if (argc != 10) { std::fprintf(stderr, "usage: %s <id-file> <all|root> <depth> <W> <H> <spp> <frames> <top.rgb> <side.rgb>\n", argv[0]); return 2; }
    const std::string id_file = argv[1];
    const bool to_root = !std::strcmp(argv[2], "root");
    const uint32_t depth = (uint32_t)std::atoi(argv[3]), W = (uint32_t)std::atoi(argv[4]), H = (uint32_t)std::atoi(argv[5]);
    const uint32_t spp = (uint32_t)std::atoi(argv[6]), frames = (uint32_t)std::atoi(argv[7]);
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
    int n_dev = 0;
    HIPCHECK(hipGetDeviceCount(&n_dev));
    const int device = env_int("LOCAL_RANK", rank) % n_dev;
    HIPCHECK(hipSetDevice(device));

    // ---- communicator: rank 0 makes the id, the others read it from the file
    ncclUniqueId id;
    if (rank == 0) {
        NCCLCHECK(ncclGetUniqueId(&id));
        { std::ofstream f(id_file + ".tmp", std::ios::binary); f.write((const char*)&id, sizeof(id)); }
        std::rename((id_file + ".tmp").c_str(), id_file.c_str());
    } else {
        for (int tries = 0;; ++tries) {
            std::ifstream f(id_file, std::ios::binary);
            if (f && f.read((char*)&id, sizeof(id))) break;
            if (tries > 600) { std::fprintf(stderr, "rank %d: no id file %s\n", rank, id_file.c_str()); return 5; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ncclComm_t comm;
    NCCLCHECK(ncclCommInitRank(&comm, world, id, rank));

    try {
        // ---- scene (replicated) and the frame pipeline of this rank's shard
        const std::unique_ptr<vrc_host::HipLSVO> lsvo = vrc_host::HipLSVO::fromFastNoiseTerrain(1337, depth, device);
        const std::vector<char> top = slurp(argv[8]), side = slurp(argv[9]);
        if (top.size() != 768 || side.size() != 768) { std::fprintf(stderr, "albedo tables must be 768 bytes\n"); return 2; }
        vrc_host::check(vrc_scene_set_textures(lsvo->handle(), (const uint8_t*)top.data(), (const uint8_t*)side.data()), "textures");
        const float size = (float)(1u << depth);
        const vrc_host::Vec3 light = {-200.0f / size + 1.0f, -1000.0f / size + 1.0f, -300.0f / size + 1.0f};   // main.cpp:124-126
        vrc_host::CameraState cam;
        const float clear = 56.0f * (size > 512.0f ? size / 512.0f : 1.0f);                                    // main.cpp:50-53 at 512^3: (256, 200, 256),
        cam.position = {size * 0.5f, size * 0.5f - clear, size * 0.5f};                                        // 56 voxels off the mid-plane
        cam.view_angle = {0.0f, -0.5f};
        const uint32_t F = 3, ROW_BLOCK = 8;
        vrc_host::HipFramePipeline pipe(*lsvo, W, H, spp, F);
        pipe.setLightPosition(light);
        pipe.setUseGI(true);
        pipe.setShard(ROW_BLOCK, (uint32_t)rank, (uint32_t)world);

        const uint64_t shard_bytes = vrc_shard_bytes(W, H, ROW_BLOCK, (uint32_t)world);
        const uint32_t S = F + 2;                               // buffer sets: more than frames in flight
        std::vector<void*> mine(S), gathered(S), full(S);
        for (uint32_t s = 0; s < S; ++s) {
            HIPCHECK(hipMalloc(&mine[s], shard_bytes));
            HIPCHECK(hipMalloc(&gathered[s], shard_bytes * (uint64_t)world));
            HIPCHECK(hipMalloc(&full[s], (uint64_t)W * H * 4));
        }
        const bool composes = !to_root || rank == 0;
        // a buffer set is used again S frames later, by a frame on ANOTHER stream: that frame waits for the set's last user
        std::vector<hipEvent_t> used(S);
        for (uint32_t s = 0; s < S; ++s) HIPCHECK(hipEventCreateWithFlags(&used[s], hipEventDisableTiming));

        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t i = 0; i < frames; ++i) {
            const uint32_t b = i % S;
            if (i >= S) HIPCHECK(hipStreamWaitEvent((hipStream_t)pipe.stream(pipe.nextSlot()), used[b], 0));
            const uint32_t slot = pipe.submit(cam, mine[b]);            // frame i: this rank's rows, resolved and packed
            hipStream_t st = (hipStream_t)pipe.stream(slot);
            if (!to_root) {
                NCCLCHECK(ncclAllGather(mine[b], gathered[b], shard_bytes, ncclUint8, comm, st));
            } else {
                NCCLCHECK(ncclGroupStart());
                if (rank == 0) {
                    for (int k = 1; k < world; ++k)
                        NCCLCHECK(ncclRecv((char*)gathered[b] + (uint64_t)k * shard_bytes, shard_bytes, ncclUint8, k, comm, st));
                } else {
                    NCCLCHECK(ncclSend(mine[b], shard_bytes, ncclUint8, 0, comm, st));
                }
                NCCLCHECK(ncclGroupEnd());
                if (rank == 0) HIPCHECK(hipMemcpyAsync(gathered[b], mine[b], shard_bytes, hipMemcpyDeviceToDevice, st));
            }
            if (composes)
                vrc_host::check(vrc_unpack_shards(gathered[b], W, H, ROW_BLOCK, (uint32_t)world, full[b], st), "vrc_unpack_shards");
            HIPCHECK(hipEventRecord(used[b], st));
        }
        pipe.waitAll();
        HIPCHECK(hipDeviceSynchronize());
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

        // ---- rank 0: the last min(frames, S) frames against the unsharded render of the same samples
        int equal = 1;
        if (rank == 0) {
            vrc_host::HipRayCaster ref(*lsvo, W, H);
            ref.setLightPosition(light);
            ref.use_gi = true; ref.use_samples = true;
            std::vector<uint8_t> got((size_t)W * H * 4);
            const uint32_t first = frames > S ? frames - S : 0;
            for (uint32_t i = first; i < frames; ++i) {
                ref.setFrameIndex(i * spp);
                ref.renderFrameResolved(cam, spp);
                const std::vector<uint8_t> want = ref.render_image();
                HIPCHECK(hipMemcpy(got.data(), full[i % S], got.size(), hipMemcpyDeviceToHost));
                if (got != want) { equal = 0; std::printf("frame %u differs from the unsharded render\n", i); }
            }
        }
        for (uint32_t s = 0; s < S; ++s) { (void)hipFree(mine[s]); (void)hipFree(gathered[s]); (void)hipFree(full[s]); (void)hipEventDestroy(used[s]); }
        if (rank == 0)
            std::printf("{\"host\": \"c++\", \"exchange\": \"%s\", \"world\": %d, \"frames\": %u, \"frames_in_flight\": %u, \"size\": [%u, %u], "
                        "\"spp\": %u, \"ms_per_frame\": %.4f, \"gathered_frames_equal_unsharded\": %s}\n",
                        to_root ? "grouped ncclSend / ncclRecv to rank 0" : "ncclAllGather", world, frames, F, W, H, spp,
                        ms / frames, equal ? "true" : "false");
        NCCLCHECK(ncclCommDestroy(comm));
        return equal ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
        return 6;
    }
}
