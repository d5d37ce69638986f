"""The benchmark's own step, held to the oracle at full size (round-2 verdict, "weak" 5): what bench.py's timed region
does -- BASELINE config 3: 512^3 FastNoise terrain generated and built on the GPU, 1920x1080, 4 spp GI; three renderers on
three streams taking turns, work units of a tile's whole sample set (setSampleChunk(spp)), resolve fused into the frame
kernel (renderFrameResolved) -- for frames 0..5 issued round-robin with no synchronisation in between, and every frame
compared with the oracle: the WHOLE resolved image and the ray / loop-iteration / pixel-sample counters, plus the kernel
symbol that ran (what a rocprofv3 trace of bench.py lists).  The same for config 4 (DOF: the general kernel).

Reference anchors: src/main.cpp:139-158 (the frame: renderRay per pixel-sample, samples_to_image), include/raycaster.hpp:
67-207, include/camera_controller.hpp:34-54 (thin lens)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

FRAMES, IN_FLIGHT = 6, 3


def bench_step_frames(vrc, svo, cam, W, H, spp, depth):
    """frames 0..FRAMES-1 exactly as bench.py's step() issues them at N = 1 (bench.py: make_renderer, step); every frame also
    resolves into a buffer of its own, so that frame i is still there when frame i + 3 has reused its renderer"""
    import torch
    rcs = []
    for _ in range(IN_FLIGHT):
        r = vrc.RayCaster(svo, (W, H))
        r.setLightPosition(vrc.reference_light(depth))
        r.use_gi, r.use_samples, r.shadow_samples = True, True, 1
        r.setSampleChunk(spp)
        rcs.append(r)
    streams = [torch.cuda.Stream() for _ in range(IN_FLIGHT)]
    keep = [torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(FRAMES)]
    for r in rcs:
        r.resetSamples(streams[0].cuda_stream)
        r.stats(reset=True, stream=streams[0].cuda_stream)
    torch.cuda.synchronize()
    for i in range(FRAMES):
        rc, ts = rcs[i % IN_FLIGHT], streams[i % IN_FLIGHT]
        with torch.cuda.stream(ts):
            rc.frame_index = i * spp
            rc.renderFrameResolved(cam, spp=spp, dst_ptr=keep[i].data_ptr(), stream=ts.cuda_stream)
    torch.cuda.synchronize()
    images = [k.cpu().numpy().reshape(H, W, 4) for k in keep]
    stats = [r.stats() for r in rcs]
    kernels = [r.lastKernel() for r in rcs]
    last_images = [r.readImage() for r in rcs]                  # frames 3, 4, 5 as the renderers hold them
    accums_clear = [not r.readAccum().any() for r in rcs]
    return images, stats, kernels, last_images, accums_clear


def oracle_frames(nodes, depth, textures, cam, W, H, spp):
    import cpuvoxelraycaster_amd as vrc
    top, side = textures
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    threads = min(64, os.cpu_count() or 8)
    out = []
    for f in range(FRAMES):
        acc = None
        rays = steps = hits = 0
        for s in range(spp):
            p = O.make_params(W, H, vrc.reference_light(depth), use_gi=1, use_samples=1, shadow_samples=1, frame_index=f * spp + s)
            _, acc, _, st = O.render_frame(nodes, depth, top, side, ocam, p, accum=acc, threads=threads)
            rays, steps, hits = rays + st.rays, steps + st.sum_complexity, hits + st.primary_hits
        out.append((O.samples_to_image(acc), rays, steps, hits))
    return out


def check(vrc, svo, nodes, textures, cam, W, H, spp, depth, kernel):
    images, stats, kernels, last_images, accums_clear = bench_step_frames(vrc, svo, cam, W, H, spp, depth)
    assert kernels == [kernel] * IN_FLIGHT
    ref = oracle_frames(nodes, depth, textures, cam, W, H, spp)
    for f in range(FRAMES):
        assert np.array_equal(images[f], ref[f][0]), f"frame {f}"
    for k in range(IN_FLIGHT):                                 # renderer k rendered frames k and k + 3
        mine = [ref[f] for f in range(FRAMES) if f % IN_FLIGHT == k]
        st = stats[k]
        assert (st.rays, st.sum_complexity) == tuple(sum(m[j] for m in mine) for j in (1, 2)), k
        assert st.pixels == W * H * spp * len(mine)
        # part of those iterations is counted without being executed (secondary rays start below the root, next to the hit)
        assert 0.1 * st.sum_complexity < st.iterations_not_executed < 0.5 * st.sum_complexity
        assert np.array_equal(last_images[k], images[FRAMES - IN_FLIGHT + k])
    assert all(accums_clear)                                   # the fused resolve leaves resetSamples' zeros


def test_bench_step_c3_pinhole(built, heights, textures):
    """bench.py default command (config c3): k_render_sync_pinhole_resolved_q (quadrant walks, round 5)"""
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp = 9, 1920, 1080, 4
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, seed=1337, textures=textures)
    nodes = svo.downloadNodes()
    assert nodes.tobytes() == vrc.build_terrain_lsvo(heights, depth).tobytes()
    cam = vrc.reference_camera(depth, pitch=-0.5)
    check(vrc, svo, nodes, textures, cam, W, H, spp, depth, "k_render_sync_pinhole_resolved_q")


def test_bench_step_c4_dof(built, heights, textures):
    """bench.py --config c4 (aperture 0.5, focal length 60 voxels as bench.py sets it): k_render_sync_resolved_w7"""
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp = 9, 1920, 1080, 4
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, seed=1337, textures=textures)
    nodes = svo.downloadNodes()
    cam = vrc.reference_camera(depth, pitch=-0.5, aperture=0.5, focal_length=60.0 * (1 << depth) / 512.0)
    check(vrc, svo, nodes, textures, cam, W, H, spp, depth, "k_render_sync_resolved_w7")


def test_bench_step_c5_band(built, heights, textures):
    """bench.py --config c5 (BASELINE config 5's shape: 1024^3 LSVO generated and built on the GPU, 3840x2160, 16 spp GI): the
    kernel symbol that bench line times -- k_render_sync_pinhole_resolved_q with whole-spp units of SIXTEEN samples (16 lanes
    per pixel in the quadrant walks), depth 10 -- against the oracle directly (round-5 verdict, Next 1a; the config-5 test of
    tests/test_gpu_configs.py runs the library's automatic lane map, another symbol).  One shard of 64 (row blocks 37, 101,
    ...: 32 rows of 3840 pixels x 16 samples), rendered and resolved the way bench.py's step() does: the packed shard rows and
    the same rows of the renderer's image bit for bit, rays and loop iterations equal to the oracle's for that shard."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp, RB, K, NSH = 10, 3840, 2160, 16, 8, 37, 64
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, seed=1337, textures=textures)
    nodes = svo.downloadNodes()
    assert nodes.shape[0] == 42436609
    cam = vrc.reference_camera(depth, pitch=-0.5)
    light = vrc.reference_light(depth)
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(light)
    rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
    rc.setSampleChunk(spp)
    L = vrc.capi.load()
    nbytes = L.vrc_shard_bytes(W, H, RB, NSH)
    packed = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    top, side = textures
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    rows = [y for y in range(H) if (y // RB) % NSH == K]
    assert len(rows) == 32
    rays = steps = 0
    for frame in range(2):                                       # the second frame finds the accumulators as the fused resolve left them
        rc.frame_index = frame * spp
        rc.renderFrameResolved(cam, spp=spp, dst_ptr=packed.data_ptr(), row_block=RB, shard_index=K, shard_count=NSH)
        torch.cuda.synchronize()
        acc = None
        for s in range(spp):
            p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=frame * spp + s,
                              row_block=RB, shard_index=K, shard_count=NSH)
            _, acc, _, st = O.render_frame(nodes, depth, top, side, ocam, p, accum=acc, threads=min(64, os.cpu_count() or 8))
            rays += st.rays
            steps += st.sum_complexity
        ref = O.samples_to_image(acc)
        img = rc.readImage()
        assert np.array_equal(img[rows], ref[rows]), frame
        got = packed.cpu().numpy().reshape(-1, RB, W, 4)         # [slot][row in block][x]
        want = np.stack([ref[(k * NSH + K) * RB:(k * NSH + K) * RB + RB] for k in range(len(rows) // RB)])
        assert np.array_equal(got[:len(rows) // RB], want), frame
        assert not got[len(rows) // RB:].any()
    st = rc.stats()
    assert (st.rays, st.sum_complexity, st.pixels) == (rays, steps, 2 * len(rows) * W * spp)
    assert rc.lastKernel() == "k_render_sync_pinhole_resolved_q"
    assert not rc.readAccum().any()
