"""Every configuration BASELINE.json lists, exercised through the C ABI on the GPU at its FULL size and checked against
the oracle (bit-exact; full frame where the oracle finishes in seconds, a strided row band plus size-independent
properties where it does not).  Config 1 (128^3 dense Grid3D, 640x360, primaries) is tests/test_gpu_grid.py::
test_grid_128_config1; the others are here, on the default frame kernel (k_render_sync).

Reference anchors: src/main.cpp:115-121 (autofocus -> focal_length), :131-152 (ray set), include/raycaster.hpp:118-207
(shadow + GI fan-out), include/camera_controller.hpp:34-60 (thin-lens DOF, getClosestPoint)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def oracle_cam(cam):
    return O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)


def make_renderer(vrc, svo, W, H, depth, use_gi):
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples = bool(use_gi), True, 1
    return rc


def oracle_band(nodes, depth, textures, cam, W, H, spp, use_gi, shard_index, shard_count, frame0=0):
    """The oracle's accumulators for the rows of 8-row blocks b with b % shard_count == shard_index."""
    import cpuvoxelraycaster_amd as vrc
    top, side = textures
    acc = None
    rays = steps = 0
    for s in range(spp):
        p = O.make_params(W, H, vrc.reference_light(depth), use_gi=use_gi, use_samples=1, shadow_samples=1,
                          frame_index=frame0 + s, row_block=8, shard_index=shard_index, shard_count=shard_count)
        _, acc, _, st = O.render_frame(nodes, depth, top, side, oracle_cam(cam), p, accum=acc, threads=8)
        rays += st.rays
        steps += st.sum_complexity
    rows = [y for y in range(H) if (y // 8) % shard_count == shard_index]
    return acc, rows, rays, steps


def test_config2_256_lsvo_720p_primary_plus_shadow(built, heights, textures):
    """BASELINE config 2: "256^3 LSVO, 1280x720, primary + 1 shadow ray, 1xMI355X" -- the whole frame against the
    oracle: every primary HitPoint bitwise (voxel, normal, uv, distance, complexity), accumulators, resolved image,
    ray and loop-iteration counts; and the same frame through the 0.4 / 0.6 temporal blend sink."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    depth, W, H = 8, 1280, 720
    nodes = vrc.build_terrain_lsvo(heights, depth)
    svo = vrc.LSVO(nodes, depth, textures=textures)
    cam = vrc.reference_camera(depth, pitch=-0.5)
    light = vrc.reference_light(depth)
    top, side = textures
    rc = make_renderer(vrc, svo, W, H, depth, use_gi=0)
    prim_t = torch.zeros(W * H * 48, dtype=torch.uint8, device="cuda")
    rc.setPrimaryCapture(prim_t.data_ptr())
    rc.renderFrame(cam, spp=1)
    p = O.make_params(W, H, light, use_gi=0, use_samples=1, shadow_samples=1)
    _, oacc, oprim, ost = O.render_frame(nodes, depth, top, side, oracle_cam(cam), p, want_prim=True, threads=8)
    st = rc.stats()
    assert (st.rays, st.sum_complexity, st.pixels) == (ost.rays, ost.sum_complexity, W * H)
    assert W * H < st.rays <= 2 * W * H                       # primary + (on a hit) one shadow ray
    assert np.array_equal(rc.readAccum(), oacc)
    rc.samples_to_image()
    assert np.array_equal(rc.readImage(), O.samples_to_image(oacc))
    torch.cuda.synchronize()
    got = np.frombuffer(prim_t.cpu().numpy().tobytes(), dtype=vrc.HIT_DTYPE)
    assert got.tobytes() == oprim.tobytes()
    assert st.primary_hits == int((oprim["hit"] != 0).sum()) > W * H // 2
    # non-sample sink (raycaster.hpp:77-85) on the same frame
    rc.setPrimaryCapture(None)
    rc.use_samples = False
    rc.clearImage()
    rc.frame_index = 0
    rc.renderFrame(cam, spp=1)
    p = O.make_params(W, H, light, use_gi=0, use_samples=0, shadow_samples=1)
    oimg, _, _, _ = O.render_frame(nodes, depth, top, side, oracle_cam(cam), p, threads=8)
    assert np.array_equal(rc.readImage(), oimg)


def test_config4_512_lsvo_1080p_gi_dof_8_shards(built, heights, textures):
    """BASELINE config 4: "512^3 LSVO, 1920x1080, 4spp GI + DOF, pixel-tile shard across 8xMI355X with RCCL gather".
    focal_length comes from the autofocus ray exactly as main.cpp:115-121 does (Camera::getClosestPoint through
    vrc_cast_ray, checked against the oracle's ray); aperture 0.5.  The frame is rendered as 8 row-block shards on
    one GPU (what the 8 ranks do), each resolved with vrc_resolve_shard into its all-gather slot, the concatenation
    unpacked with vrc_unpack_shards: equal to the unsharded frame, counters add up, and a 1/16 strided band of it
    equals the oracle."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp, RB, N = 9, 1920, 1080, 4, 8, 8
    S = np.float32(1 << depth)
    nodes = vrc.build_terrain_lsvo(heights, depth)
    svo = vrc.LSVO(nodes, depth, textures=textures)
    cam = vrc.reference_camera(depth, pitch=-0.5, aperture=0.5)
    # main.cpp:115-121
    org = (np.array(cam.position[:], np.float32) * (np.float32(1.0) / S) + np.float32(1.0)).astype(np.float32)
    camera_vec = np.array([cam.rot[2], cam.rot[5], cam.rot[8]], np.float32)     # viewToWorld((0, 0, 1))
    hit = svo.castRay(org, camera_vec)
    ohit = O.cast_rays(nodes, depth, org[None], camera_vec[None])[0]
    assert hit.tobytes() == ohit.tobytes()
    cam.focal_length = float(np.float32(hit["distance"]) * S) if hit["hit"] else 100.0
    assert hit["hit"] and 10.0 < cam.focal_length < 512.0

    full = make_renderer(vrc, svo, W, H, depth, use_gi=1)
    full.renderFrame(cam, spp=spp)
    acc_full = full.readAccum()
    full.samples_to_image()
    ref = full.readImage()
    st_full = full.stats()
    assert (acc_full[..., 3] == spp).all()

    L = vrc.capi.load()
    nbytes = L.vrc_shard_bytes(W, H, RB, N)
    gathered = torch.zeros(N * nbytes, dtype=torch.uint8, device="cuda")
    part = make_renderer(vrc, svo, W, H, depth, use_gi=1)
    rays = steps = pixels = 0
    for k in range(N):
        part.frame_index = 0
        part.renderFrame(cam, spp=spp, row_block=RB, shard_index=k, shard_count=N)
        part.resolveShard(RB, k, N, gathered.data_ptr() + k * nbytes, reset=True)
        st = part.stats(reset=True)
        rays, steps, pixels = rays + st.rays, steps + st.sum_complexity, pixels + st.pixels
    out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(gathered.data_ptr()), W, H, RB, N, vrc.capi.ptr(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(H, W, 4), ref)
    assert (rays, steps, pixels) == (st_full.rays, st_full.sum_complexity, W * H * spp)
    assert not part.readAccum().any()                          # every shard's rows were reset by its resolve pass

    oacc, rows, _, _ = oracle_band(nodes, depth, textures, cam, W, H, spp, 1, shard_index=5, shard_count=16)
    assert np.array_equal(acc_full[rows], oacc[rows])
    # the DOF jitter is live: the same band with aperture 0 differs
    pin = vrc.reference_camera(depth, pitch=-0.5, aperture=0.0, focal_length=cam.focal_length)
    oacc0, _, _, _ = oracle_band(nodes, depth, textures, pin, W, H, 1, 1, shard_index=5, shard_count=16)
    oacc1, _, _, _ = oracle_band(nodes, depth, textures, cam, W, H, 1, 1, shard_index=5, shard_count=16)
    assert not np.array_equal(oacc0[rows], oacc1[rows])


def test_config3_512_lsvo_1080p_4spp_gi_band(built, heights, textures):
    """BASELINE config 3: "512^3 LSVO (FastNoise terrain), 1920x1080, 4spp GI ..., 1xMI355X" -- the bench workload:
    the scene generated AND built on the GPU (vrc_scene_build_fastnoise_terrain) equals the host-built array, the
    whole frame's counters are reproducible, and a 1/16 strided band of the 4-sample accumulators equals the oracle
    (tests/test_gpu_frame.py::test_full_size_properties holds the determinism / linearity properties)."""
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp = 9, 1920, 1080, 4
    nodes = vrc.build_terrain_lsvo(heights, depth)
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, seed=1337, textures=textures)
    assert svo.downloadNodes().tobytes() == nodes.tobytes()
    cam = vrc.reference_camera(depth, pitch=-0.5)
    rc = make_renderer(vrc, svo, W, H, depth, use_gi=1)
    rc.renderFrame(cam, spp=spp)
    acc = rc.readAccum()
    st = rc.stats()
    assert st.pixels == W * H * spp
    oacc, rows, _, _ = oracle_band(nodes, depth, textures, cam, W, H, spp, 1, shard_index=11, shard_count=16)
    assert np.array_equal(acc[rows], oacc[rows])


def test_config5_1024_lsvo_4k_16spp_gi(built, heights, textures):
    """BASELINE config 5: "1024^3 LSVO, 3840x2160, 16spp GI ..., 8xMI355X (HBM-resident SVO)" -- 323.8 MiB tree
    (42 436 609 nodes), 8.3 M pixels x 16 samples.  Size-independent properties: determinism (same accumulators and
    the same ray / loop-iteration counts when rendered again), linearity of accumulation (8 + 8 samples == 16), every
    pixel sampled 16 times; and a 1/64 strided row band of the 16-sample accumulators equals the oracle, as do its
    ray and loop-iteration counts (rendered as that shard alone).  ("blurry reflections" do not exist in the reference,
    SURVEY App. C.)"""
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp = 10, 3840, 2160, 16
    nodes = vrc.build_terrain_lsvo(heights, depth)
    assert nodes.shape[0] == 42436609
    svo = vrc.LSVO(nodes, depth, textures=textures)
    cam = vrc.reference_camera(depth, pitch=-0.5)
    rc = make_renderer(vrc, svo, W, H, depth, use_gi=1)
    rc.renderFrame(cam, spp=spp)
    a1 = rc.readAccum()
    st1 = rc.stats(reset=True)
    assert (a1[..., 3] == spp).all()
    assert st1.pixels == W * H * spp and W * H * spp <= st1.rays <= 4 * W * H * spp
    rc.resetSamples()
    rc.frame_index = 0
    rc.renderFrame(cam, spp=8)
    rc.renderFrame(cam, spp=8)
    st2 = rc.stats(reset=True)
    assert (st2.rays, st2.sum_complexity, st2.primary_hits) == (st1.rays, st1.sum_complexity, st1.primary_hits)
    assert np.array_equal(rc.readAccum(), a1)
    # 1/64 band against the oracle, and that shard rendered alone for the counters
    K, NSH = 37, 64
    oacc, rows, orays, osteps = oracle_band(nodes, depth, textures, cam, W, H, spp, 1, shard_index=K, shard_count=NSH)
    assert len(rows) >= 24
    assert np.array_equal(a1[rows], oacc[rows])
    rc.resetSamples()
    rc.frame_index = 0
    rc.renderFrame(cam, spp=spp, row_block=8, shard_index=K, shard_count=NSH)
    st3 = rc.stats()
    assert (st3.rays, st3.sum_complexity) == (orays, osteps)
    band = rc.readAccum()
    assert np.array_equal(band[rows], oacc[rows])
    other = np.ones(H, bool)
    other[rows] = False
    assert not band[other].any()
