"""The lane <-> (pixel, sample) map of the stage-synchronous kernel (vrc_renderer_set_lane_samples; round-3 verdict item 2):
8 x 8 pixels per wave with the samples in a loop, or 4 x 4 pixels with four samples abreast.  A free parameter of the launch --
like the reference's static 4 x 4 area grid, main.cpp:140-143 -- so every result must be the same under both: image,
accumulators, packed shard rows, primary-hit capture, counters; against each other over sizes that are no multiple of a tile,
every chunking, sharded and fused, pinhole and lens; and against the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    nodes = vrc.build_terrain_lsvo(heights, 7)
    return nodes, vrc.LSVO(nodes, 7, textures=textures)


def make(vrc, svo, W, H, lane_samples, chunk=None, gi=True):
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(7))
    rc.use_gi, rc.use_samples, rc.shadow_samples = gi, True, 1
    rc.setLaneSamples(lane_samples)
    if chunk is not None:
        rc.setSampleChunk(chunk)
    return rc


@pytest.mark.parametrize("spp,chunk", [(4, None), (4, 4), (8, None), (8, 4), (16, 8), (12, None)])
@pytest.mark.parametrize("size", [(200, 100), (161, 93), (7, 5), (3, 2)])
def test_four_samples_abreast_equals_the_pixel_tile_map(scene, size, spp, chunk):
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    W, H = size
    L = vrc.capi.load()
    for aperture in (0.0, 0.6):
        cam = vrc.reference_camera(7, pitch=-0.5, aperture=aperture, focal_length=30.0 if aperture else 1.0)
        for (RB, N, k, fused) in [(0, 1, 0, False), (0, 1, 0, True), (8, 3, 1, True), (16, 2, 0, False)]:
            a, b = make(vrc, svo, W, H, 1, chunk), make(vrc, svo, W, H, 4, chunk)
            nbytes = L.vrc_shard_bytes(W, H, RB, N) if RB else W * H * 4
            da = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            db = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            for frame in range(2):
                for rc, d in ((a, da), (b, db)):
                    if fused:
                        rc.renderFrameResolved(cam, spp=spp, dst_ptr=d.data_ptr(), row_block=RB, shard_index=k, shard_count=N)
                    else:
                        rc.renderFrame(cam, spp=spp, row_block=RB, shard_index=k, shard_count=N)
                torch.cuda.synchronize()
                what = (size, spp, chunk, aperture, RB, fused, frame)
                if fused:
                    assert np.array_equal(a.readImage(), b.readImage()), what
                    assert torch.equal(da, db), what
                    assert not b.readAccum().any(), what
                else:
                    assert np.array_equal(a.readAccum(), b.readAccum()), what
            if b.lastKernel():                        # (a shard without rows launches nothing)
                assert b.lastKernel().endswith("_s4") and not a.lastKernel().endswith("_s4"), (a.lastKernel(), b.lastKernel())
                assert ("pinhole" in b.lastKernel()) == (aperture == 0.0)
            sa, sb = a.stats(), b.stats()
            assert (sa.rays, sa.sum_complexity, sa.pixels, sa.primary_hits) == (sb.rays, sb.sum_complexity, sb.pixels, sb.primary_hits)


def test_frames_the_map_does_not_apply_to_use_the_pixel_tiles(scene):
    """spp not a multiple of four, the temporal blend of the non-sample mode (its order matters, raycaster.hpp:79-85), checkerboard
    frames, the 2-bounce extension and the measurement switches keep the 8 x 8 map whatever is set"""
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    cam = vrc.reference_camera(7, pitch=-0.5)
    rc = make(vrc, svo, 64, 40, 4)
    rc.renderFrame(cam, spp=5)
    assert not rc.lastKernel().endswith("_s4")
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("_s4")
    rc.gi_bounces = 2
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_s4")
    rc.gi_bounces = 1
    rc.renderFrame(cam, spp=4, checker_parity=1)
    assert not rc.lastKernel().endswith("_s4")
    rc.use_samples = False
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_s4")
    rc.use_samples = True
    rc.setWalkFromRoot(True)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("from_root")
    with pytest.raises(Exception):
        rc.setLaneSamples(3)


@pytest.mark.parametrize("aperture", [0.0, 0.5])
def test_four_samples_abreast_equals_oracle(scene, textures, aperture):
    """frame, counters and every primary HitPoint against the oracle (the capture is written by the lane of sample 0)"""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    top, side = textures
    W, H, spp = 161, 91, 4
    cam = vrc.reference_camera(7, pitch=-0.5, aperture=aperture, focal_length=30.0 if aperture else 1.0)
    rc = make(vrc, svo, W, H, 4)
    light = vrc.reference_light(7)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    prim = torch.zeros(W * H * 48, dtype=torch.uint8, device="cuda")
    rc.setPrimaryCapture(prim.data_ptr())
    rays = steps = 0
    for frame in range(2):
        rc.renderFrameResolved(cam, spp=spp)
        torch.cuda.synchronize()
        acc = None
        for s in range(spp):
            p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=frame * spp + s)
            _, acc, hits, st = O.render_frame(nodes, 7, top, side, ocam, p, accum=acc, threads=8, want_prim=(s == 0))
            rays += st.rays
            steps += st.sum_complexity
            if s == 0:
                first = hits
        assert np.array_equal(rc.readImage(), O.samples_to_image(acc)), frame
        got = np.frombuffer(prim.cpu().numpy().tobytes(), dtype=vrc.HIT_DTYPE)
        assert got.tobytes() == first.tobytes(), frame
    rc.setPrimaryCapture(None)
    st = rc.stats()
    assert (st.rays, st.sum_complexity) == (rays, steps)
    assert rc.lastKernel().endswith("_s4")
