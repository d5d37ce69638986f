"""Quadrant walks of the pinhole kernel (render_sync_body's QUAD, round 5): the sample-invariant primary and shadow walks laid out
as 4 x 4 pixels x 4 samples instead of 8 x 8 pixels x 1 sample.  Every ray the reference casts (raycaster.hpp:131,153: once per
sample) is still walked by its own lane, so every result must equal the plain kernel's bit for bit -- image, packed shard rows,
accumulators, all counters -- and the oracle's."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
D = 8


@pytest.fixture(scope="module")
def scene(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    nodes = vrc.build_terrain_lsvo(heights, D)      # (the quadrant walks park a walk's final state in stack rows 3..7: depth >= 8)
    return nodes, vrc.LSVO(nodes, D, textures=textures)


def make(vrc, svo, W, H, chunk, shadow_samples=1):
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(D))
    rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, shadow_samples
    if chunk:
        rc.setSampleChunk(chunk)
    rc.setLaneSamples(1)
    return rc


def render(rc, quad, fn):
    rc.setQuadWalks(quad)
    fn(rc)


@pytest.mark.parametrize("spp,chunk,shadows", [(4, 4, 1), (8, 8, 1), (8, 4, 1), (16, 16, 1), (4, 4, 2), (12, 4, 1), (16, 8, 1), (8, 8, 2), (4, 4, 4), (8, 8, 0)])
@pytest.mark.parametrize("size", [(200, 100), (161, 93), (17, 9), (7, 5)])
def test_quadrant_walks_equal_the_plain_kernel(scene, size, spp, chunk, shadows):
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    W, H = size
    L = vrc.capi.load()
    for pitch in (-0.5, 0.3):
        cam = vrc.reference_camera(D, pitch=pitch)
        for (RB, N, k, fused) in [(0, 1, 0, True), (0, 1, 0, False), (8, 3, 1, True), (16, 2, 0, False)]:
            a, b = make(vrc, svo, W, H, chunk, shadows), make(vrc, svo, W, H, chunk, shadows)
            nbytes = L.vrc_shard_bytes(W, H, RB, N) if RB else W * H * 4
            da = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            db = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            for frame in range(2):
                for rc, d, quad in ((a, da, False), (b, db, True)):
                    rc.frame_index = frame * spp

                    def go(rc):
                        if fused:
                            rc.renderFrameResolved(cam, spp=spp, dst_ptr=d.data_ptr(), row_block=RB, shard_index=k, shard_count=N)
                        else:
                            rc.renderFrame(cam, spp=spp, row_block=RB, shard_index=k, shard_count=N)
                    render(rc, quad, go)
                torch.cuda.synchronize()
                what = (size, spp, chunk, shadows, pitch, RB, fused, frame)
                if fused:
                    assert np.array_equal(a.readImage(), b.readImage()), what
                    assert torch.equal(da, db), what
                    assert not b.readAccum().any(), what
                else:
                    assert np.array_equal(a.readAccum(), b.readAccum()), what
            if b.lastKernel():
                assert b.lastKernel().endswith("_q") and not a.lastKernel().endswith("_q"), (a.lastKernel(), b.lastKernel())
            sa, sb = a.stats(), b.stats()
            assert (sa.rays, sa.sum_complexity, sa.pixels, sa.primary_hits, sa.iterations_not_executed) == \
                   (sb.rays, sb.sum_complexity, sb.pixels, sb.primary_hits, sb.iterations_not_executed), what


_SCENES = {}


def scene_at(vrc, heights, textures, depth):
    if depth not in _SCENES:
        nodes = vrc.build_terrain_lsvo(heights, depth)
        _SCENES[depth] = (nodes, vrc.LSVO(nodes, depth, textures=textures))
    return _SCENES[depth]


# (spp, chunk, shadow_samples, gi_bounces, depth): 4 / 8 / 16 lanes per pixel (quad_lanes_lg), the 16-lane shadow walks of two
# shadow samples, no shadow ray at all, and the 2-bounce _q kernels -- each DIRECTLY against orc_render_frame (round-5 verdict,
# Next 1b; before, only (4, 4, 1, 1, 8) met the oracle and the others met the plain kernel)
@pytest.mark.parametrize("spp,chunk,shadows,bounces,depth", [(4, 4, 1, 1, 8), (8, 8, 1, 1, 8), (16, 16, 1, 1, 9), (4, 4, 0, 1, 8),
                                                              (8, 8, 2, 1, 9), (4, 4, 1, 2, 8), (16, 8, 4, 1, 8)])
def test_quadrant_walks_equal_oracle(built, heights, textures, spp, chunk, shadows, bounces, depth):
    """raycaster.hpp:131,147-158: one primary cast and `shadow_sample` shadow casts per sample, every one of them walked by a
    lane of its own in the quadrant layout: image, rays and loop iterations of three consecutive frames vs the oracle"""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene_at(vrc, heights, textures, depth)
    top, side = textures
    W, H = (161, 91) if spp <= 8 else (97, 61)
    cam = vrc.reference_camera(depth, pitch=-0.5)
    rc = vrc.RayCaster(svo, (W, H))
    light = vrc.reference_light(depth)
    rc.setLightPosition(light)
    rc.use_gi, rc.use_samples, rc.shadow_samples, rc.gi_bounces = True, True, shadows, bounces
    rc.setSampleChunk(chunk)
    rc.setLaneSamples(1)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    rays = steps = 0
    for frame in range(3):
        rc.frame_index = frame * spp
        render(rc, True, lambda r: r.renderFrameResolved(cam, spp=spp))
        torch.cuda.synchronize()
        acc = None
        for s in range(spp):
            p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=shadows, frame_index=frame * spp + s, gi_bounces=bounces)
            _, acc, _, st = O.render_frame(nodes, depth, top, side, ocam, p, accum=acc, threads=8)
            rays += st.rays
            steps += st.sum_complexity
        assert np.array_equal(rc.readImage(), O.samples_to_image(acc)), frame
    st = rc.stats()
    assert (st.rays, st.sum_complexity) == (rays, steps)
    assert rc.lastKernel() == ("k_render_sync_pinhole_resolved_q" if bounces == 1 else "k_render_sync_pinhole2_resolved_q")


def test_two_bounce_extension_and_fallbacks(scene):
    """the 2-bounce kernels have the layout too; launches it does not apply to run the plain kernels whatever is set: units
    that are not a multiple of four samples, a lens, invariant-ray reuse, the primary-hit capture, the samples-abreast map, the
    from-root switch, frames without samples"""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    cam = vrc.reference_camera(D, pitch=-0.5)
    W, H, spp = 72, 40, 4
    a, b = make(vrc, svo, W, H, spp), make(vrc, svo, W, H, spp)
    a.gi_bounces = b.gi_bounces = 2
    render(a, False, lambda r: r.renderFrameResolved(cam, spp=spp))
    render(b, True, lambda r: r.renderFrameResolved(cam, spp=spp))
    torch.cuda.synchronize()
    assert np.array_equal(a.readImage(), b.readImage())
    sa, sb = a.stats(), b.stats()
    assert (sa.rays, sa.sum_complexity) == (sb.rays, sb.sum_complexity)
    assert b.lastKernel() == "k_render_sync_pinhole2_resolved_q" and a.lastKernel() == "k_render_sync_pinhole2_resolved"
    rc = make(vrc, svo, W, H, spp)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel() == "k_render_sync_pinhole_q"            # on by default
    rc.renderFrame(cam, spp=6)
    assert not rc.lastKernel().endswith("_q")
    rc.setSampleChunk(2)
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_q")
    rc.setSampleChunk(4)
    rc.renderFrame(vrc.reference_camera(D, pitch=-0.5, aperture=0.5, focal_length=30.0), spp=4)
    assert not rc.lastKernel().endswith("_q")
    rc.setInvariantRayReuse(True)
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_q")
    rc.setInvariantRayReuse(False)
    prim = torch.zeros(W * H * 48, dtype=torch.uint8, device="cuda")
    rc.setPrimaryCapture(prim.data_ptr())
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_q")
    rc.setPrimaryCapture(None)
    rc.setLaneSamples(4)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("_s4")
    rc.setLaneSamples(1)
    rc.setWalkFromRoot(True)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("from_root")
    rc.setWalkFromRoot(False)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel() == "k_render_sync_pinhole_q"
    assert vrc.capi.load().vrc_renderer_set_quad_walks(rc._h, 2) == -1 and vrc.capi.load().vrc_renderer_set_quad_walks(None, 1) == -1
