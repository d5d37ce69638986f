"""The stage-synchronous frame kernel exists in builds for 6 waves per SIMD and, the lens kernel, 7 (vrc_internal.h: sync_waves_for).
Every build renders the same frame bit for bit -- against the oracle on a small frame with the build forced through
vrc_renderer_set_tuning(blocks_per_cu), and build against build at BASELINE size where the library picks by the kind of
launch; vrc_renderer_last_kernel names what ran."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    nodes = vrc.build_terrain_lsvo(heights, 7)
    return nodes, vrc.LSVO(nodes, 7, textures=textures)


def make(vrc, svo, depth, W, H, blocks=0, chunk=None, use_gi=True, gi_bounces=1):
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples, rc.gi_bounces = use_gi, True, 1, gi_bounces
    if chunk is not None:
        rc.setSampleChunk(chunk)
    if blocks:
        rc.setTuning(blocks_per_cu=blocks)
    return rc


@pytest.mark.parametrize("aperture", [0.0, 0.6])
def test_forced_builds_equal_the_oracle(scene, textures, aperture):
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    top, side = textures
    W, H, spp = 161, 93, 3
    cam = vrc.reference_camera(7, pitch=-0.5, aperture=aperture, focal_length=30.0 if aperture else 1.0)
    light = vrc.reference_light(7)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    acc, rays, steps = None, 0, 0
    for s in range(spp):
        p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=s)
        _, acc, _, st = O.render_frame(nodes, 7, top, side, ocam, p, accum=acc, threads=8)
        rays += st.rays; steps += st.sum_complexity
    names = set()
    for blocks in (3, 6, 7, 8):
        for fused in (False, True):
            rc = make(vrc, svo, 7, W, H, blocks)
            if fused:
                rc.renderFrameResolved(cam, spp=spp)
                assert np.array_equal(rc.readImage(), O.samples_to_image(acc)), (blocks, fused)
            else:
                rc.renderFrame(cam, spp=spp)
                assert np.array_equal(rc.readAccum(), acc), (blocks, fused)
            st = rc.stats()
            assert (st.rays, st.sum_complexity) == (rays, steps), (blocks, fused)
            names.add(rc.lastKernel())
    base = "k_render_sync_pinhole" if aperture == 0.0 else "k_render_sync"
    want = {base, base + "_resolved"}
    if aperture != 0.0:
        want |= {base + "_w7", base + "_resolved_w7"}
    assert names == want


def test_library_choice_at_baseline_size(textures):
    """1920x1080, 512^3: the build the library picks per kind of launch, each equal to the 6-wave build of the same frame"""
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp = 9, 1920, 1080, 4
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=textures)
    pin = vrc.reference_camera(depth, pitch=-0.5)
    dof = vrc.reference_camera(depth, pitch=-0.5, aperture=0.5, focal_length=60.0)
    cases = [  # camera, whole-spp units, GI, spp, bounces, expected kernel
        (pin, False, True, spp, 1, "k_render_sync_pinhole_s4"),     # a launch alone on the chip: four samples abreast (round 4)
        (pin, True, True, spp, 1, "k_render_sync_pinhole_q"),       # whole-spp units of four samples: quadrant walks (round 5)
        (pin, True, False, spp, 1, "k_render_sync_pinhole_q"),
        (pin, False, False, 1, 1, "k_render_sync_pinhole"),
        (dof, False, True, spp, 1, "k_render_sync_s4"),
        (dof, True, True, spp, 1, "k_render_sync_w7"),
        (dof, False, False, spp, 1, "k_render_sync_s4"),
        (pin, True, True, spp, 2, "k_render_sync_pinhole2_q"),
        (pin, False, True, spp, 2, "k_render_sync_pinhole2"),
        (dof, True, True, spp, 2, "k_render_sync2"),
    ]
    for cam, whole, gi, n, bounces, kernel in cases:
        a = make(vrc, svo, depth, W, H, 0, n if whole else None, gi, bounces)
        b = make(vrc, svo, depth, W, H, 6, n if whole else None, gi, bounces)
        b.setLaneSamples(1)                                          # the 6-wave build of the 8 x 8 map,
        b.setQuadWalks(False)                                        # ... with the plain layout of the sample-invariant walks
        a.renderFrame(cam, spp=n); b.renderFrame(cam, spp=n)
        assert a.lastKernel() == kernel, (kernel, a.lastKernel())
        assert b.lastKernel() in ("k_render_sync_pinhole", "k_render_sync", "k_render_sync_pinhole2", "k_render_sync2"), b.lastKernel()
        assert np.array_equal(a.readAccum(), b.readAccum()), kernel
        sa, sb = a.stats(), b.stats()
        assert (sa.rays, sa.sum_complexity, sa.primary_hits) == (sb.rays, sb.sum_complexity, sb.primary_hits), kernel
