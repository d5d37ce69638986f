"""Hand-over of the GI stages' stragglers inside a workgroup (vrc_renderer_set_handover, round 5; the offline bound is
tests/tools/handover_study.py).  A wave leaves the walk loop when at most `cut` of its rays are still walking and one wave of the
workgroup walks the four waves' stragglers together.  A ray's walk does not depend on the lane that runs it -- like the
reference's swarm, whose image does not depend on which worker takes which area (main.cpp:139-154) -- so every result must equal
the plain kernel's bit for bit: image, packed shard rows, accumulators, primary-hit capture and all four counters; for every cut
threshold, sizes that leave waves of a workgroup without a tile, shards, both resolves; and the oracle's."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    nodes = vrc.build_terrain_lsvo(heights, 7)
    return nodes, vrc.LSVO(nodes, 7, textures=textures)


def make(vrc, svo, W, H, spp, cut):
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(7))
    rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
    rc.setSampleChunk(spp)                         # whole-spp work units: what hand-over applies to
    rc.setLaneSamples(1)
    rc.setHandover(cut)
    return rc


@pytest.mark.parametrize("cut", [1, 5, 16])
@pytest.mark.parametrize("size", [(200, 100), (161, 93), (17, 9), (7, 5), (64, 8)])
def test_handover_equals_the_plain_kernel(scene, size, cut):
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    W, H = size
    L = vrc.capi.load()
    for pitch in (-0.5, 0.3):
        cam = vrc.reference_camera(7, pitch=pitch)
        for spp in (4, 3):
            for (RB, N, k, fused) in [(0, 1, 0, True), (0, 1, 0, False), (8, 3, 1, True), (16, 2, 0, False)]:
                a, b = make(vrc, svo, W, H, spp, 0), make(vrc, svo, W, H, spp, cut)
                nbytes = L.vrc_shard_bytes(W, H, RB, N) if RB else W * H * 4
                da = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
                db = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
                for frame in range(2):
                    for rc, d in ((a, da), (b, db)):
                        rc.frame_index = frame * spp
                        if fused:
                            rc.renderFrameResolved(cam, spp=spp, dst_ptr=d.data_ptr(), row_block=RB, shard_index=k, shard_count=N)
                        else:
                            rc.renderFrame(cam, spp=spp, row_block=RB, shard_index=k, shard_count=N)
                    torch.cuda.synchronize()
                    what = (size, cut, pitch, spp, RB, fused, frame)
                    if fused:
                        assert np.array_equal(a.readImage(), b.readImage()), what
                        assert torch.equal(da, db), what
                        assert not b.readAccum().any(), what
                    else:
                        assert np.array_equal(a.readAccum(), b.readAccum()), what
                if b.lastKernel():                    # (a shard without rows launches nothing)
                    assert b.lastKernel().endswith("_ho") and not a.lastKernel().endswith("_ho"), (a.lastKernel(), b.lastKernel())
                sa, sb = a.stats(), b.stats()
                assert (sa.rays, sa.sum_complexity, sa.pixels, sa.primary_hits, sa.iterations_not_executed) == \
                       (sb.rays, sb.sum_complexity, sb.pixels, sb.primary_hits, sb.iterations_not_executed), what


def test_handover_equals_oracle(scene, textures):
    """frame after frame against the oracle: image, rays, loop iterations, and every primary HitPoint"""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    top, side = textures
    W, H, spp = 161, 91, 4
    cam = vrc.reference_camera(7, pitch=-0.5)
    rc = make(vrc, svo, W, H, spp, 16)
    light = vrc.reference_light(7)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    prim = torch.zeros(W * H * 48, dtype=torch.uint8, device="cuda")
    rc.setPrimaryCapture(prim.data_ptr())
    rays = steps = 0
    for frame in range(3):
        rc.frame_index = frame * spp
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
    assert rc.lastKernel().endswith("_ho")


def test_launches_handover_does_not_apply_to_run_the_plain_kernel(scene):
    """units that are not a tile's whole sample set, a lens, the 2-bounce extension, frames without GI or without samples, the
    samples-abreast map and the from-root switch keep the plain kernels whatever is set"""
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    cam = vrc.reference_camera(7, pitch=-0.5)
    rc = make(vrc, svo, 64, 40, 4, 16)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("_ho")
    rc.setSampleChunk(2)
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_ho")
    rc.setSampleChunk(4)
    rc.renderFrame(vrc.reference_camera(7, pitch=-0.5, aperture=0.5, focal_length=30.0), spp=4)
    assert not rc.lastKernel().endswith("_ho")
    rc.gi_bounces = 2
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_ho")
    rc.gi_bounces = 1
    rc.use_gi = False
    rc.renderFrame(cam, spp=4)
    assert not rc.lastKernel().endswith("_ho")
    rc.use_gi = True
    rc.setLaneSamples(4)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("_s4")
    rc.setLaneSamples(1)
    rc.setWalkFromRoot(True)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("from_root")
    rc.setWalkFromRoot(False)
    rc.renderFrame(cam, spp=4)
    assert rc.lastKernel().endswith("_ho")
    with pytest.raises(Exception):
        rc.setHandover(17)
