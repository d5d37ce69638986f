"""vrc_render_frame_resolved: the progressive frame (render + samples_to_image + pack + resetSamples) as ONE launch -- the
tile's last work unit resolves the tile -- against the two-call form it replaces: same image, same packed shard rows,
same counters, accumulators left at zero; for every chunking (a tile's samples in one unit, split with and without
the half-chunk tail), sharded and unsharded, pinhole and DOF, odd sizes (accumulator lines shared between tiles), frame
after frame (the two queue sets alternate), and the fall-back for kernels / modes that cannot fuse."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    nodes = vrc.build_terrain_lsvo(heights, 7)
    return nodes, vrc.LSVO(nodes, 7, textures=textures)


def make(vrc, svo, W, H, chunk=None):
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(7))
    rc.use_gi = rc.use_samples = True
    rc.shadow_samples = 1
    if chunk is not None:
        rc.setSampleChunk(chunk)
    return rc


@pytest.mark.parametrize("chunk", [0, 1, 2, 3, 5])
@pytest.mark.parametrize("size", [(200, 100), (161, 93), (7, 5)])
def test_fused_equals_two_calls(scene, chunk, size):
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    W, H = size
    L = vrc.capi.load()
    for aperture in (0.0, 0.6):
        cam = vrc.reference_camera(7, pitch=-0.5, aperture=aperture, focal_length=30.0 if aperture else 1.0)
        for (RB, N, k) in [(0, 1, 0), (8, 3, 1), (16, 2, 0)]:
            nbytes = L.vrc_shard_bytes(W, H, RB, N) if RB else W * H * 4
            a, b = make(vrc, svo, W, H, chunk), make(vrc, svo, W, H, chunk)
            da = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            db = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            for frame in range(3):
                a.renderFrame(cam, spp=5, row_block=RB, shard_index=k, shard_count=N)
                a.resolveShard(RB, k, N, da.data_ptr(), reset=True)
                b.renderFrameResolved(cam, spp=5, dst_ptr=db.data_ptr(), row_block=RB, shard_index=k, shard_count=N)
                torch.cuda.synchronize()
                assert np.array_equal(a.readImage(), b.readImage()), (chunk, size, aperture, RB, frame)
                # the two-call form writes zeros into the packed buffer's padding rows, the fused one leaves them alone
                # (both buffers start at zero here)
                assert torch.equal(da, db), (chunk, size, aperture, RB, frame)
                assert not b.readAccum().any()
            sa, sb = a.stats(), b.stats()
            assert (sa.rays, sa.sum_complexity, sa.pixels, sa.primary_hits) == (sb.rays, sb.sum_complexity, sb.pixels, sb.primary_hits)


def test_fused_frame_equals_oracle(scene, textures):
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    top, side = textures
    W, H, spp = 160, 90, 4
    cam = vrc.reference_camera(7, pitch=-0.5)
    rc = make(vrc, svo, W, H)
    light = vrc.reference_light(7)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    for frame in range(2):
        rc.renderFrameResolved(cam, spp=spp)
        acc = None
        for s in range(spp):
            p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=frame * spp + s)
            _, acc, _, _ = O.render_frame(nodes, 7, top, side, ocam, p, accum=acc, threads=8)
        assert np.array_equal(rc.readImage(), O.samples_to_image(acc)), frame


def test_fused_falls_back_where_it_cannot_fuse(scene):
    """without use_samples the call is an error; mixing fused and plain frames (and both lane maps) on one renderer keeps the
    queue sets consistent."""
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    W, H = 96, 56
    cam = vrc.reference_camera(7, pitch=-0.5)
    ref = make(vrc, svo, W, H)
    ref.renderFrameResolved(cam, spp=3)
    want = ref.readImage()
    rc = make(vrc, svo, W, H)
    rc.use_samples = False
    with pytest.raises(vrc.VrcError):
        rc.renderFrameResolved(cam, spp=1)
    # fused, plain + separate resolve, the other lane map, fused again -- same renderer
    rc = make(vrc, svo, W, H)
    for step in range(2):
        rc.frame_index = 0
        rc.renderFrameResolved(cam, spp=3)
        assert np.array_equal(rc.readImage(), want)
        rc.frame_index = 0
        rc.renderFrame(cam, spp=3)
        rc.samples_to_image()
        assert np.array_equal(rc.readImage(), want)
        rc.resetSamples()
        rc.setLaneSamples(4 if step else 1)
        rc.frame_index = 0
        rc.renderFrame(cam, spp=4)
        rc.resolveShard(reset=True)
        rc.frame_index = 0
        rc.renderFrame(cam, spp=3)
        rc.resolveShard(reset=True)
        assert np.array_equal(rc.readImage(), want)
        rc.setLaneSamples(0)


def test_fused_full_size_c3(built, heights, textures):
    """BASELINE config 3 size: fused == two calls on the whole 1920x1080 4 spp frame, twice."""
    import cpuvoxelraycaster_amd as vrc
    svo = vrc.LSVO.fromFastNoiseTerrain(9, textures=textures)
    cam = vrc.reference_camera(9, pitch=-0.5)

    def mk():
        rc = vrc.RayCaster(svo, (1920, 1080))
        rc.setLightPosition(vrc.reference_light(9))
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        return rc
    a, b = mk(), mk()
    for frame in range(2):
        a.renderFrame(cam, spp=4)
        a.resolveShard(reset=True)
        b.renderFrameResolved(cam, spp=4)
        assert np.array_equal(a.readImage(), b.readImage())
    assert not b.readAccum().any()
    sa, sb = a.stats(), b.stats()
    assert (sa.rays, sa.sum_complexity) == (sb.rays, sb.sum_complexity)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("chunk", [0, 2, 4])
def test_image_target_composes_shards_in_place(scene, fused, chunk):
    """vrc_renderer_set_image_target (the direct-peer-write exchange, here inside one process): N renderers each render their
    row blocks of the frame and resolve them straight into renderer 0's framebuffer -- full-frame layout, no pack / unpack --,
    fused into the frame kernel or through vrc_resolve_shard; the composed framebuffer equals the unsharded frame, frame after
    frame, and the writers' own images stay untouched."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene
    W, H, spp, RB, N = 203, 117, 4, 8, 3
    L = vrc.capi.load()
    cam = vrc.reference_camera(7, pitch=-0.5)
    parts = [make(vrc, svo, W, H, chunk) for _ in range(N)]
    ref = make(vrc, svo, W, H, chunk)
    target = L.vrc_image_device_ptr(parts[0]._h)
    for p in parts[1:]:
        p.setImageTarget(target)
    untouched = [p.readImage() for p in parts[1:]]
    for frame in range(3):
        for k, p in enumerate(parts):
            p.frame_index = frame * spp
            if fused:
                p.renderFrameResolved(cam, spp=spp, row_block=RB, shard_index=k, shard_count=N)
            else:
                p.renderFrame(cam, spp=spp, row_block=RB, shard_index=k, shard_count=N)
                p.resolveShard(RB, k, N, None, reset=True)
        ref.frame_index = frame * spp
        ref.renderFrameResolved(cam, spp=spp)
        torch.cuda.synchronize()
        assert np.array_equal(parts[0].readImage(), ref.readImage()), (fused, chunk, frame)
        for p, img in zip(parts[1:], untouched):
            assert np.array_equal(p.readImage(), img)
    for p in parts[1:]:
        p.setImageTarget(None)
    parts[1].frame_index = 0
    parts[1].renderFrameResolved(cam, spp=spp, row_block=RB, shard_index=1, shard_count=N)
    own = parts[1].readImage()
    rows = [y for y in range(H) if (y // RB) % N == 1]
    ref.frame_index = 0
    ref.renderFrameResolved(cam, spp=spp)
    assert np.array_equal(own[rows], ref.readImage()[rows])       # back to its own image
