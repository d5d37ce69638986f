"""N2 on the GPU (main.cpp:160-182): vrc_present -- the device-resident blend / upscale / median chain on
the renderer's framebuffer -- against the oracle's restatement (orc_present), bit for bit: the persistent denoised
image over several frames, the window image at the reference's 960x540 -> 1280x720 (scale 1 / 0.75) and at odd sizes,
with no median, the 3x3 and the 5x5 network.  (Tolerance against a real GL device: +-1 LSB on the two blend steps,
tests/test_present.py; upscale and median are exact.)"""
import numpy as np
import pytest

import oracle_lib as O
from test_present import orc_present

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("median", [0, 3, 5])
def test_present_random_frames(built, median):
    import torch
    import cpuvoxelraycaster_amd as vrc
    rng = np.random.default_rng(11 + median)
    for (W, H, OW, OH) in [(96, 54, 128, 72), (33, 17, 44, 23), (40, 30, 20, 15), (7, 5, 31, 9), (300, 200, 400, 267)]:
        p = vrc.Presenter((W, H), (OW, OH))
        den = np.zeros((H, W, 4), np.uint8)
        for frame, old in enumerate((0.1, 0.1, 0.0, 0.1, 0.35, 1.0)):
            img = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
            img[..., 3] = 255
            if frame == 1:
                img[..., :3] = 255
            t = torch.from_numpy(img).cuda()
            p.presentImage(t.data_ptr(), old, median)
            win, gden = p.read()
            owin = orc_present(img, den, old, median, (OW, OH))
            assert np.array_equal(gden, den), (W, H, frame)
            assert np.array_equal(win, owin), (W, H, frame)
        p.clear()
        assert not p.read()[1].any()


def test_present_on_the_renderers_framebuffer(built, heights, textures):
    """The reference's interactive frame (main.cpp:131-182): 960x540, checkerboard, 0.4 / 0.6 temporal blend in
    renderRay, then the present chain with old_value_conservation = 0.1, window 1280x720 -- four frames, frame kept on
    the device throughout; and sample mode (old = 0) with the 3x3 median."""
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, OW, OH = 9, 960, 540, 1280, 720
    nodes = vrc.build_terrain_lsvo(heights, depth)
    svo = vrc.LSVO(nodes, depth, textures=textures)
    top, side = textures
    light = vrc.reference_light(depth)
    cam = vrc.reference_camera(depth, pitch=-0.5)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(light)
    rc.use_gi, rc.use_samples, rc.shadow_samples = True, False, 0
    pr = vrc.Presenter((W, H), (OW, OH))
    oimg, den = None, np.zeros((H, W, 4), np.uint8)
    for f in range(4):
        parity = f % 2
        rc.renderFrame(cam, spp=1, checker_parity=parity)
        pr.present(rc)                                           # old = 0.1: use_samples is off
        p = O.make_params(W, H, light, use_gi=1, use_samples=0, shadow_samples=0, checker_parity=parity, frame_index=f)
        oimg, _, _, _ = O.render_frame(nodes, depth, top, side, ocam, p, image=oimg, threads=8)
        owin = orc_present(oimg, den, 0.1, 0, (OW, OH))
        win, gden = pr.read()
        assert np.array_equal(gden, den) and np.array_equal(win, owin), f
    assert win[..., :3].any()
    # sample mode: samples_to_image, then the chain without persistence, 3x3 median
    rc.use_samples, rc.shadow_samples = True, 1
    rc.frame_index = 0
    rc.renderFrame(cam, spp=2)
    rc.samples_to_image()
    pr.present(rc, median=3)                                     # old = 0
    acc = None
    for s in range(2):
        p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=s)
        _, acc, _, _ = O.render_frame(nodes, depth, top, side, ocam, p, accum=acc, threads=8)
    owin = orc_present(O.samples_to_image(acc), den, 0.0, 3, (OW, OH))
    win, gden = pr.read()
    assert np.array_equal(gden, den) and np.array_equal(win, owin)
    # argument checks
    with pytest.raises(vrc.VrcError):
        pr.present(rc, median=4)
    with pytest.raises(vrc.VrcError):
        vrc.Presenter((W + 1, H), (OW, OH)).present(rc)


def test_sharded_frame_through_the_present_chain(built, heights, textures):
    """The multi-GPU frame path end to end on one GPU: three row-block shards rendered and resolved into their
    all-gather slots, vrc_unpack_shards composes the frame, vrc_present_image runs the chain on it (5x5 network) --
    equal to the oracle's chain on the oracle's unsharded frame."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, OW, OH, RB, N = 7, 200, 100, 267, 133, 8, 3
    nodes = vrc.build_terrain_lsvo(heights, depth)
    svo = vrc.LSVO(nodes, depth, textures=textures)
    top, side = textures
    light = vrc.reference_light(depth)
    cam = vrc.reference_camera(depth, pitch=-0.5)
    L = vrc.capi.load()
    nbytes = L.vrc_shard_bytes(W, H, RB, N)
    gathered = torch.zeros(N * nbytes, dtype=torch.uint8, device="cuda")
    for k in range(N):
        rc = vrc.RayCaster(svo, (W, H))
        rc.setLightPosition(light)
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        rc.renderFrame(cam, spp=2, row_block=RB, shard_index=k, shard_count=N)
        rc.resolveShard(RB, k, N, gathered.data_ptr() + k * nbytes, reset=True)
    frame = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(gathered.data_ptr()), W, H, RB, N, vrc.capi.ptr(frame.data_ptr()), None))
    pr = vrc.Presenter((W, H), (OW, OH))
    pr.presentImage(frame.data_ptr(), 0.0, median=5)
    win, den = pr.read()
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    acc = None
    for s in range(2):
        p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=s)
        _, acc, _, _ = O.render_frame(nodes, depth, top, side, ocam, p, accum=acc, threads=8)
    oden = np.zeros((H, W, 4), np.uint8)
    owin = orc_present(O.samples_to_image(acc), oden, 0.0, 5, (OW, OH))
    assert np.array_equal(den, oden) and np.array_equal(win, owin)
