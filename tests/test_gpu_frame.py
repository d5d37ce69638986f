"""R0 + S1..S4 parity: the persistent-threads frame kernel (vrc_render_frame)
against the oracle's restatement of main.cpp:139-152 / raycaster.hpp:67-207 --
image bytes, sample accumulators, primary hits and the ray / step counters must
all be identical (same counter-based RNG on both sides)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["tiles8x8", "auto"])
def frame_kernel(request, built):
    """Every frame test runs with the 8 x 8 pixel map of the frame kernel forced and under the library's choice, which is four
    samples abreast for the sample-mode frames here whose spp is a multiple of four (vrc_set_lane_samples;
    tests/test_gpu_lane_map.py compares the two maps directly)."""
    import cpuvoxelraycaster_amd as vrc
    vrc.capi.check(vrc.capi.load().vrc_set_lane_samples(request.param))
    yield request.param
    vrc.capi.check(vrc.capi.load().vrc_set_lane_samples(0))


@pytest.fixture(scope="module")
def scene(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    out = {}
    for depth in (7, 9):
        nodes = vrc.build_terrain_lsvo(heights, depth)
        out[depth] = (nodes, vrc.LSVO(nodes, depth, textures=textures))
    return out


def oracle_cam(cam):
    return O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)


def run_pair(scene, textures, depth, W, H, pitch, spp=1, frames=1, use_gi=0, use_samples=1, shadow_samples=1,
             gi_bounces=1, aperture=0.0, focal_length=1.0, checker=False, capture=False, tuning=None):
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene[depth]
    top, side = textures
    if tuning:
        vrc.capi.check(vrc.capi.load().vrc_set_tuning(tuning[0]))
    rc = vrc.RayCaster(svo, (W, H))
    light = vrc.reference_light(depth)
    rc.setLightPosition(light)
    rc.use_gi, rc.use_samples, rc.shadow_samples, rc.gi_bounces = bool(use_gi), bool(use_samples), shadow_samples, gi_bounces
    cam = vrc.reference_camera(depth, pitch=pitch, aperture=aperture, focal_length=focal_length)
    prim_t = None
    if capture:
        prim_t = torch.zeros(W * H * 48, dtype=torch.uint8, device="cuda")
        rc.setPrimaryCapture(prim_t.data_ptr())
    ocam = oracle_cam(cam)
    oimg, oacc, oprim = None, None, None
    rays = steps = 0
    fi = 0
    for f in range(frames):
        parity = (f % 2) if checker else -1
        rc.renderFrame(cam, spp=spp, checker_parity=parity)
        for s in range(spp):
            p = O.make_params(W, H, light, use_gi=use_gi, use_samples=use_samples, shadow_samples=shadow_samples,
                              gi_bounces=gi_bounces, checker_parity=parity, frame_index=fi)
            oimg, oacc, pr, st = O.render_frame(nodes, depth, top, side, ocam, p, image=oimg, accum=oacc,
                                                want_prim=(capture and f == 0 and s == 0), threads=8)
            if pr is not None:
                oprim = pr
            rays += st.rays
            steps += st.sum_complexity
            fi += 1
    st = rc.stats()
    if tuning:
        vrc.capi.check(vrc.capi.load().vrc_set_tuning(0))
    assert (st.rays, st.sum_complexity) == (rays, steps)
    if use_samples:
        assert np.array_equal(rc.readAccum(), oacc)
        rc.samples_to_image()
        assert np.array_equal(rc.readImage(), O.samples_to_image(oacc))
    else:
        assert np.array_equal(rc.readImage(), oimg)
    if capture:
        torch.cuda.synchronize()
        got = np.frombuffer(prim_t.cpu().numpy().tobytes(), dtype=vrc.HIT_DTYPE)
        assert got.tobytes() == oprim.tobytes()
    return st


@pytest.mark.parametrize("pitch", [0.0, -0.5, -1.2])
def test_primary_plus_shadow(scene, textures, pitch):
    st = run_pair(scene, textures, 9, 320, 180, pitch, capture=True)
    assert st.pixels == 320 * 180


def test_gi_sample_mode(scene, textures):
    run_pair(scene, textures, 9, 320, 180, -0.5, spp=4, use_gi=1)


def test_gi_reference_default_shadow_samples(scene, textures):
    """use_samples => 4 identical shadow rays (raycaster.hpp:147)."""
    st = run_pair(scene, textures, 7, 160, 90, -0.5, spp=1, use_gi=1, shadow_samples=0)
    assert st.rays > 160 * 90 * 4


def test_gi_two_bounce_extension(scene, textures):
    run_pair(scene, textures, 7, 160, 90, -0.5, spp=2, use_gi=1, gi_bounces=2)


def test_dof(scene, textures):
    run_pair(scene, textures, 9, 320, 180, -0.5, spp=4, use_gi=1, aperture=0.5, focal_length=60.0)


def test_dof_two_bounce_extension(scene, textures):
    """general camera + 2-bounce extension: the fourth specialisation of the stage-synchronous kernel (k_render_sync2)"""
    run_pair(scene, textures, 7, 160, 90, -0.5, spp=3, use_gi=1, gi_bounces=2, aperture=0.7, focal_length=40.0)


@pytest.mark.parametrize("aperture,focal_length", [(0.0, 1.0), (0.0, 37.5), (0.0, -1.0), (-0.0, 1.0), (0.0, 1e-35), (1e-30, 1.0)])
def test_pinhole_shortcut_is_exact(scene, textures, aperture, focal_length):
    """aperture == +0 skips the two lens draws (rand_vec = 0 * getRand() is +-0): exact only while focal_point.x / .y
    cannot be -0, so negative / tiny focal lengths and a -0 aperture must take the general path.  Even widths put
    lens.x = +0 on the centre column (the case where the sign of a zero would show), primary hits are compared too."""
    run_pair(scene, textures, 7, 160, 90, -0.5, spp=4, use_gi=1, aperture=aperture, focal_length=focal_length, capture=True)


def test_temporal_blend_checkerboard(scene, textures):
    """Non-sample mode: 0.4/0.6 temporal blend (raycaster.hpp:77-85) over 4 checkerboard frames."""
    run_pair(scene, textures, 9, 320, 180, -0.5, frames=4, use_gi=1, use_samples=0, checker=True)


@pytest.mark.parametrize("size", [(161, 93), (50, 38), (7, 5), (16, 8), (17, 9)])
@pytest.mark.parametrize("use_samples", [0, 1])
def test_checkerboard_sizes(scene, textures, size, use_samples):
    """Checkerboard frames (main.cpp:137,143: every other pixel inside each of the 4 x 4 swarm areas) at sizes that are
    not multiples of the 16 x 8 tiles the stage-synchronous kernel uses for them, both sinks, alternating parity."""
    run_pair(scene, textures, 7, size[0], size[1], -0.5, frames=3, spp=1 + use_samples, use_gi=1, use_samples=use_samples,
             checker=True)


def test_odd_sizes(scene, textures):
    run_pair(scene, textures, 7, 161, 93, -0.8, spp=4, use_gi=1)
    run_pair(scene, textures, 7, 7, 5, -0.8, spp=1, use_gi=0)


@pytest.mark.parametrize("tuning", [(1, 1, 64), (8, 64, 4096), (2, 32, 128)])
def test_tuning_does_not_change_results(scene, textures, tuning):
    run_pair(scene, textures, 7, 160, 90, -0.5, spp=2, use_gi=1, tuning=tuning)


def test_sharded_frames_compose(scene, textures):
    """Row-block sharding (multi-GPU path) on one GPU: 3 shards rendered separately,
    packed, concatenated as an all-gather would, unpacked == unsharded frame."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene[7]
    W, H, RB, N = 200, 100, 8, 3
    light = vrc.reference_light(7)
    cam = vrc.reference_camera(7, pitch=-0.5)
    full = vrc.RayCaster(svo, (W, H))
    full.setLightPosition(light)
    full.use_gi = full.use_samples = True
    full.shadow_samples = 1
    full.renderFrame(cam, spp=2)
    full.samples_to_image()
    ref = full.readImage()
    L = vrc.capi.load()
    nbytes = L.vrc_shard_bytes(W, H, RB, N)
    gathered = torch.zeros(N * nbytes, dtype=torch.uint8, device="cuda")
    total_rays = 0
    for k in range(N):
        rc = vrc.RayCaster(svo, (W, H))
        rc.setLightPosition(light)
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        rc.renderFrame(cam, spp=2, row_block=RB, shard_index=k, shard_count=N)
        rc.samples_to_image()
        rc.packShard(RB, k, N, gathered.data_ptr() + k * nbytes)
        total_rays += rc.stats().rays
    out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(gathered.data_ptr()), W, H, RB, N, vrc.capi.ptr(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(H, W, 4), ref)
    assert total_rays == full.stats().rays


@pytest.mark.parametrize("parity", [0, 1])
def test_sharded_checkerboard_frames_compose(scene, textures, parity):
    """Sharding and the checkerboard together (the 16 x 8 checkerboard tiles are laid over the shard's compact rows,
    the checker pattern follows the global row): 3 shards composed == the unsharded checkerboard frame."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene[7]
    W, H, RB, N = 203, 101, 8, 3
    light = vrc.reference_light(7)
    cam = vrc.reference_camera(7, pitch=-0.5)

    def make():
        rc = vrc.RayCaster(svo, (W, H))
        rc.setLightPosition(light)
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        return rc

    full = make()
    full.renderFrame(cam, spp=2, checker_parity=parity)
    full.samples_to_image()
    ref = full.readImage()
    L = vrc.capi.load()
    nbytes = L.vrc_shard_bytes(W, H, RB, N)
    gathered = torch.zeros(N * nbytes, dtype=torch.uint8, device="cuda")
    rays = 0
    for k in range(N):
        rc = make()
        rc.renderFrame(cam, spp=2, checker_parity=parity, row_block=RB, shard_index=k, shard_count=N)
        rc.resolveShard(RB, k, N, gathered.data_ptr() + k * nbytes, reset=True)
        rays += rc.stats().rays
    out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(gathered.data_ptr()), W, H, RB, N, vrc.capi.ptr(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(H, W, 4), ref)
    assert rays == full.stats().rays and rays > 0


def test_resolve_shard_equals_resolve_pack_reset(scene, textures):
    """vrc_resolve_shard = samples_to_image + pack_shard (+ reset_samples) in one pass: same shard bytes, same image
    rows, accumulators of the shard's rows cleared and nobody else's, and two frames in a row through it (the
    work-queue heads are reset by the same kernel) equal two frames through the separate calls."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene[7]
    W, H, RB, N = 200, 100, 8, 3
    light = vrc.reference_light(7)
    cam = vrc.reference_camera(7, pitch=-0.5)
    L = vrc.capi.load()
    nbytes = L.vrc_shard_bytes(W, H, RB, N)

    def make():
        rc = vrc.RayCaster(svo, (W, H))
        rc.setLightPosition(light)
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        return rc

    for k in range(N):
        a, b = make(), make()
        sa = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        sb = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        for frame in range(2):
            a.resetSamples()
            a.renderFrame(cam, spp=2, row_block=RB, shard_index=k, shard_count=N)
            a.samples_to_image()
            a.packShard(RB, k, N, sa.data_ptr())
            b.renderFrame(cam, spp=2, row_block=RB, shard_index=k, shard_count=N)
            b.resolveShard(RB, k, N, sb.data_ptr(), reset=True)
            torch.cuda.synchronize()
            assert torch.equal(sa, sb)
            assert np.array_equal(a.readImage(), b.readImage())
            assert not b.readAccum().any()
    # whole frame, no reset: accumulators stay, image equals samples_to_image
    a, b = make(), make()
    a.renderFrame(cam, spp=3)
    b.renderFrame(cam, spp=3)
    a.samples_to_image()
    b.resolveShard()
    assert np.array_equal(a.readImage(), b.readImage())
    assert np.array_equal(a.readAccum(), b.readAccum()) and b.readAccum().any()
    # reset only touches the shard's rows
    b.resolveShard(RB, 1, N, None, reset=True)
    acc = b.readAccum()
    rows = np.arange(H)
    mine = (rows // RB) % N == 1
    assert not acc[mine].any() and np.array_equal(acc[~mine], a.readAccum()[~mine])


def test_full_size_properties(scene, textures):
    """BASELINE config 3 size (512^3, 1920x1080, GI): size-independent checks."""
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene[9]
    W, H = 1920, 1080
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(9))
    rc.use_gi = rc.use_samples = True
    rc.shadow_samples = 1
    cam = vrc.reference_camera(9, pitch=-0.5)
    rc.renderFrame(cam, spp=2)
    a1 = rc.readAccum()
    st1 = rc.stats(reset=True)
    assert (a1[..., 3] == 2).all()
    assert st1.pixels == W * H * 2 and W * H * 2 <= st1.rays <= W * H * 2 * 4
    # determinism / idempotence: reset and re-render the same sample indices
    rc.resetSamples()
    rc.frame_index = 0
    rc.renderFrame(cam, spp=2)
    assert np.array_equal(rc.readAccum(), a1)
    st2 = rc.stats()
    assert (st2.rays, st2.sum_complexity) == (st1.rays, st1.sum_complexity)
    # linearity of accumulation: 2 x spp=1 == 1 x spp=2
    rc.resetSamples()
    rc.frame_index = 0
    rc.renderFrame(cam, spp=1)
    rc.renderFrame(cam, spp=1)
    assert np.array_equal(rc.readAccum(), a1)
    # a strided band of rows against the oracle
    top, side = textures
    ocam = oracle_cam(cam)
    oacc = None
    for s in range(2):
        p = O.make_params(W, H, vrc.reference_light(9), use_gi=1, use_samples=1, shadow_samples=1, frame_index=s,
                          row_block=8, shard_index=5, shard_count=16)
        _, oacc, _, _ = O.render_frame(nodes, 9, top, side, ocam, p, accum=oacc, threads=8)
    rows = [y for y in range(H) if (y // 8) % 16 == 5]
    assert np.array_equal(a1[rows], oacc[rows])


def test_camera_inside_solid_terminates(scene, textures):
    """A camera inside a solid voxel gives t = 0 hits with an all-zero normal
    (App. A.10); the GI direction is then NaN, which the reference cannot survive.
    Both sides treat the NaN ray as a miss; the frame must match and terminate."""
    import cpuvoxelraycaster_amd as vrc
    nodes, svo = scene[7]
    top, side = textures
    W, H = 64, 40
    rc = vrc.RayCaster(svo, (W, H))
    light = vrc.reference_light(7)
    rc.setLightPosition(light)
    rc.use_gi = rc.use_samples = True
    rc.shadow_samples = 1
    cam = vrc.make_camera((64.0, 50.0, 64.0), vrc.make_rotation(0.0, -0.5))   # inside the (reflected) terrain band
    rc.renderFrame(cam, spp=1)
    p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1)
    _, oacc, prim, st = O.render_frame(nodes, 7, top, side, oracle_cam(cam), p, want_prim=True, threads=4)
    assert (np.abs(prim["normal"]).sum(1) == 0)[prim["hit"] != 0].all()
    assert np.array_equal(rc.readAccum(), oacc)
    s = rc.stats()
    assert (s.rays, s.sum_complexity) == (st.rays, st.sum_complexity)


@pytest.mark.parametrize("spp", [5, 9])
def test_many_samples_per_call(scene, textures, spp):
    """spp above the tile kernel's 4-samples-per-pass pool: full passes + a partial one."""
    run_pair(scene, textures, 7, 96, 56, -0.5, spp=spp, use_gi=1)


def test_temporal_blend_multiple_samples_per_call(scene, textures):
    """Non-sample mode with spp > 1: the 0.4/0.6 blend is applied once per sample, in order."""
    run_pair(scene, textures, 7, 96, 56, -0.5, spp=3, frames=2, use_gi=1, use_samples=0)


@pytest.mark.parametrize("chunk", [1, 2, 3])
def test_sample_chunked_units(scene, textures, chunk, frame_kernel):
    """Sync kernel: a pixel's samples split over several work units (multi-GPU small shards),
    accumulated with integer atomics -- same sums, same counters."""
    import cpuvoxelraycaster_amd as vrc
    L = vrc.capi.load()
    vrc.capi.check(L.vrc_set_sample_chunk(chunk))
    try:
        run_pair(scene, textures, 7, 96, 56, -0.5, spp=5, use_gi=1)
        run_pair(scene, textures, 7, 96, 56, -0.5, spp=2, frames=2, use_gi=1, use_samples=0)   # blend mode ignores it
    finally:
        vrc.capi.check(L.vrc_set_sample_chunk(0))


def test_gpu_pack_unpack_match_documented_layout(scene, textures):
    """k_pack_shard / k_unpack_shards against the numpy restatement of the shard layout in
    cpuvoxelraycaster_amd/dist.py (the one the world-size-2 gloo test exercises)."""
    import torch
    import cpuvoxelraycaster_amd as vrc
    import cpuvoxelraycaster_amd.dist as vd
    nodes, svo = scene[7]
    W, H, N = 203, 101, 3
    rc = vrc.RayCaster(svo, (W, H))
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    rc.writeImage(img)
    L = vrc.capi.load()
    nbytes = L.vrc_shard_bytes(W, H, vd.ROW_BLOCK, N)
    assert nbytes == vd.shard_bytes(W, H, vd.ROW_BLOCK, N)
    parts = []
    for k in range(N):
        buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        rc.packShard(vd.ROW_BLOCK, k, N, buf.data_ptr())
        torch.cuda.synchronize()
        got = buf.cpu().numpy()
        assert np.array_equal(got, vd.pack_shard_np(img, vd.ROW_BLOCK, k, N))
        parts.append(got)
    gathered = torch.from_numpy(np.concatenate(parts)).cuda()
    out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(gathered.data_ptr()), W, H, vd.ROW_BLOCK, N, vrc.capi.ptr(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(H, W, 4), img)
    assert np.array_equal(vd.unpack_shards_np(np.concatenate(parts), W, H, vd.ROW_BLOCK, N), img)
