"""T1 parity: the HIP batch castRay (vrc_cast_rays, through the C ABI) against the
CPU oracle, bit for bit, on the same seeded rays."""
import numpy as np
import pytest

import oracle_lib as O
import raygen

pytestmark = pytest.mark.gpu

FIELDS = ["position", "normal", "voxel_coord", "hit", "node", "distance", "complexity"]


def assert_hits_equal(a, b):
    assert a.shape == b.shape
    for f in FIELDS:
        x = np.ascontiguousarray(a[f]).view(np.uint32)
        y = np.ascontiguousarray(b[f]).view(np.uint32)
        if not np.array_equal(x, y):
            bad = np.nonzero((x != y).reshape(len(a), -1).any(1))[0]
            raise AssertionError(f"field {f}: {len(bad)} of {len(a)} rays differ, first {bad[:5]}: "
                                 f"{a[bad[0]]} vs {b[bad[0]]}")


@pytest.fixture(scope="module")
def terrain(built, heights):
    import cpuvoxelraycaster_amd as vrc
    out = {}
    for depth in (7, 9):
        nodes = vrc.build_terrain_lsvo(heights, depth)
        out[depth] = (nodes, vrc.LSVO(nodes, depth))
    return out


@pytest.mark.parametrize("depth", [7, 9])
@pytest.mark.parametrize("pitch", [0.0, -0.5, -1.2])
def test_primary_camera_rays(terrain, depth, pitch):
    nodes, svo = terrain[depth]
    org, d = raygen.camera_rays(depth, 480, 270, pitch)
    got = svo.castRays(org, d)
    ref = O.cast_rays(nodes, depth, org, d, threads=8)
    assert_hits_equal(got, ref)
    assert (got["hit"] != 0).sum() > 1000


@pytest.mark.parametrize("depth", [7, 9])
def test_mixed_rays_with_edge_cases(terrain, depth):
    nodes, svo = terrain[depth]
    org, d = raygen.mixed_rays(200000, seed=depth)
    got = svo.castRays(org, d)
    ref = O.cast_rays(nodes, depth, org, d, threads=8)
    assert_hits_equal(got, ref)


@pytest.mark.parametrize("coef,bias", [(0.5, 0.0), (0.05, 0.0), (0.0, 0.01), (2.0, 0.001)])
def test_lod_rays(terrain, coef, bias):
    """ray_size_coef / ray_size_bias LOD cut-off (lsvo.hpp:82-85), as GI rays use it."""
    nodes, svo = terrain[9]
    org, d = raygen.mixed_rays(100000, seed=77)
    got = svo.castRays(org, d, coef=coef, bias=bias)
    ref = O.cast_rays(nodes, 9, org, d, coef=coef, bias=bias, threads=8)
    assert_hits_equal(got, ref)
    assert ((got["hit"] & 0xff) == 2).sum() > 100        # LOD hits occur


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_small_scenes_exhaustive_grid(built, depth):
    """Hand-sized scenes: random voxels, rays from a regular grid of origins in [0.5, 2.5]^3."""
    import cpuvoxelraycaster_amd as vrc
    S = 1 << depth
    rng = np.random.default_rng(depth)
    vol = (rng.random((S, S, S)) < 0.2).astype(np.uint8)
    nodes = vrc.build_volume_lsvo(vol, depth)
    svo = vrc.LSVO(nodes, depth)
    g = np.linspace(0.5, 2.5, 17, dtype=np.float32)
    org = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    dirs = rng.normal(size=(8, 3)).astype(np.float32)
    for dv in dirs:
        d = np.tile(dv, (org.shape[0], 1))
        assert_hits_equal(svo.castRays(org, d), O.cast_rays(nodes, depth, org, d))


def test_empty_and_single_voxel_scene(built):
    import cpuvoxelraycaster_amd as vrc
    org, d = raygen.mixed_rays(4096, seed=5)
    vol = np.zeros((8, 8, 8), np.uint8)
    nodes = vrc.build_volume_lsvo(vol, 3)
    # an empty scene has one node; pad to the 8-slot block the root points at
    svo_nodes = np.concatenate([nodes, np.zeros(8, nodes.dtype)])
    got = vrc.LSVO(svo_nodes, 3).castRays(org, d)
    assert_hits_equal(got, O.cast_rays(svo_nodes, 3, org, d))
    assert not got["hit"].any()
    vol[3, 4, 5] = 1
    nodes = vrc.build_volume_lsvo(vol, 3)
    got = vrc.LSVO(nodes, 3).castRays(org, d)
    assert_hits_equal(got, O.cast_rays(nodes, 3, org, d))


def test_zero_rays_and_single_ray(terrain):
    nodes, svo = terrain[9]
    assert len(svo.castRays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))) == 0
    org = np.array([0.5 + 1, 200 / 512 + 1, 0.5 + 1], np.float32)
    d = np.array([0.0, 0.6, 0.8], np.float32)
    one = svo.castRay(org, d)
    ref = O.cast_rays(nodes, 9, org[None], d[None])[0]
    assert one.tobytes() == ref.tobytes()


def test_full_size_properties(terrain):
    """BASELINE-size batch (1920x1080 primaries at 512^3): size-independent properties."""
    nodes, svo = terrain[9]
    org, d = raygen.camera_rays(9, 1920, 1080, -0.5)
    got = svo.castRays(org, d)
    hit = got["hit"] != 0
    assert hit.mean() > 0.95
    # hit positions lie on the ray and inside the cube; normals are the reference's (+-1,+-2,+-4) set
    p = got["position"][hit]
    assert (p >= 1).all() and (p < 2).all()
    t = got["distance"][hit]
    assert np.abs(org[hit] + t[:, None] * d[hit] - p).max() < 1e-5
    assert set(np.unique(np.abs(got["normal"][hit]))) <= {0.0, 1.0, 2.0, 4.0}
    # idempotence: same input, same bits
    again = svo.castRays(org, d)
    assert got.tobytes() == again.tobytes()
    # checksum against the oracle on a strided sample of the same rays
    sl = slice(0, None, 37)
    assert_hits_equal(got[sl], O.cast_rays(nodes, 9, org[sl], d[sl], threads=8))


def test_non_finite_rays_terminate_as_misses(terrain):
    """NaN / inf input never returns in the reference (no comparison holds, the walk
    makes no progress); defined as an immediate miss in oracle and kernel alike."""
    nodes, svo = terrain[9]
    org, d = raygen.mixed_rays(4096, seed=11)
    d[::7, 0] = np.nan
    d[1::7, 2] = np.inf
    org[2::7, 1] = np.nan
    org[3::7, 0] = -np.inf
    got = svo.castRays(org, d)
    ref = O.cast_rays(nodes, 9, org, d)
    assert_hits_equal(got, ref)
    bad = ~(np.isfinite(org).all(1) & np.isfinite(d).all(1))
    assert not got["hit"][bad].any() and not got["complexity"][bad].any()


@pytest.mark.parametrize("seed", range(12))
def test_differential_fuzz_random_scenes(built, seed):
    """Random occupancy volumes (depth 3..6, sparse to dense), random rays with random
    LOD parameters: kernel and oracle must agree bit for bit on every field."""
    import cpuvoxelraycaster_amd as vrc
    rng = np.random.default_rng(1000 + seed)
    depth = int(rng.integers(3, 7))
    S = 1 << depth
    density = float(rng.choice([0.002, 0.02, 0.1, 0.4, 0.9]))
    vol = (rng.random((S, S, S)) < density).astype(np.uint8)
    if seed % 3 == 0:       # structured: a few solid slabs and columns
        vol[:] = 0
        vol[:, S // 3, :] = 1
        vol[S // 2, :, S // 4] = 1
        vol[rng.integers(0, S, 30), rng.integers(0, S, 30), rng.integers(0, S, 30)] = 1
    svo = vrc.LSVO.fromVolume(vol, depth)
    nodes = svo.downloadNodes()
    org, d = raygen.mixed_rays(40000, seed=seed)
    coef = rng.choice([0.0, 0.0, 0.25, 0.5, 1.5], size=len(org)).astype(np.float32)
    bias = rng.choice([0.0, 0.0, 0.0, 0.002, 0.05], size=len(org)).astype(np.float32)
    got = svo.castRays(org, d, coef=coef, bias=bias)
    ref = O.cast_rays(nodes, depth, org, d, coef=coef, bias=bias, threads=8)
    assert_hits_equal(got, ref)


def test_maximum_depth_scene(built):
    """VRC_MAX_DEPTH = 11 (2048^3): a sparse scene built by the oracle's setCell + compileSVO."""
    import cpuvoxelraycaster_amd as vrc
    depth, S = 11, 2048
    rng = np.random.default_rng(42)
    vox = rng.integers(0, S, (300, 3))
    vox = np.concatenate([vox, [[0, 0, 0], [S - 1, S - 1, S - 1], [1024, 1024, 1024], [1023, 1023, 1023]]])
    nodes = O.compile_voxels(depth, vox)
    svo = vrc.LSVO(nodes, depth)
    # rays aimed at the voxels from random origins, plus random rays.  The walk addresses child
    # slots with inverted position bits (lsvo.hpp:79), i.e. it sees the scene point-reflected
    # through the cube centre: voxel v appears at S-1-v.
    tgt = (S - 1 - vox[rng.integers(0, len(vox), 20000)] + 0.5) / S + 1.0
    org = rng.uniform(1.0, 2.0, tgt.shape)
    d = tgt - org
    o2, d2 = raygen.mixed_rays(20000, seed=9)
    org = np.concatenate([org, o2]).astype(np.float32)
    d = np.concatenate([d, d2]).astype(np.float32)
    got = svo.castRays(org, d)
    ref = O.cast_rays(nodes, depth, org, d, threads=8)
    assert_hits_equal(got, ref)
    assert (got["hit"] != 0).sum() > 1000
    # frame kernel at maximum depth: LDS stack of 10 slots
    rc = vrc.RayCaster(svo, (64, 48))
    rc.setLightPosition((0.6, -0.9, 0.4))
    rc.use_gi = rc.use_samples = True
    rc.shadow_samples = 1
    cam = vrc.make_camera((1024.0, 900.0, 1000.0), vrc.make_rotation(0.3, -0.4))
    rc.renderFrame(cam, spp=2)
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), 1.0, 0.0, 1.0)
    top = side = np.full(768, 255, np.uint8)
    acc = None
    for s in range(2):
        p = O.make_params(64, 48, (0.6, -0.9, 0.4), use_gi=1, use_samples=1, shadow_samples=1, frame_index=s)
        _, acc, _, _ = O.render_frame(nodes, depth, top, side, ocam, p, accum=acc, threads=4)
    assert np.array_equal(rc.readAccum(), acc)


def test_argument_validation(built):
    import ctypes as C
    import cpuvoxelraycaster_amd as vrc
    L = vrc.capi.load()
    nodes = np.zeros(9, vrc.LNODE_DTYPE)
    h = C.c_void_p()
    assert L.vrc_scene_create(vrc.capi.ptr(nodes), 9, 12, 0, C.byref(h)) == -1      # depth > VRC_MAX_DEPTH
    assert L.vrc_scene_create(vrc.capi.ptr(nodes), 0, 3, 0, C.byref(h)) == -1       # empty
    assert L.vrc_scene_create(vrc.capi.ptr(nodes), 9, 3, 99, C.byref(h)) == -1      # no such device
    assert b"device" in L.vrc_last_error()
    svo = vrc.LSVO(nodes, 3)
    rc = vrc.RayCaster(svo, (8, 8))
    rc.gi_bounces = 3
    with pytest.raises(vrc.VrcError):
        rc.renderFrame(vrc.make_camera((4, 4, 4), vrc.make_rotation(0, 0)))
    with pytest.raises(vrc.VrcError):
        vrc.RayCaster(svo, (0, 8))


def test_stray_leaf_bits_are_ignored(built, heights):
    """A leaf bit without its child bit is never read by the reference (lsvo.hpp:81,90).  The device copy clears
    such bits at upload (k_sanitize_nodes) so that the walk can test both masks at once: results must equal the
    oracle's on the unmodified array, with and without the LOD cut-off."""
    import cpuvoxelraycaster_amd as vrc
    nodes = vrc.build_terrain_lsvo(heights, 7).copy()
    rng = np.random.default_rng(5)
    stray = rng.integers(0, 256, len(nodes)).astype(np.uint8) & ~nodes["child_mask"]
    assert stray.any()
    nodes["leaf_mask"] |= stray
    svo = vrc.LSVO(nodes, 7)
    org, d = raygen.mixed_rays(100000, seed=3)
    assert_hits_equal(svo.castRays(org, d), O.cast_rays(nodes, 7, org, d, threads=8))
    assert_hits_equal(svo.castRays(org, d, coef=0.5, bias=0.0), O.cast_rays(nodes, 7, org, d, coef=0.5, bias=0.0, threads=8))
    back = svo.downloadNodes()
    assert np.array_equal(back["leaf_mask"], nodes["leaf_mask"] & nodes["child_mask"])
    assert np.array_equal(back["child_mask"], nodes["child_mask"]) and np.array_equal(back["child_offset"], nodes["child_offset"])


@pytest.mark.parametrize("exp2", [-120, -101, -99, -60, -24, 60, 100, 125, 126, 127])
def test_direction_magnitudes_across_the_shortcut_guards(terrain, exp2):
    """Directions scaled by 2^k (the operator takes non-unit directions): t_coef = -1 / |d| switches between the exact
    v_rcp_f32 + Newton form and the IEEE division at |d| = 2^126, and for huge |d| the t values are denormal --
    every field must still equal the oracle's, LOD arguments included."""
    nodes, svo = terrain[9]
    org, d = raygen.camera_rays(9, 192, 108, -0.5)
    o2, d2 = raygen.mixed_rays(4096, 1000 + exp2)
    org, d = np.concatenate([org, o2]), np.concatenate([d, d2])
    with np.errstate(over="ignore"):
        d = (d.astype(np.float64) * 2.0 ** exp2).astype(np.float32)   # the non-unit rays of the set overflow to inf at 2^126: misses on both sides
    assert_hits_equal(svo.castRays(org, d), O.cast_rays(nodes, 9, org, d, threads=8))
    coef = np.full(len(org), 0.5, np.float32)
    bias = np.where(np.arange(len(org)) % 3 == 0, 1e-3, 0.0).astype(np.float32)
    assert_hits_equal(svo.castRays(org, d, coef, bias), O.cast_rays(nodes, 9, org, d, coef, bias, threads=8))
