"""N1 parity: the GPU LSVO builder (vrc_scene_build_terrain / _volume) against the
oracle's restatement of SVO::setCell + compileSVO -- bit-identical arrays -- and
the scene it leaves in HBM renders identically to an uploaded one."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [4, 6, 7, 8, 9])
def test_terrain_builder_matches_oracle(built, heights, depth):
    import cpuvoxelraycaster_amd as vrc
    svo = vrc.LSVO.fromTerrain(heights, depth)
    got = svo.downloadNodes()
    ref = O.compile_terrain(depth, heights)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))
    assert svo.build_ms is not None and svo.build_ms > 0


@pytest.mark.parametrize("depth", [2, 3, 5])
@pytest.mark.parametrize("density", [0.0, 0.01, 0.3, 1.0])
def test_volume_builder_matches_oracle(built, depth, density):
    import cpuvoxelraycaster_amd as vrc
    S = 1 << depth
    rng = np.random.default_rng(depth * 31 + int(density * 100))
    vol = (rng.random((S, S, S)) < density).astype(np.uint8)
    got = vrc.LSVO.fromVolume(vol, depth).downloadNodes()
    ref = O.compile_voxels(depth, np.argwhere(vol))
    assert np.array_equal(got.view(np.uint64), ref.view(np.uint64))


def test_built_scene_renders_like_uploaded_scene(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    depth, W, H = 8, 320, 180
    a = vrc.LSVO.fromTerrain(heights, depth, textures=textures)
    b = vrc.LSVO(vrc.build_terrain_lsvo(heights, depth), depth, textures=textures)
    imgs = []
    for svo in (a, b):
        rc = vrc.RayCaster(svo, (W, H))
        rc.setLightPosition(vrc.reference_light(depth))
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        rc.renderFrame(vrc.reference_camera(depth), spp=2)
        imgs.append(rc.readAccum())
    assert np.array_equal(imgs[0], imgs[1]) and imgs[0][..., :3].any()


def test_builder_1024_properties(built, heights):
    """1024^3 (BASELINE config 5 scene): node count from SURVEY App. B, root record, determinism."""
    import cpuvoxelraycaster_amd as vrc
    svo = vrc.LSVO.fromTerrain(heights, 10)
    assert svo.n_nodes == 42436609
    n = svo.downloadNodes()
    assert np.all(n["color"] == 1) and n[0]["child_offset"] == 1
    assert int((n["child_mask"] != 0).sum()) * 8 + 1 == svo.n_nodes
    again = vrc.LSVO.fromTerrain(heights, 10).downloadNodes()
    assert np.array_equal(n.view(np.uint64), again.view(np.uint64))


def test_gpu_noise_matches_real_fastnoise(built, heights):
    """N4, pinned by reference execution: the GPU terrain-height kernel against the
    fixture produced by the reference's own FastNoise.cpp (tests/tools/make_golden.py)."""
    import ctypes as C
    import cpuvoxelraycaster_amd as vrc
    h = np.zeros((1024, 1024), np.int32)
    vrc.capi.check(vrc.capi.load().vrc_terrain_heights(1337, 1024, 0, vrc.capi.ptr(h)))
    assert np.array_equal(h, heights)
    other = np.zeros((64, 64), np.int32)
    vrc.capi.check(vrc.capi.load().vrc_terrain_heights(7, 64, 0, vrc.capi.ptr(other)))
    assert np.array_equal(other, O.terrain_heights(64, seed=7)) and not np.array_equal(other, heights[:64, :64])


def test_fastnoise_scene_equals_fixture_scene(built, heights):
    import cpuvoxelraycaster_amd as vrc
    a = vrc.LSVO.fromFastNoiseTerrain(9).downloadNodes()
    b = vrc.LSVO.fromTerrain(heights, 9).downloadNodes()
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)) and len(a) == 10528393
