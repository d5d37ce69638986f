"""Committed regression vectors (tests/tools/make_golden_vectors.py): the oracle must keep
producing them (CPU), and the HIP path must reproduce them on the GPU box where
/root/reference -- and therefore nothing but these fixtures -- exists."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fnv1a64_words(b):
    h = 1469598103934665603
    for v in np.frombuffer(b, np.uint64).tolist():
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


@pytest.fixture(scope="module")
def rays():
    z = np.load(os.path.join(GOLD, "rays_d9_4096.npz"))
    return z["org"], z["dir"], z["coef"], z["bias"], np.frombuffer(z["hits"].tobytes(), O.HIT_DTYPE)


def test_oracle_reproduces_ray_vectors(heights, rays):
    org, d, coef, bias, want = rays
    got = O.cast_rays(O.compile_terrain(9, heights), 9, org, d, coef=coef, bias=bias)
    assert got.tobytes() == want.tobytes()
    assert (want["hit"] != 0).sum() > 500 and ((want["hit"] & 0xff) == 2).sum() > 20


def test_oracle_reproduces_frame_vector(heights, textures):
    import cpuvoxelraycaster_amd as vrc
    z = np.load(os.path.join(GOLD, "frame_d9_160x90.npz"))
    nodes = O.compile_terrain(9, heights)
    top, side = textures
    cam = O.make_camera(vrc.reference_camera_position(9), O.make_rotation(0.0, -0.5))
    acc, rays_n, steps = None, 0, 0
    for s in range(3):
        p = O.make_params(160, 90, vrc.reference_light(9), use_gi=1, use_samples=1, shadow_samples=1, frame_index=s)
        _, acc, _, st = O.render_frame(nodes, 9, top, side, cam, p, accum=acc, threads=8)
        rays_n += st.rays
        steps += st.sum_complexity
    assert np.array_equal(acc, z["accum"]) and (rays_n, steps) == (int(z["rays"]), int(z["steps"]))


def test_lsvo_hashes(built, heights):
    import cpuvoxelraycaster_amd as vrc
    want = json.load(open(os.path.join(GOLD, "lsvo_hashes.json")))
    for depth in (7, 8, 9):
        n = vrc.build_terrain_lsvo(heights, depth)
        assert len(n) == want[str(depth)]["nodes"]
        assert fnv1a64_words(n.tobytes()) == want[str(depth)]["fnv1a64_of_u64_words"]


@pytest.mark.gpu
def test_gpu_reproduces_ray_vectors(built, rays):
    import cpuvoxelraycaster_amd as vrc
    org, d, coef, bias, want = rays
    svo = vrc.LSVO.fromFastNoiseTerrain(9)
    got = svo.castRays(org, d, coef=coef, bias=bias)
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
def test_gpu_reproduces_frame_vector_and_hashes(built, textures):
    import cpuvoxelraycaster_amd as vrc
    z = np.load(os.path.join(GOLD, "frame_d9_160x90.npz"))
    want = json.load(open(os.path.join(GOLD, "lsvo_hashes.json")))
    for depth in (7, 8, 9, 10):
        svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=textures)
        assert svo.n_nodes == want[str(depth)]["nodes"]
        assert fnv1a64_words(svo.downloadNodes().tobytes()) == want[str(depth)]["fnv1a64_of_u64_words"]
        if depth == 9:
            for lane_samples in (1, 4):
                rc = vrc.RayCaster(svo, (160, 90))
                rc.setLaneSamples(lane_samples)
                rc.setLightPosition(vrc.reference_light(9))
                rc.use_gi = rc.use_samples = True
                rc.shadow_samples = 1
                rc.renderFrame(vrc.reference_camera(9), spp=3)
                st = rc.stats()
                assert np.array_equal(rc.readAccum(), z["accum"])
                assert (st.rays, st.sum_complexity) == (int(z["rays"]), int(z["steps"]))
