"""G1 parity: dense-grid DDA kernel (Grid3D::castRay, grid_3d.hpp:36-132) vs the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from test_gpu_cast import assert_hits_equal

pytestmark = pytest.mark.gpu


def terrain_grid(heights, S):
    top = np.maximum(16, np.minimum(S, heights[:S, :S]))
    y = np.arange(S)[None, :, None]
    cells = ((y >= 1 + S // 2) & (y < top[:, None, :] + S // 2)).astype(np.uint8)
    return np.ascontiguousarray(cells)


def test_grid_128_config1(built, heights):
    """BASELINE config 1: 128^3 dense grid, 640x360 primaries."""
    import cpuvoxelraycaster_amd as vrc
    S, W, H = 128, 640, 360
    cells = terrain_grid(heights, S)
    f = np.float32
    xs, ys = np.meshgrid(np.arange(W, dtype=f), np.arange(H, dtype=f))
    v = np.stack([xs / f(H) - f(W) / f(H) * f(0.5), ys / f(H) - f(0.5), np.ones_like(xs)], -1).reshape(-1, 3)
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(f)
    c, s = f(np.cos(-0.5)), f(np.sin(-0.5))
    d = np.stack([v[:, 0], c * v[:, 1] - s * v[:, 2], s * v[:, 1] + c * v[:, 2]], -1).astype(f)
    org = np.tile(np.array([S / 2, 200.0 * S / 512, S / 2], f), (d.shape[0], 1))
    grid = vrc.Grid3D(cells)
    got = grid.castRays(org, d)
    ref = O.grid_cast_rays(cells, org, d, threads=8)
    assert_hits_equal(got, ref)
    assert (got["hit"] != 0).mean() > 0.5


def test_grid_random_rays(built):
    import cpuvoxelraycaster_amd as vrc
    rng = np.random.default_rng(3)
    cells = (rng.random((24, 17, 33)) < 0.05).astype(np.uint8)
    n = 50000
    org = (rng.random((n, 3)) * np.array([24, 17, 33])).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:1000, 0] = 0.0        # division by zero -> inf, as the reference does
    d[1000:2000, 1] = -0.0
    got = vrc.Grid3D(cells).castRays(org, d)
    ref = O.grid_cast_rays(cells, org, d)
    assert_hits_equal(got, ref)
