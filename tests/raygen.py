"""Seeded ray sets for the parity tests (SURVEY.md section 4 case list)."""
import numpy as np


def camera_rays(depth, W, H, pitch, rng=None):
    """Primary rays of the reference camera, as float32 arrays (computed in numpy;
    only used as *inputs* shared by oracle and kernel)."""
    S = float(1 << depth)
    f = np.float32
    xs, ys = np.meshgrid(np.arange(W, dtype=f), np.arange(H, dtype=f))
    lx = xs / f(H) - f(W) / f(H) * f(0.5)
    ly = ys / f(H) - f(0.5)
    v = np.stack([lx, ly, np.ones_like(lx)], -1).reshape(-1, 3).astype(f)
    v /= np.linalg.norm(v, axis=1, keepdims=True).astype(f)
    c, s = f(np.cos(pitch)), f(np.sin(pitch))
    d = np.stack([v[:, 0], c * v[:, 1] - s * v[:, 2], s * v[:, 1] + c * v[:, 2]], -1).astype(f)
    org = np.tile(np.array([S / 2, 200.0 * S / 512.0, S / 2], f) / f(S) + f(1), (d.shape[0], 1)).astype(f)
    return org, d


def mixed_rays(n, seed):
    """Random origins inside / outside the [1,2)^3 cube, random directions, plus the
    edge cases: axis-parallel, zero and negative-zero components, tiny components."""
    rng = np.random.default_rng(seed)
    f = np.float32
    org = rng.uniform(0.0, 3.0, (n, 3)).astype(f)
    inside = rng.random(n) < 0.6
    org[inside] = rng.uniform(1.0, 2.0, (int(inside.sum()), 3)).astype(f)
    d = rng.normal(size=(n, 3)).astype(f)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(f)
    k = n // 16
    # axis-parallel
    for a in range(3):
        sl = slice(a * k, (a + 1) * k)
        d[sl] = 0
        d[sl, a] = rng.choice([-1.0, 1.0], k).astype(f)
    # one zero / negative-zero component
    d[3 * k:4 * k, 0] = 0.0
    d[4 * k:5 * k, 1] = -0.0
    d[5 * k:6 * k, 2] = f(1e-9)
    d[6 * k:7 * k, 0] = f(-1e-8)
    # non-unit directions
    d[7 * k:8 * k] *= f(7.5)
    # origins exactly on voxel boundaries
    org[8 * k:9 * k] = (np.round(org[8 * k:9 * k] * 64) / 64).astype(f)
    return org, d
