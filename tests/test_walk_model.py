"""The formulation of the HIP walk (csrc/vrc_walk.h) checked against the oracle on the CPU, ray by ray, on chains of rays shaped
like the ones a frame casts (raycaster.hpp:118-207): primary -> shadow ray from its hit, GI ray from its hit -> GI shadow ray.

oracle/vrc_walk_model.c restates what the kernel does differently from lsvo.hpp:33-172 -- no t_max on the stack (it is always
min(1, exit of the parent cell)) and secondary rays that start at the deepest ancestor they share with the previous hit instead
of at the root -- and must return the oracle's hit, bit for bit, with the oracle's iteration count (skipped iterations
included)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

STATE_DTYPE = np.dtype([("rows", "<u4", 24), ("scale", "<u4"), ("valid", "<u4")])


def model_lib():
    path = os.path.join(O.ORACLE_DIR, "libvrc_walk_model.so")
    src = os.path.join(O.ORACLE_DIR, "vrc_walk_model.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", O.ORACLE_DIR, "libvrc_walk_model.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orc_model_cast_rays.argtypes = [vp, C.c_uint32, C.c_uint64, vp, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp, C.c_int]
    L.orc_model_cast_camera_rays.argtypes = [vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_int]
    return L


def model_cast_camera(L, nodes, depth, cam_pos, dir_, threads=8):
    """primary rays of a pinhole camera: all from cam_pos, started on the camera cell's path"""
    n = len(dir_)
    dir_ = np.ascontiguousarray(dir_, np.float32)
    cam_pos = np.ascontiguousarray(cam_pos, np.float32)
    out = np.zeros(n, O.HIT_DTYPE)
    st = np.zeros(n, STATE_DTYPE)
    skipped = np.zeros(n, np.uint32)
    scales = np.zeros(8, np.uint32)
    L.orc_model_cast_camera_rays(O._ptr(nodes), depth, n, O._ptr(cam_pos), O._ptr(dir_), O._ptr(out), O._ptr(st), O._ptr(skipped),
                                 O._ptr(scales), threads)
    return out, st, skipped, scales


def model_cast(L, nodes, depth, org, dir_, coef, prev=None, prev_pos=None, threads=8):
    n = len(org)
    org = np.ascontiguousarray(org, np.float32)
    dir_ = np.ascontiguousarray(dir_, np.float32)
    out = np.zeros(n, O.HIT_DTYPE)
    st = np.zeros(n, STATE_DTYPE)
    skipped = np.zeros(n, np.uint32)
    pp = np.ascontiguousarray(prev_pos, np.float32) if prev_pos is not None else None
    L.orc_model_cast_rays(O._ptr(nodes), depth, n, O._ptr(org), O._ptr(dir_), coef, 0.0,
                          O._ptr(prev) if prev is not None else None, O._ptr(pp) if pp is not None else None,
                          O._ptr(out), O._ptr(st), O._ptr(skipped), threads)
    return out, st, skipped


def normalize(v):
    """glm::normalize: v * (1 / sqrt(dot(v, v))) in float32 (overflow to inf -> 0 * v, as in the reference)"""
    v = v.astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        n = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]).astype(np.float32)
        inv = (np.float32(1.0) / np.sqrt(n).astype(np.float32)).astype(np.float32)
        return (v * inv[:, None]).astype(np.float32)


def same(a, b):
    return a.tobytes() == b.tobytes()


def chains(nodes, depth, org, dir_, light, rng, L):
    """returns the number of rays compared and the iterations skipped / executed by the secondary rays"""
    f = np.float32
    inv = f(1.0) / f(1 << depth)
    nn = f(inv * f(0.0078125) * f(2.0))
    ref = O.cast_rays(nodes, depth, org, dir_, 0.0, 0.0, threads=8)
    mod, st, sk = model_cast(L, nodes, depth, org, dir_, 0.0)
    assert same(ref, mod) and not sk.any()
    hit = ref["hit"] != 0
    assert hit.sum() > len(org) // 4
    P, stP = ref[hit], np.ascontiguousarray(st[hit])
    total, skipped, iters = len(org), 0, 0
    # shadow ray: raycaster.hpp:139,151-153
    so = (P["position"] + (P["normal"] * inv) * f(0.001)).astype(f)
    sd = normalize(light[None, :] - so)
    ref_s = O.cast_rays(nodes, depth, so, sd, 0.0, 0.0, threads=8)
    mod_s, _, sk = model_cast(L, nodes, depth, so, sd, 0.0, stP, P["position"])
    assert same(ref_s, mod_s)
    total += len(so); skipped += int(sk.sum()); iters += int(ref_s["complexity"].sum())
    # GI ray: :171-194, from the primary hit as well (here through the primary's stack; the kernel may start it at the root)
    go = (P["position"] + P["normal"] * nn).astype(f)
    noise = rng.uniform(-1000.0, 1000.0, (len(go), 3)).astype(f)
    noise[P["normal"] != 0] = 0.0
    gd = normalize((P["normal"] + noise) * nn)
    ok = np.isfinite(gd).all(axis=1)
    go, gd, stG, Pg = go[ok], gd[ok], np.ascontiguousarray(stP[ok]), P[ok]
    ref_g = O.cast_rays(nodes, depth, go, gd, 0.5, 0.0, threads=8)
    mod_g, st_g, sk = model_cast(L, nodes, depth, go, gd, 0.5, stG, Pg["position"])
    assert same(ref_g, mod_g)
    total += len(go); skipped += int(sk.sum()); iters += int(ref_g["complexity"].sum())
    mod_g0, st_g0, sk0 = model_cast(L, nodes, depth, go, gd, 0.5)             # and from the root, as the kernel does
    assert same(ref_g, mod_g0) and not sk0.any()
    # GI shadow ray: :195-198 (LOD coefficient 0.5 as the reference passes it)
    gh = ref_g["hit"] != 0
    G, stGh = ref_g[gh], np.ascontiguousarray(st_g0[gh])
    gso = (G["position"] + G["normal"] * nn).astype(f)
    gsd = normalize(light[None, :] - gso)
    ref_gs = O.cast_rays(nodes, depth, gso, gsd, 0.5, 0.0, threads=8)
    mod_gs, _, sk = model_cast(L, nodes, depth, gso, gsd, 0.5, stGh, G["position"])
    assert same(ref_gs, mod_gs)
    total += len(gso); skipped += int(sk.sum()); iters += int(ref_gs["complexity"].sum())
    return total, skipped, iters


def camera_rays(n, depth, rng, pitch):
    import cpuvoxelraycaster_amd as vrc
    cam = vrc.reference_camera(depth, pitch=pitch, make_rotation=lambda x, y: O.make_rotation(x, y))
    rot = np.asarray(cam.rot, np.float32).reshape(3, 3)
    f = np.float32
    lens = np.stack([rng.uniform(-0.89, 0.89, n), rng.uniform(-0.5, 0.5, n), np.full(n, cam.fov)], axis=1).astype(f)
    d = normalize(lens) @ rot.T
    inv = f(1.0) / f(1 << depth)
    o = (np.asarray(cam.position, f) * inv + f(1.0)).astype(f)
    return np.broadcast_to(o, (n, 3)).copy(), normalize(d)


def test_model_equals_oracle_with_the_light_anywhere():
    """the light inside the cube, on a voxel boundary, and so far away that normalize() of the direction to it returns 0 (then
    lsvo.hpp:44-46 make the direction (+-EPS, +-EPS, +-EPS) and a ray cast with an LOD coefficient ends in its first iteration)"""
    import cpuvoxelraycaster_amd as vrc
    L = model_lib()
    depth = 7
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    rng = np.random.default_rng(11)
    for light in ([1.31, 1.62, 1.4], [1.5, 1.75, 1.25], [3e29, -8e29, 1e30], [1e19, 1e19, -1e19], [-40.0, 900.0, 12.0]):
        org, dir_ = camera_rays(60000, depth, rng, -0.7)
        chains(nodes, depth, org, dir_, np.asarray(light, np.float32), rng, L)


@pytest.mark.parametrize("depth,n", [(7, 120000), (9, 250000)])
def test_model_equals_oracle_on_frame_like_chains(depth, n):
    import cpuvoxelraycaster_amd as vrc
    L = model_lib()
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    rng = np.random.default_rng(depth)
    light = np.asarray(vrc.reference_light(depth), np.float32)
    total = skipped = iters = 0
    for pitch in (-0.5, -1.2, 0.0):
        org, dir_ = camera_rays(n, depth, rng, pitch)
        t, s, i = chains(nodes, depth, org, dir_, light, rng, L)
        total, skipped, iters = total + t, skipped + s, iters + i
    assert skipped > 0.1 * iters            # the start below the root is taken, and for a large part of the iterations
    print(f"depth {depth}: {total} rays; secondary rays: {skipped} of {iters} iterations not executed ({skipped / iters:.1%})")


def test_model_equals_oracle_on_random_rays_and_edges():
    """rays from anywhere (outside the cube, on cell boundaries, axis-parallel), chained to whatever they hit"""
    import cpuvoxelraycaster_amd as vrc
    L = model_lib()
    depth = 6
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    rng = np.random.default_rng(5)
    n = 200000
    f = np.float32
    org = rng.uniform(0.9, 2.1, (n, 3)).astype(f)
    grid = (np.round(org * 64.0) / 64.0).astype(f)
    org[: n // 4] = grid[: n // 4]                                            # on cell boundaries
    org[n // 4: n // 2, 1] = grid[n // 4: n // 2, 1]
    dir_ = normalize(rng.normal(size=(n, 3)))
    dir_[::7, 0] = 0.0
    dir_[::11, 2] = -0.0
    light = np.asarray(vrc.reference_light(depth), f)
    chains(nodes, depth, org, normalize(dir_ + f(0.0)), light, rng, L)


@pytest.mark.parametrize("depth", [7, 9])
def test_camera_rays_start_on_the_camera_path(depth):
    """a pinhole camera's primary rays started at the end of the camera cell's path: cameras in general position, and the
    reference's own pose, whose x and z are exactly 1.5 -- the centre plane of the cube, where lsvo.hpp:66-68 compares
    fl(1.5 t_coef) with itself and every ray deterministically takes the half it travels into"""
    import cpuvoxelraycaster_amd as vrc
    L = model_lib()
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    rng = np.random.default_rng(40 + depth)
    f = np.float32
    S = float(1 << depth)
    ref_cam = vrc.reference_camera(depth, pitch=-0.5, make_rotation=lambda x, y: O.make_rotation(x, y))
    poses = [np.asarray(ref_cam.position, f)]
    poses.append(np.asarray([S / 2, ref_cam.position[1] + 3.25, S / 2 + 7.3], f))          # on the plane in x only
    for _ in range(3):
        poses.append(np.asarray([S / 2 + rng.uniform(-S / 9, S / 9), S / 2 - rng.uniform(0.43, 0.48) * S, S / 2 + rng.uniform(-S / 9, S / 9)], f))
    n, total_skipped = 150000, []
    for k, pos in enumerate(poses):
        cam_pos = (pos * (f(1.0) / f(S)) + f(1.0)).astype(f)
        d = normalize(rng.normal(size=(n, 3)) * np.asarray([1.0, 0.6, 1.0]) + np.asarray([0.0, 0.7, 0.0]))
        d[::5] = normalize(rng.normal(size=(len(d[::5]), 3)))                            # some in every direction
        d[::13, 0] = 0.0
        d[::17, 2] = -0.0
        d = normalize(d)
        org = np.broadcast_to(cam_pos, (n, 3)).copy()
        ref = O.cast_rays(nodes, depth, org, d, 0.0, 0.0, threads=8)
        mod, st, sk, scales = model_cast_camera(L, nodes, depth, cam_pos, d)
        assert same(ref, mod), (k, pos)
        total_skipped.append(int(sk.sum()))
        if k == 0:
            assert (cam_pos[0], cam_pos[2]) == (1.5, 1.5) and sk.min() > 0        # the reference pose starts below the root
        # ... and the shadow rays of those hits start next to them on the path the camera ray's walk left
        hit = ref["hit"] != 0
        P, stP = ref[hit], np.ascontiguousarray(st[hit])
        inv = f(1.0) / f(S)
        so = (P["position"] + (P["normal"] * inv) * f(0.001)).astype(f)
        light = np.asarray(vrc.reference_light(depth), f)
        sd = normalize(light[None, :] - so)
        ref_s = O.cast_rays(nodes, depth, so, sd, 0.0, 0.0, threads=8)
        mod_s, _, _ = model_cast(L, nodes, depth, so, sd, 0.0, stP, P["position"])
        assert same(ref_s, mod_s), (k, pos)
    assert all(t > 0 for t in total_skipped)


def _adversary_rays(nodes, depth, L, n_hits, rng, light, cam_rays):
    """primary hits -> origins k ulps either side of the planes of every scale -> directions (tests/adversaries.py)"""
    import adversaries as A
    org, dir_ = cam_rays
    ref = O.cast_rays(nodes, depth, org, dir_, 0.0, 0.0, threads=8)
    mod, st, _ = model_cast(L, nodes, depth, org, dir_, 0.0)
    assert same(ref, mod)
    hit = np.flatnonzero(ref["hit"] != 0)
    hit = hit[rng.permutation(len(hit))[:n_hits]]
    P, stP = ref[hit], np.ascontiguousarray(st[hit])
    o, owner, meta = A.origins_near_planes(P["position"], depth)
    d, oi = A.directions(o, light, rng)
    return o[oi], d, np.ascontiguousarray(stP[owner[oi]]), np.ascontiguousarray(P["position"][owner[oi]]), meta[oi]


def test_constructed_margin_adversaries():
    """Round-3 verdict item 6: origins constructed k = 0..16 ulps either side of a plane of EVERY scale the tree has, on one, two
    and three axes, x 8 mirror masks x directions with components at +-2^-23 and +-0, with and without the LOD coefficient.  The
    model (= the kernel's formulation) must equal the oracle at the shipped margin of 8 ulps; run at margins 0..8 it records where
    the start below the root first holds -- the error analysis says above 4.5 ulps."""
    import cpuvoxelraycaster_amd as vrc
    L = model_lib()
    margin = C.c_uint32.in_dll(L, "orc_model_margin")
    assert margin.value == 8
    rng = np.random.default_rng(2024)
    f = np.float32
    scenes = []
    depth = 9
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    scenes.append((nodes, depth, camera_rays(4000, depth, rng, -0.6), 24, np.asarray(vrc.reference_light(depth), f)))
    # leaves and empty space at every level: blobs + noise at 64^3 through the oracle's builder
    depth = 6
    S0 = 1 << depth
    g = np.indices((S0, S0, S0)).astype(np.float32)
    vol = rng.random((S0, S0, S0)) < 0.02
    for _ in range(5):
        c, r = rng.uniform(0, S0, 3), rng.uniform(3, S0 / 3)
        vol |= ((g[0] - c[0]) ** 2 + (g[1] - c[1]) ** 2 + (g[2] - c[2]) ** 2) < r * r
    nodes6 = O.compile_voxels(depth, np.argwhere(vol).astype(np.int64))
    org6 = rng.uniform(1.05, 1.95, (4000, 3)).astype(f)
    scenes.append((nodes6, depth, (org6, normalize(rng.normal(size=(4000, 3)))), 24, np.asarray([1.3, 1.9, 1.6], f)))
    bad = np.zeros(9, np.int64)
    total = skipped_at_8 = 0
    try:
        for nodes, depth, cam_rays, n_hits, light in scenes:
            o, d, stP, pp, meta = _adversary_rays(nodes, depth, L, n_hits, rng, light, cam_rays)
            for coef in (0.0, 0.5, 0.25):
                ref = O.cast_rays(nodes, depth, o, d, coef, 0.0, threads=8)
                total += len(o)
                for m in range(9):
                    margin.value = m
                    mod, _, sk = model_cast(L, nodes, depth, o, d, coef, stP, pp)
                    differ = (mod.view(np.uint8).reshape(len(o), -1) != ref.view(np.uint8).reshape(len(o), -1)).any(axis=1)
                    bad[m] += int(differ.sum())
                    if m == 8:
                        skipped_at_8 += int(sk.sum())
                        assert (sk > 0).mean() > 0.5                                     # most of these rays DO start below the root
    finally:
        margin.value = 8
    print(f"{total} constructed rays; rays that differ from the oracle by margin 0..8: {bad.tolist()}")
    assert bad[8] == 0 and skipped_at_8 > 0
    assert bad[0] > 0 and bad[1] > 0                       # the construction bites: without a margin the walk goes wrong
    first_ok = next(m for m in range(9) if not bad[m:].any())
    # measured (9.98 M rays, LOD coefficients 0 / 0.25 / 0.5): [272977, 61245, 504, 0, 0, 0, 0, 0, 0] differ at margins 0..8 -- the walk first holds at 3 ulps, below
    # the 4.5 of the hand-derived bound; the shipped 8 keeps 5 ulps of slack over what these adversaries need
    assert first_ok <= 4, (first_ok, bad.tolist())


def model_cast_lens(L, nodes, depth, cam_pos, org, dir_, threads=8):
    """camera rays of a camera with a lens: origins next to cam_pos, started on the camera cell's path of their side"""
    L.orc_model_cast_lens_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64] + [C.c_void_p] * 7 + [C.c_int]
    n = len(org)
    org, dir_ = np.ascontiguousarray(org, np.float32), np.ascontiguousarray(dir_, np.float32)
    cam_pos = np.ascontiguousarray(cam_pos, np.float32)
    out, st, skipped, scales = np.zeros(n, O.HIT_DTYPE), np.zeros(n, STATE_DTYPE), np.zeros(n, np.uint32), np.zeros(8, np.uint32)
    L.orc_model_cast_lens_rays(O._ptr(nodes), depth, n, O._ptr(cam_pos), O._ptr(org), O._ptr(dir_), O._ptr(out), O._ptr(st), O._ptr(skipped),
                               O._ptr(scales), threads)
    return out, st, skipped, scales


@pytest.mark.parametrize("depth", [7, 9])
def test_lens_rays_start_on_their_side_of_the_centre_planes(depth):
    """A formulation that is proven here and NOT shipped (round 4, DESIGN.md section 9: measured 1 % slower on C4).  With a lens
    (camera_controller.hpp:41,46) the ray starts at position + rand_vec * rot, next to the camera.  The reference's pose sits
    exactly on the cube's centre planes x = z = 1.5, where the kernel starts every lens ray at the root; a lens point is on one
    side of such a plane by POSITION, so it could start on that side's camera path -- unless it is on the plane or within the
    margin of it (getRand's level 50 is exactly 0).  The model does that (orc_model_cast_lens_rays) and equals the oracle.
    Origins: the aperture's real offsets, and constructed ones k = 0..16 ulps either side of the planes, x directions in every
    octant and with +-EPS / +-0 components."""
    import cpuvoxelraycaster_amd as vrc
    L = model_lib()
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    rng = np.random.default_rng(90 + depth)
    f = np.float32
    S = float(1 << depth)
    ref_cam = vrc.reference_camera(depth, pitch=-0.5, make_rotation=lambda x, y: O.make_rotation(x, y))
    poses = [np.asarray(ref_cam.position, f), np.asarray([S / 2, ref_cam.position[1] + 3.25, S / 2 + 7.3], f),
             np.asarray([S / 2 + 11.7, S / 2 - 0.45 * S, S / 2 - 5.1], f)]
    for k, pos in enumerate(poses):
        cam_pos = (pos * (f(1.0) / f(S)) + f(1.0)).astype(f)
        n = 120000
        # rand_vec = aperture * (getRand(), getRand(), 0) with 100 levels in [-0.5, 0.5), rotated; aperture 0.5 .. 4 world units
        ap = rng.choice([0.5, 1.0, 4.0], n)
        lv = (rng.integers(0, 100, (n, 3)) / 100.0 - 0.5) * ap[:, None]
        lv[:, 2] = 0.0
        rot = np.asarray(O.make_rotation(float(rng.uniform(-3, 3)), -0.5), np.float64).reshape(3, 3)
        w = (lv @ rot.T).astype(f)
        org = ((pos[None, :] + w).astype(f) * (f(1.0) / f(S)) + f(1.0)).astype(f)
        # ... a quarter of them constructed: +-k ulps off the camera's coordinate on one or more axes
        m = n // 4
        cb = cam_pos.view(np.uint32).astype(np.int64)
        kk = rng.integers(0, 17, (m, 3)) * rng.choice([-1, 1], (m, 3)) * (rng.random((m, 3)) < 0.6)
        org[:m] = (cb[None, :] + kk).astype(np.uint32).view(f)
        d = normalize(rng.normal(size=(n, 3)))
        d[::13, 0] = 0.0
        d[::17, 2] = -0.0
        d[::19, 1] = f(2.0 ** -23) * f(1.3)
        ref = O.cast_rays(nodes, depth, org, d, 0.0, 0.0, threads=8)
        mod, st, sk, scales = model_cast_lens(L, nodes, depth, cam_pos, org, d)
        assert same(ref, mod), (k, pos)
        on_plane = cam_pos == 1.5
        far = (np.abs(org.view(np.uint32).astype(np.int64) - cb[None, :]) > 8)[:, on_plane].all(axis=1) if on_plane.any() else np.ones(n, bool)
        assert scales.max() < 22
        if on_plane.any():
            assert (sk[far] > 0).mean() > 0.9, (k, (sk[far] > 0).mean())       # off the planes: they start below the root
            near = (np.abs(org.view(np.uint32).astype(np.int64) - cb[None, :]) < 8)[:, on_plane].any(axis=1)
            assert near.sum() > 1000 and (sk[near] == 0).all()                 # on / next to a plane: at the root
        else:
            assert (sk > 0).mean() > 0.9
        # ... and the shadow rays of their hits start next to them on the path the lens ray's walk left
        hit = ref["hit"] != 0
        P, stP = ref[hit], np.ascontiguousarray(st[hit])
        so = (P["position"] + (P["normal"] * (f(1.0) / f(S))) * f(0.001)).astype(f)
        sd = normalize(np.asarray(vrc.reference_light(depth), f)[None, :] - so)
        ref_s = O.cast_rays(nodes, depth, so, sd, 0.0, 0.0, threads=8)
        mod_s, _, _ = model_cast(L, nodes, depth, so, sd, 0.0, stP, P["position"])
        assert same(ref_s, mod_s), (k, pos)
