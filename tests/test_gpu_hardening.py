"""Boundary hardening of the C ABI (include/vrc.h), on the GPU: a malformed LNode array is an error code and never an
out-of-bounds gather; scheduling knobs live in the renderer; the single-ray operator (Camera::getClosestPoint,
camera_controller.hpp:56-60) goes through the scene's pinned slot and is thread-safe."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def try_create(nodes, depth):
    from cpuvoxelraycaster_amd import capi
    L = capi.load()
    h = C.c_void_p()
    nodes = np.ascontiguousarray(nodes)
    rc = L.vrc_scene_create(capi.ptr(nodes), nodes.shape[0], depth, 0, C.byref(h))
    msg = L.vrc_last_error().decode()
    if rc == 0:
        L.vrc_scene_destroy(h)
    return rc, msg


def test_malformed_node_arrays_are_rejected(built, heights):
    import cpuvoxelraycaster_amd as vrc
    good = vrc.build_terrain_lsvo(heights, 7)
    assert try_create(good, 7)[0] == 0
    # truncated: the last child blocks are missing
    rc, msg = try_create(good[:len(good) // 2], 7)
    assert rc == -1 and "past the end" in msg
    # ... while cutting only slots the walk can never fetch is fine: the last block written by compileSVO belongs to a
    # parent of unit voxels, whose children are decided by its leaf bits alone (lsvo.hpp:90-95)
    assert try_create(good[:len(good) - 8], 7)[0] == 0
    # one offset pointing far outside (also past 2^32 when added to the index)
    bad = good.copy()
    internal = np.flatnonzero(bad["child_mask"] & ~bad["leaf_mask"])
    bad["child_offset"][internal[len(internal) // 2]] = 0xFFFFFFF0
    rc, msg = try_create(bad, 7)
    assert rc == -1 and "past the end" in msg
    # a tree deeper than the depth the caller states: the walk's LDS stack has depth - 1 slots
    rc, msg = try_create(good, 6)
    assert rc == -1 and "deeper" in msg
    assert try_create(good, 8)[0] == 0                 # shallower than stated is fine (leaves end every descent)
    # a subtree hung under two different levels (m, a level-2 node, adopts the child block of a level-1 node)
    def internal_slots(a, i):
        return [c for c in range(8) if (a["child_mask"][i] >> c) & 1 and not (a["leaf_mask"][i] >> c) & 1]
    bad = good.copy()
    blk0 = int(bad["child_offset"][0])
    k1, k2 = [blk0 + c for c in internal_slots(bad, 0)[:2]]
    b1, b2 = k1 + int(bad["child_offset"][k1]), k2 + int(bad["child_offset"][k2])
    m = b1 + internal_slots(bad, k1)[0]
    assert m < b2
    bad["child_offset"][m] = b2 - m
    rc, msg = try_create(bad, 7)
    assert rc == -1 and "two different levels" in msg
    # backward offset (a node adopting its own block): parent + offset wraps in 32 bits, which is out of bounds here
    bad = good.copy()
    bad["child_offset"][k1] = np.uint32((blk0 - k1) % (1 << 32))
    rc, msg = try_create(bad, 7)
    assert rc == -1 and "past the end" in msg
    # stray leaf bits (leaf without child) stay harmless; tiny hand-made scenes stay valid
    stray = good.copy()
    stray["leaf_mask"] |= 0xA5
    assert try_create(stray, 7)[0] == 0
    for depth in (2, 3, 4):
        rng = np.random.default_rng(depth)
        S = 1 << depth
        vox = [tuple(v) for v in rng.integers(0, S, (S * S, 3))]
        assert try_create(O.compile_voxels(depth, vox), depth)[0] == 0
    assert try_create(O.compile_voxels(3, []), 3)[0] == 0      # empty scene: the root alone


def test_frame_limits_are_errors(built, heights, textures):
    import cpuvoxelraycaster_amd as vrc
    L = vrc.capi.load()
    svo = vrc.LSVO(vrc.build_terrain_lsvo(heights, 7), 7, textures=textures)
    rc = vrc.RayCaster(svo, (64, 40))
    rc.use_samples = True
    cam = vrc.reference_camera(7)
    with pytest.raises(vrc.VrcError):
        rc.renderFrame(cam, spp=65537)
    with pytest.raises(vrc.VrcError):
        rc.setSampleChunk(70000)
    with pytest.raises(vrc.VrcError):
        rc.setTuning(9)
    with pytest.raises(vrc.VrcError):
        rc.setLaneSamples(2)
    rc.renderFrame(cam, spp=2)                          # still usable
    assert rc.stats().pixels == 64 * 40 * 2
    assert L.vrc_set_tuning(9) == -1 and L.vrc_set_lane_samples(3) == -1


def test_failed_call_does_not_poison_the_next_frames(built, heights, textures):
    """a call that fails after the work-queue bookkeeping (here: more work units than 32 bits number)
    must not leave a queue set marked as zeroed that no launch ever zeroed -- the frames after it would find every head
    exhausted, render nothing and resolve a stale image."""
    import cpuvoxelraycaster_amd as vrc
    svo = vrc.LSVO(vrc.build_terrain_lsvo(heights, 7), 7, textures=textures)
    cam = vrc.reference_camera(7, pitch=-0.5)
    W, H = 3840, 2160

    def make():
        rc = vrc.RayCaster(svo, (W, H))
        rc.setLightPosition(vrc.reference_light(7))
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        return rc

    good, bad = make(), make()
    for rc in (good, bad):
        rc.renderFrameResolved(cam, spp=2)
    bad.setSampleChunk(1)
    with pytest.raises(vrc.VrcError, match="work units"):
        bad.renderFrameResolved(cam, spp=65536)           # 129 600 tiles x 65 536 units
    bad.setSampleChunk(0)
    for f in range(3):                                    # both queue sets come round
        for rc in (good, bad):
            rc.renderFrameResolved(cam, spp=2)
        assert np.array_equal(bad.readImage(), good.readImage()), f
        sg, sb = good.stats(), bad.stats()
        assert (sb.rays, sb.sum_complexity, sb.pixels) == (sg.rays, sg.sum_complexity, sg.pixels)


def test_scheduling_knobs_belong_to_the_renderer(built, heights, textures):
    """Two renderers on one scene with different frame kernels / chunkings, used alternately: identical results;
    process-wide setters only seed renderers created afterwards."""
    import cpuvoxelraycaster_amd as vrc
    L = vrc.capi.load()
    svo = vrc.LSVO(vrc.build_terrain_lsvo(heights, 7), 7, textures=textures)
    cam = vrc.reference_camera(7, pitch=-0.5)

    def make():
        rc = vrc.RayCaster(svo, (160, 90))
        rc.setLightPosition(vrc.reference_light(7))
        rc.use_gi = rc.use_samples = True
        rc.shadow_samples = 1
        return rc

    a, b, c = make(), make(), make()
    a.setLaneSamples(1)
    b.setTuning(3)
    c.setSampleChunk(1)
    vrc.capi.check(L.vrc_set_lane_samples(4))            # must not affect a, b, c
    try:
        d = make()                                       # ... but seeds d
    finally:
        vrc.capi.check(L.vrc_set_lane_samples(0))
    for f in range(2):
        for rc in (a, b, c, d):
            rc.renderFrame(cam, spp=3)
    ref = a.readAccum()
    for rc in (b, c, d):
        assert np.array_equal(rc.readAccum(), ref)
    sa = a.stats()
    for rc in (b, c, d):
        s = rc.stats()
        assert (s.rays, s.sum_complexity) == (sa.rays, sa.sum_complexity)


def test_single_ray_slot(built, heights):
    """vrc_cast_ray: same HitPoints as the batch call and the oracle, from several threads at once, with and without
    LOD arguments, while a frame is in flight on another stream."""
    import cpuvoxelraycaster_amd as vrc
    from raygen import camera_rays, mixed_rays
    depth = 9
    nodes = vrc.build_terrain_lsvo(heights, depth)
    svo = vrc.LSVO(nodes, depth)
    org, d = camera_rays(depth, 64, 36, -0.5)
    o2, d2 = mixed_rays(512, 11)
    org, d = np.concatenate([org[::9], o2]), np.concatenate([d[::9], d2])
    n = org.shape[0]
    coef = np.where(np.arange(n) % 3 == 0, 0.5, 0.0).astype(np.float32)
    bias = np.where(np.arange(n) % 5 == 0, 1e-4, 0.0).astype(np.float32)
    ref = O.cast_rays(nodes, depth, org, d, coef, bias)
    assert svo.castRays(org, d, coef, bias).tobytes() == ref.tobytes()
    rc = vrc.RayCaster(svo, (640, 360))
    rc.use_gi = rc.use_samples = True
    rc.setLightPosition(vrc.reference_light(depth))
    out = np.zeros(n, vrc.HIT_DTYPE)
    errors = []

    def worker(k, nthreads):
        try:
            for i in range(k, n, nthreads):
                out[i] = svo.castRay(org[i], d[i], float(coef[i]), float(bias[i]))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    rc.renderFrame(vrc.reference_camera(depth), spp=4)   # asynchronous; the single rays run beside it
    threads = [threading.Thread(target=worker, args=(k, 4)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    assert out.tobytes() == ref.tobytes()
    assert rc.stats().pixels == 640 * 360 * 4


def test_arithmetic_shortcuts_are_exact(built):
    """-1 / |d|, normalize's 1 / sqrt and getRand's / 100 use v_rcp_f32 / v_rsq_f32 + one FMA correction inside ranges where
    that equals the correctly rounded IEEE result: checked here for EVERY float bit pattern of those ranges (2^32
    reciprocals, 2^31 roots) on the device that runs the tests."""
    from cpuvoxelraycaster_amd import capi
    out = np.zeros(4, np.uint64)
    capi.check(capi.load().vrc_selftest_exact_arith(0, capi.ptr(out)))
    assert out.tolist() == [0, 0, 0, 0]


def test_host_batch_calls_from_several_threads(built, heights):
    """vrc_cast_rays with host buffers keeps one staging block per device between calls; threads that find it busy
    allocate privately -- every call returns its own rays' HitPoints whatever the interleaving and the batch sizes."""
    import cpuvoxelraycaster_amd as vrc
    from raygen import mixed_rays
    depth = 7
    nodes = vrc.build_terrain_lsvo(heights, depth)
    svo = vrc.LSVO(nodes, depth)
    errors = []

    def worker(k):
        try:
            for j in range(12):
                n = [17, 4096, 1, 70000, 300][(j + k) % 5]
                org, d = mixed_rays(max(n, 16), 100 * k + j)
                org, d = org[:n], d[:n]
                got = svo.castRays(org, d)
                ref = O.cast_rays(nodes, depth, org, d)
                if got.tobytes() != ref.tobytes():
                    errors.append((k, j, n))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors


def test_frame_flag_segments_are_not_taken_from_a_live_owner(built, tmp_path):
    """vrc_ipc_flags_open(create) used to unlink any segment of the same name.  Now: a segment whose owner is
    alive is refused; the leftover of a process that died is replaced; an opener with another count is refused before the
    mapping is touched (no SIGBUS); host-side flag stores and loads work; out-of-range indices are errors."""
    import ctypes as C
    import subprocess
    import sys
    import time
    import cpuvoxelraycaster_amd as vrc
    L = vrc.capi.load()
    name = b"/vrc_test_flags_%d" % os.getpid()
    child_src = (
        "import sys, time, ctypes as C\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import cpuvoxelraycaster_amd as vrc\n"
        "L = vrc.capi.load(); f = C.c_void_p()\n"
        "vrc.capi.check(L.vrc_ipc_flags_open(sys.argv[1].encode(), 16, 0, 1, C.byref(f)))\n"
        "L.vrc_ipc_flag_set(f, 3, 77)\n"
        "print('ready', flush=True)\n"
        "if sys.argv[2] == 'stay': time.sleep(60)\n"
        "import os; os._exit(0)\n")                            # no close: the segment stays behind, as after a crash
    script = tmp_path / "owner.py"
    script.write_text(child_src)
    # (1) a live owner: refused, and the owner's segment is still there for an opener with the right count
    p = subprocess.Popen([sys.executable, str(script), name.decode(), "stay"], stdout=subprocess.PIPE, text=True)
    try:
        assert p.stdout.readline().strip() == "ready"
        f = C.c_void_p()
        assert L.vrc_ipc_flags_open(name, 16, 0, 1, C.byref(f)) == -1 and b"in use by process" in L.vrc_last_error()
        assert L.vrc_ipc_flags_open(name, 17, 0, 0, C.byref(f)) == -1            # made for another count
        assert L.vrc_ipc_flags_open(name, 4096, 0, 0, C.byref(f)) == -1          # larger than the segment: refused, not mapped
        vrc.capi.check(L.vrc_ipc_flags_open(name, 16, 0, 0, C.byref(f)))
        assert L.vrc_ipc_flag_value(f, 3) == 77
        assert L.vrc_ipc_flag_set(f, 16, 1) == -1 and L.vrc_ipc_flag_value(f, 16) == 0
        vrc.capi.check(L.vrc_ipc_flags_close(f))
    finally:
        p.kill()
        p.wait()
    # (2) the owner is gone and left the segment behind: the next creator replaces it (zeros again)
    f = C.c_void_p()
    vrc.capi.check(L.vrc_ipc_flags_open(name, 16, 0, 1, C.byref(f)))
    assert L.vrc_ipc_flag_value(f, 3) == 0
    # (3) the watchdog on an idle stream is an immediate OK; with a pid that does not exist and work pending it would give up
    st = C.c_void_p()
    vrc.capi.check(L.vrc_stream_create(0, C.byref(st)))
    assert L.vrc_ipc_stream_wait(f, st, None, 0, 1000) == 0
    vrc.capi.check(L.vrc_stream_destroy(0, st))
    vrc.capi.check(L.vrc_ipc_flags_close(f))
    assert L.vrc_ipc_flags_open(name, 16, 0, 0, C.byref(f)) == -1                # closed by its owner: unlinked
