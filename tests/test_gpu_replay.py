"""N3 on the GPU (include/replay.hpp:8-35): the committed camera-path file tests/golden/orbit.replay is
read by the host adapter's restatement of ReplayElements::loadFromFile and rendered tick by tick through
HipRayCaster::renderFrame (a g++-built C++ program over the C ABI, like the reference's main.cpp would be); every
frame's image hash, ray count and loop-iteration count must equal the oracle's frame for the same tick."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "tests", "golden", "orbit.replay")


def fnv1a(buf):
    h = 1469598103934665603
    for b in bytes(buf):
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_replay_file_through_cpp_adapter_matches_oracle(built, heights, textures, tmp_path):
    import cpuvoxelraycaster_amd as vrc
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replay_bench
    depth, W, H, spp = 9, 160, 90, 2
    ticks = replay_bench.load_replay(REPLAY)
    assert len(ticks) == 24
    nodes = vrc.build_terrain_lsvo(heights, depth)
    top, side = textures
    nodes.tofile(tmp_path / "nodes.bin")
    top.tofile(tmp_path / "top.rgb")
    side.tofile(tmp_path / "side.rgb")
    exe = str(tmp_path / "replay_main")
    libdir = os.path.join(ROOT, "cpuvoxelraycaster_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", os.path.join(ROOT, "tests", "cpp", "replay_main.cpp"),
                           "-o", exe, "-L" + libdir, "-l:libvrc_hip.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path / "nodes.bin"), str(depth), str(tmp_path / "top.rgb"), str(tmp_path / "side.rgb"),
                          str(W), str(H), str(spp), REPLAY], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"{len(ticks)} ticks loaded" in out.stdout
    frames = re.findall(r"frame (\d+) t=(\S+) fnv=([0-9a-f]+) rays=(\d+) steps=(\d+) hits=(\d+)", out.stdout)
    assert len(frames) == len(ticks)
    light = vrc.reference_light(depth)
    hit_frames = 0
    for i, ((t, x, y, z, vx, vy), fr) in enumerate(zip(ticks, frames)):
        assert float.fromhex(fr[1]) == t
        cam = O.make_camera((x, y, z), O.make_rotation(vx, vy), 1.0, 0.0, 1.0)
        acc = None
        rays = steps = 0
        for s in range(spp):
            # HipRayCaster keeps shadow_samples at the reference default (4 identical shadow rays in sample mode,
            # raycaster.hpp:147) and counts samples across frames (frame_index = i * spp + s)
            p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=0, frame_index=i * spp + s)
            _, acc, _, st = O.render_frame(nodes, depth, top, side, cam, p, accum=acc, threads=8)
            rays += st.rays
            steps += st.sum_complexity
        img = O.samples_to_image(acc)
        assert (int(fr[3]), int(fr[4])) == (rays, steps), f"tick {i}"
        assert int(fr[2], 16) == fnv1a(img.tobytes()), f"tick {i}"
        hit_frames += int(fr[5]) > W * H // 4
    assert hit_frames >= len(ticks) // 2        # the orbit looks at the terrain


def test_replay_through_frame_pipeline_equals_tick_by_tick(built, heights, textures, tmp_path):
    """vrc_host::HipFramePipeline (frames in flight: one renderer + stream per slot, whole-spp units, fused resolve) renders the
    replay frame for frame like the tick-by-tick loop -- same image hash and counters per frame, for 1, 2, 3 and 5 frames in
    flight --, and with the invariant-ray reuse the same images from fewer walks."""
    import cpuvoxelraycaster_amd as vrc
    depth, W, H, spp = 9, 161, 93, 3
    nodes = vrc.build_terrain_lsvo(heights, depth)
    top, side = textures
    nodes.tofile(tmp_path / "nodes.bin")
    top.tofile(tmp_path / "top.rgb")
    side.tofile(tmp_path / "side.rgb")
    ticks = [ln.split() for ln in open(REPLAY).read().splitlines() if ln.strip()]
    exe = str(tmp_path / "replay_main")
    libdir = os.path.join(ROOT, "cpuvoxelraycaster_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", os.path.join(ROOT, "tests", "cpp", "replay_main.cpp"),
                           "-o", exe, "-L" + libdir, "-l:libvrc_hip.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    base = [exe, str(tmp_path / "nodes.bin"), str(depth), str(tmp_path / "top.rgb"), str(tmp_path / "side.rgb"),
            str(W), str(H), str(spp), REPLAY]

    def frames(extra):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        return re.findall(r"frame (\d+) t=(\S+) fnv=([0-9a-f]+) rays=(\d+) steps=(\d+) hits=(\d+)", out.stdout)

    want = frames([])
    assert len(want) == len(ticks) and len({f[2] for f in want}) > len(ticks) // 2      # the orbit moves
    for F in (1, 2, 3, 5):
        assert frames([str(F)]) == want, F
    reuse = frames(["3", "reuse"])
    assert [f[:3] for f in reuse] == [f[:3] for f in want] and [f[5] for f in reuse] == [f[5] for f in want]
    assert all(int(a[3]) < int(b[3]) and int(a[4]) < int(b[4]) for a, b in zip(reuse, want) if int(b[5]))
