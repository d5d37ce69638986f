"""The C++ host adapter (HipLSVO / HipRayCaster, the reference's Volumetric /
RayCaster interface) compiled with plain g++ against the C ABI and run on the GPU:
autofocus ray and a GI frame must match the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_adapter_matches_oracle(built, heights, textures, tmp_path):
    import cpuvoxelraycaster_amd as vrc
    depth, W, H = 9, 160, 90
    nodes = vrc.build_terrain_lsvo(heights, depth)
    top, side = textures
    nodes.tofile(tmp_path / "nodes.bin")
    top.tofile(tmp_path / "top.rgb")
    side.tofile(tmp_path / "side.rgb")
    exe = str(tmp_path / "host_adapter_main")
    libdir = os.path.join(ROOT, "cpuvoxelraycaster_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "host_adapter_main.cpp"),
                           "-o", exe, "-L" + libdir, "-l:libvrc_hip.so", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, str(tmp_path / "nodes.bin"), str(depth), str(tmp_path / "top.rgb"),
                          str(tmp_path / "side.rgb"), str(W), str(H)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"autofocus hit=(\d) distance=(\S+) complexity=(\d+) normal=(\S+),(\S+),(\S+)", out.stdout)
    f = re.search(r"frame sum=(\d+) fnv=([0-9a-f]+) rays=(\d+) steps=(\d+)", out.stdout)
    assert m and f, out.stdout
    assert "sharded equal=1" in out.stdout, out.stdout      # setShard + resolveShard + vrc_unpack_shards == unsharded
    assert "sharded fused equal=1" in out.stdout, out.stdout   # ... and with renderFrameResolved (one launch per shard)
    # oracle: same autofocus ray, same frame
    S = float(1 << depth)
    rot = O.make_rotation(0.0, -0.5)
    org = np.array([[0.5 + 1, (S / 2 - 56) / S + 1, 0.5 + 1]], np.float32)
    cam_vec = np.array([[rot[2], rot[5], rot[8]]], np.float32)
    h = O.cast_rays(nodes, depth, org, cam_vec)[0]
    assert int(m.group(1)) == int(h["hit"] != 0)
    assert float.fromhex(m.group(2)) == float(h["distance"]) and int(m.group(3)) == int(h["complexity"])
    assert [float.fromhex(m.group(i)) for i in (4, 5, 6)] == [float(v) for v in h["normal"]]
    focal = float(np.float32(h["distance"]) * np.float32(S)) if h["hit"] else 100.0
    cam = O.make_camera((S / 2, S / 2 - 56, S / 2), rot, 1.0, 0.0, focal)
    light = np.array([-200.0 / 512 + 1, -1000.0 / 512 + 1, -300.0 / 512 + 1], np.float32)
    acc = None
    rays = steps = 0
    for s in range(2):
        # HipRayCaster::renderFrame leaves shadow_samples at the reference default (4 in sample mode)
        p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=0, frame_index=s)
        _, acc, _, st = O.render_frame(nodes, depth, top, side, cam, p, accum=acc, threads=8)
        rays += st.rays
        steps += st.sum_complexity
    img = O.samples_to_image(acc)
    assert int(f.group(1)) == int(img.sum(dtype=np.uint64))
    fnv = 1469598103934665603
    for b in img.tobytes():
        fnv = ((fnv ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert f.group(2) == f"{fnv:016x}"
    assert (int(f.group(3)), int(f.group(4))) == (rays, steps)
