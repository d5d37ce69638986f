"""Direct peer writes (round-2 verdict item 4): two processes on ONE GPU, each rendering its row blocks of the
frame; rank 1's frame kernel writes its rows straight into rank 0's framebuffer through an hipIpc mapping
(vrc_ipc_export_image / vrc_ipc_open_image / vrc_renderer_set_image_target), ordered by frame flags in shared memory.  Three frames in
flight, nine consecutive frames (every framebuffer reused twice): each composed frame must equal the unsharded render bit for
bit.  The multi-GPU form of the same code is bench.py --gather peer."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3, 8])
def test_peer_written_frames_equal_unsharded(built, world):
    port = _free_port()
    procs = []
    # world 8 = the node the metric is quoted on, rehearsed on one GPU: 1080 rows = 135 row blocks, seven ranks own 17, rank 7 owns 16
    size = dict(PW_W="1920", PW_H="1080") if world == 8 else {}
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", **size)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_write_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=40 if world < 8 else 180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            tails = [q.communicate()[1][-600:] for q in procs]
            raise AssertionError("peer-write workers timed out:\n" + "\n---\n".join(tails))
        outs.append((p.returncode, o, e))
    assert all(rc == 0 for rc, _, _ in outs), "\n".join(f"--- rank {k} rc {rc}\n{e[-2500:]}" for k, (rc, _, e) in enumerate(outs))
    line = [ln for ln in outs[0][1].splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["frames"] == 9 and d["frames_in_flight"] == 3 and d["world"] == world
    assert d["composed_frame_equals_unsharded"] == [True] * 9, d
    log = os.path.join(ROOT, "gpurun_out", "r04")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, f"peer_write_world{world}.json.log"), "w") as f:
        f.write(line + "\n")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_cpp_host_peer_writes(built, tmp_path, world):
    """the same exchange from a C++ host: tests/cpp/peer_write_main.cpp (vrc_host::HipFramePipeline + HipRayCaster::exportImage /
    setImageTarget + the frame flags), `world` forked processes on GPU 0"""
    exe = str(tmp_path / "peer_write_main")
    libdir = os.path.join(ROOT, "cpuvoxelraycaster_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "peer_write_main.cpp"), "-o", exe,
                           "-L" + libdir, "-l:libvrc_hip.so", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    data = os.path.join(libdir, "data")
    W, H = ("1920", "1080") if world == 8 else ("1280", "720")
    p = subprocess.Popen([exe, str(world), "8", W, H, "4", "9", os.path.join(data, "grass_top_16x16.rgb"),
                          os.path.join(data, "grass_side_16x16.rgb")], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=60 if world < 8 else 180)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, 9)                                    # the forked ranks are in the program's own process group
        raise AssertionError("peer_write_main timed out")
    assert p.returncode == 0, out + err
    d = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert d["composed_frames_equal_unsharded"] is True and d["world"] == world and d["frames"] == 9 and d["frames_in_flight"] == 3
    log = os.path.join(ROOT, "gpurun_out", "r04")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, f"cpp_peer_write_world{world}.json.log"), "w") as f:
        f.write(json.dumps(d) + "\n")


def test_a_dead_writer_ends_the_run_instead_of_hanging_it(built):
    """a stream-ordered flag wait has no timeout.  Rank 1 of 3 vanishes before its fifth frame, flags unwritten,
    with frames in flight on every rank: the watchdog (vrc_ipc_stream_wait via PeerFrames.wait) must notice the dead process,
    release every wait in every process and end the other ranks non-zero within seconds -- not at the test's kill timeout."""
    import time
    port = _free_port()
    procs = []
    t0 = time.monotonic()
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PW_DIE_RANK="1", PW_DIE_AT_FRAME="4", PW_FRAMES="12", PW_TIMEOUT_S="30")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_write_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the surviving ranks hung on the dead writer's flag")
        outs.append((p.returncode, o, e))
    assert [rc for rc, _, _ in outs] == [7, 5, 7], [(rc, e[-500:]) for rc, _, e in outs]
    for k in (0, 2):
        d = json.loads([ln for ln in outs[k][1].splitlines() if ln.startswith("{")][-1])
        assert "vrc error -5" in d["error"], d                       # VRC_ERR_PEER, with the reason
    assert time.monotonic() - t0 < 80
    log = os.path.join(ROOT, "gpurun_out", "r04")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, "peer_write_dead_writer.json.log"), "w") as f:
        f.write(json.dumps({"exit_codes": [rc for rc, _, _ in outs], "seconds": round(time.monotonic() - t0, 1),
                            "rank0": outs[0][1].strip().splitlines()[-1]}) + "\n")


def test_cpp_host_dead_writer(built, tmp_path):
    """the same failure from the C++ host (HipFramePipeline::waitAllWatched): a forked rank _exit()s mid-run"""
    exe = str(tmp_path / "peer_write_main")
    libdir = os.path.join(ROOT, "cpuvoxelraycaster_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "peer_write_main.cpp"), "-o", exe,
                           "-L" + libdir, "-l:libvrc_hip.so", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    data = os.path.join(libdir, "data")
    p = subprocess.Popen([exe, "3", "8", "1280", "720", "4", "12", os.path.join(data, "grass_top_16x16.rgb"), os.path.join(data, "grass_side_16x16.rgb")],
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PW_DIE_RANK="2", PW_DIE_AT_FRAME="5"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=150)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, 9)
        raise AssertionError("peer_write_main hung on the dead writer's flag")
    assert p.returncode == 6 and "vrc_ipc_stream_wait" in err, (p.returncode, err[-800:])


def test_a_rank_that_cannot_map_the_framebuffers_fails_every_rank_at_once(built):
    """round-5 advice: PeerFrames' set-up had a barrier behind the hipIpc opens, so a rank whose open failed (an IPC mode
    mismatch) left the others waiting for the process-group timeout and the run reported the hang, not the cause.  Now every
    rank reports its set-up and every rank raises the first failure by name: all three leave with the worker's exit code 8
    within seconds, and the message names the rank and the reason."""
    import time
    port = _free_port()
    procs = []
    t0 = time.monotonic()
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   PW_BAD_SETUP_RANK="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "peer_write_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("the other ranks hung on the rank whose set-up failed")
        outs.append((p.returncode, o, e))
    assert [rc for rc, _, _ in outs] == [8, 8, 8], [(rc, e[-500:]) for rc, _, e in outs]
    for rc, o, _ in outs:
        d = json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1])
        assert "could not be set up" in d["error"] and "rank 2" in d["error"] and "out of range" in d["error"], d
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("vrc_peer_test_%d" % port)]      # the segment's name is gone
