#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its configuration:

    Mrays/s (+ ms/frame), 512^3 LSVO terrain @ 1920x1080, 4 spp GI, N GPUs.

A "step" is one frame: every pixel, `spp` accumulated renderRay-equivalents
(primary + shadow + GI + GI-shadow rays, reference semantics: one indirect
bounce = camera hit + GI hit, raycaster.hpp:169-207) through the persistent HIP
frame kernel, the sample resolve, and -- for N > 1 -- the row-block shard pack,
the RCCL all-gather of the RGBA8 framebuffer and the unpack.  rays = traversals
actually executed (counted by the kernel, equal to the oracle's count).

Frames are independent, and a launch of the persistent frame kernel is mostly
tail at its end: the last work units finish at latency speed while most of the
chip idles (~4 % of a 1080p frame on one GPU, more on a multi-GPU shard).  So
three renderers / streams take turns by default (`--inflight 3`): the end of one
launch overlaps the start of the next frames, just as the exchange of frame i
overlaps the renders after it.  ms_per_step is then the steady-state frame PERIOD
(all K frames are complete inside the timed region).  Overlapping launches share
the chip, so the events around ONE launch (`roofline.kernel_ms_bracketed`, what
rocprofv3 --stats reports as the kernel's average duration for this command) span
about F frame periods; `roofline.kernel_ms`, the duration `achieved` is priced on,
is the chip time per launch: HIP events from the first timed launch's start to the
last one's end on the launch streams, divided by K.  At N = 1 the plain
one-stream loop (`--inflight 1`: each frame starts when the previous one has
ended, ms_per_step = a frame's latency) is measured right after the timed region
and reported as `extra.one_frame_in_flight`.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Both lines work for every N.  Started plainly (no WORLD_SIZE in the environment) with N > 1, this process is only the
fork/join (the one call of main.cpp:139-154, execute + waitExecutionDone): before anything that could touch the GPU is
imported it starts N fresh children, one rank per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, waits for them
and exits with the worst child's code (`launch_ranks`).  Rank 0 inherits stdout, so the JSON line stays the last line.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto 4 hardware queues by default; the frame streams, the default stream and (N > 1) RCCL's streams
# then share queues, and two frames "in flight" on one hardware queue run one after the other (measured: 1.35 instead of
# 1.30 ms per frame with four frame streams).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

CONFIGS = {
    # name: depth, width, height, spp, use_gi, aperture
    "c2": dict(depth=8, width=1280, height=720, spp=1, use_gi=0, aperture=0.0,
               label="256^3 LSVO terrain, 1280x720, primary + 1 shadow ray"),
    "c3": dict(depth=9, width=1920, height=1080, spp=4, use_gi=1, aperture=0.0,
               label="512^3 LSVO terrain (FastNoise seed 1337, generated and built on the GPU), 1920x1080, 4 spp GI "
                     "(1 indirect bounce), shadow_samples=1"),
    "c3b2": dict(depth=9, width=1920, height=1080, spp=4, use_gi=1, aperture=0.0, gi_bounces=2,
                 label="512^3 LSVO terrain, 1920x1080, 4 spp GI with the 2-bounce EXTENSION (recursive S2 at the GI hit; the "
                       "reference never reads gi_bounce, SURVEY App. C), shadow_samples=1"),
    "c3nogi": dict(depth=9, width=1920, height=1080, spp=4, use_gi=0, aperture=0.0,
                   label="512^3 LSVO terrain, 1920x1080, 4 spp, primary + 1 shadow ray (no GI)"),
    "c4": dict(depth=9, width=1920, height=1080, spp=4, use_gi=1, aperture=0.5,
               label="512^3 LSVO terrain, 1920x1080, 4 spp GI + DOF (aperture 0.5)"),
    "c5": dict(depth=10, width=3840, height=2160, spp=16, use_gi=1, aperture=0.0,
               label="1024^3 LSVO terrain, 3840x2160, 16 spp GI"),
}
ROW_BLOCK = 8
FORM_LABEL = {"root": "RCCL gather to rank 0", "all": "RCCL all-gather", "peer": "direct peer writes into rank 0's framebuffers (hipIpc), no collective"}
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


KERNEL_SOURCES = ["vrc_kernels.hip", "vrc_walk.h", "vrc_device.h", "vrc_internal.h"]


def kernel_source_hash():
    """Identifies the frame-kernel sources AND the compiler flags a PMC measurement belongs to (profiles/traffic.json is
    stamped with it)."""
    import hashlib
    from cpuvoxelraycaster_amd import build
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "cpuvoxelraycaster_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(build.FLAGS).encode())
    extra = os.environ.get("VRC_EXTRA_FLAGS", "").strip()       # experiment builds (tools/ab_variants.sh) never match the product's record
    if extra:
        h.update(("|" + extra).encode())
    variant = os.environ.get("VRC_LIB", "").strip()              # ... nor does an experiment library loaded in the product's place
    if variant:
        h.update(("|lib:" + variant).encode())
    return h.hexdigest()[:16]


def pmc_key(config, world, inflight, whole_spp_units):
    """The key of a workload's record in profiles/traffic.json: c3_n1 (frames in flight, the default command), c3_n1_inflight1
    (one frame at a time, the library's own unit policy), c3_n1_inflight1_wholespp (the default command's kernel symbol one
    launch at a time: `--inflight 1 --whole-spp 1 --fused 1 --lane-samples 1`)."""
    return f"{config}_n{world}" + ("" if inflight > 1 else "_inflight1" + ("_wholespp" if whole_spp_units else ""))


def pmc_record(key, kernel=None, fused=None, inflight=None):
    """The rocprofv3 PMC record of this workload from profiles/traffic.json (tools/pmc_roofline.sh +
    tools/update_traffic.py), or (None, reason) when there is none FOR THE KERNEL SOURCES OF THIS TREE, or when it was taken on
    another build of the frame kernel / other bench settings than this run's."""
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        rec = json.load(open(prof)).get("entries", {}).get(key)
    except Exception as e:                                       # noqa: BLE001
        return None, f"profiles/traffic.json unreadable: {e}"
    if rec is None:
        return None, f"no PMC record for {key} in profiles/traffic.json"
    cur = kernel_source_hash()
    if rec.get("source_hash") != cur:
        return None, (f"profiles/traffic.json[{key}] was measured on kernel sources {rec.get('source_hash')} (commit "
                      f"{rec.get('commit')}); this tree's are {cur}: re-run tools/pmc_roofline.sh + tools/update_traffic.py")
    if kernel and rec.get("kernel") and rec["kernel"] != kernel:
        return None, f"profiles/traffic.json[{key}] was measured on {rec['kernel']}, this run launches {kernel}"
    st = rec.get("settings") or {}
    if fused is not None and st.get("fused") is not None and bool(st["fused"]) != bool(fused):
        return None, f"profiles/traffic.json[{key}] was measured with fused resolve = {st['fused']}, this run has {fused}"
    if inflight is not None and st.get("frames_in_flight") is not None and int(st["frames_in_flight"]) != int(inflight):
        return None, f"profiles/traffic.json[{key}] was measured with {st['frames_in_flight']} frames in flight, this run has {inflight}"
    return rec, None


def cpu_baseline(cfg, nodes, textures, cam, light, budget_s=3.5):
    """The oracle (bit-identical CPU restatement, kind "port") timed on this box's host cores on a bounded sample of
    the same workload: whole frame-samples (every pixel once: 1/spp of a frame), three ways:
      4x4x16     the reference's own decomposition: 16 threads, thread t owns area (t % 4, t / 4), x-outer / y-inner
                 (main.cpp:90-92,139-152), parity build (-O2 -ffp-contract=off) -- on OpenMP threads and, where
                 oracle/_ref/libref_swarm.so exists, on the reference's own swarm thread pool
      omp        OpenMP dynamic rows, parity build, on team sizes around the container's CPU quota (cgroup v2 cpu.max or v1
                 cpu.cfs_quota_us: quota / 2, quota, 2 x, 4 x) or, where no quota is found, on 16 / 32 / 64 / half / all usable
                 host cores -- the fastest is the headline `value` (with its thread count as `cores`)
      native     the same source built -O3 -march=native with FMA contraction allowed: NON-parity (its results may
                 differ from the reference's; it is only timed), at the fastest omp thread count
    each for ~budget_s seconds and at least one whole frame-sample."""
    import ctypes as C
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    try:
        usable = len(os.sched_getaffinity(0))                    # the cores this process may run on (cgroup / affinity)
    except AttributeError:
        usable = os.cpu_count() or 1
    top, side = textures
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    W, H = cfg["width"], cfg["height"]
    L = O.lib()
    sig = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(O.Camera), C.POINTER(O.FrameParams),
           C.c_void_p, C.c_void_p]
    L.orc_render_frame_areas16.argtypes = sig + [C.POINTER(O.FrameStats)]
    native = None
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        native = C.CDLL(os.path.join(ROOT, "oracle", "_native", "libvrc_oracle_native.so"))
        native.orc_render_frame.argtypes = sig + [C.c_void_p, C.POINTER(O.FrameStats), C.c_int]
    except Exception:                                            # noqa: BLE001  (no compiler: the variant is left out)
        native = None
    import numpy as np
    img = np.zeros((H, W, 4), np.uint8)
    acc = np.zeros((H, W, 4), np.uint32)

    def cpu_seconds():
        t = os.times()
        return t.user + t.system                                 # of all threads of this process

    busy = {}                                                    # (kind, threads) -> CPU seconds per wall second while timed

    def run(kind, nthreads):
        rays = 0
        n = -1                                                   # the first pass is untimed: thread team start-up, first touches
        t0 = time.perf_counter()
        c0 = cpu_seconds()
        while n <= 0 or time.perf_counter() - t0 < budget_s:
            if n == 0:
                rays = 0
                t0 = time.perf_counter()
                c0 = cpu_seconds()
            p = O.make_params(W, H, light, use_gi=cfg["use_gi"], use_samples=1, shadow_samples=1, frame_index=n,
                              gi_bounces=cfg.get("gi_bounces", 1))
            st = O.FrameStats()
            args = (O._ptr(nodes), cfg["depth"], O._ptr(top), O._ptr(side), C.byref(ocam), C.byref(p), O._ptr(img), O._ptr(acc))
            if kind == "areas16":
                L.orc_render_frame_areas16(*args, C.byref(st))
            elif kind == "native":
                native.orc_render_frame(*args, None, C.byref(st), nthreads)
            else:
                L.orc_render_frame(*args, None, C.byref(st), nthreads)
            rays += st.rays
            n += 1
        dt = time.perf_counter() - t0
        busy[(kind, nthreads)] = round((cpu_seconds() - c0) / dt, 1)
        return rays, n, dt

    # a process's first second or so of multi-threaded work runs several times slower than the steady state on these
    # hosts (thread start-up, first touches, frequency): spend it before anything is timed
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 1.5:
        run_once = O.make_params(W, H, light, use_gi=cfg["use_gi"], use_samples=1, shadow_samples=1, gi_bounces=cfg.get("gi_bounces", 1))
        L.orc_render_frame(O._ptr(nodes), cfg["depth"], O._ptr(top), O._ptr(side), C.byref(ocam), C.byref(run_once), O._ptr(img),
                           O._ptr(acc), None, C.byref(O.FrameStats()), min(16, usable))
    out = []
    r, n, dt = run("areas16", 16)
    out.append({"decomp": "4x4x16", "threads": 16, "flags": "-O2 -ffp-contract=off", "parity": True,
                "value": round(r / dt / 1e6, 3), "frame_samples": n, "seconds": round(dt, 2)})
    if O.RefSwarm.available():
        # the same decomposition on the reference's OWN thread pool (lib/swarm/swarm.hpp compiled where it lies into
        # oracle/_ref/libref_swarm.so; the .so travels to the GPU box): Swarm(16), execute(lambda), waitExecutionDone()
        # exactly as main.cpp:90,139,154; the lambda's body is the oracle's per-area job
        sw = O.RefSwarm(16)
        rays_sw = 0
        n_sw = -1
        t0 = time.perf_counter()
        while n_sw <= 0 or time.perf_counter() - t0 < budget_s:
            if n_sw == 0:
                rays_sw = 0
                sw.retries = 0
                t0 = time.perf_counter()
            p = O.make_params(W, H, light, use_gi=cfg["use_gi"], use_samples=1, shadow_samples=1, frame_index=max(n_sw, 0),
                              gi_bounces=cfg.get("gi_bounces", 1))
            _, _, st = sw.render_frame(nodes, cfg["depth"], top, side, ocam, p, image=img, accum=acc)
            rays_sw += st.rays
            n_sw += 1
        dt_sw = time.perf_counter() - t0
        out.append({"decomp": "4x4x16 on the reference's own swarm pool (swrm::Swarm(16), lib/swarm/swarm.hpp)", "threads": 16,
                    "flags": "-O2 -ffp-contract=off", "parity": True, "value": round(rays_sw / dt_sw / 1e6, 3),
                    "frame_samples": n_sw, "seconds": round(dt_sw, 2), "dropped_executes_retried": sw.retries})
        sw.close()
    # OpenMP over a few thread counts: on a shared host more threads are not always faster -- the fastest is the headline
    # How many cores does this process really get?  The GPU boxes of this pool give a container a CFS quota
    # (/sys/fs/cgroup/cpu.max, e.g. "1600000 100000" = 16 CPUs' worth of time per period) on a host with 256 hardware threads:
    # os.sched_getaffinity() says 256, but a team of more threads than the quota only gets throttled -- which is why rounds 2-3
    # saw the rate PEAK at 16 threads and fall to a third at 256 (round-3 verdict item 8).  So the team sizes tried are those
    # around the quota, and each variant records the CPU seconds it got per wall second.  (Pinning threads -- OMP_PROC_BIND /
    # OMP_PLACES -- does not help under a quota, and binds the main thread to one core, which then is all that
    # sched_getaffinity() and the reference's swarm pool see: measured, 8.8 instead of 60 Mrays/s.)
    quota_cores, quota_from = cgroup_cpu_quota()
    cap = usable if not quota_cores else min(usable, max(1, int(round(quota_cores))))
    if quota_cores:
        cand = (cap // 2, cap, 2 * cap, 4 * cap)
    else:
        # no quota found (cgroup v1 without cfs limits, "max", or no cgroup files at all): nothing says where the rate peaks,
        # so the fixed team sizes of rounds 1-3 stay in the set beside the usable count
        cand = (16, 32, 64, usable // 2, usable)
    counts = sorted({c for c in cand if 1 <= c <= usable} or {usable})
    best = None
    for c in counts:
        r, n, dt = run("omp", c)
        v = {"decomp": "omp", "threads": c, "flags": "-O2 -ffp-contract=off", "parity": True,
             "value": round(r / dt / 1e6, 3), "frame_samples": n, "seconds": round(dt, 2), "rays": r,
             "cpu_seconds_per_second": busy[("omp", c)]}
        out.append(v)
        if best is None or v["value"] > best["value"]:
            best = v
    if native is not None:
        r2, n2, dt2 = run("native", best["threads"])
        out.append({"decomp": "omp", "threads": best["threads"], "flags": "-O3 -march=native (FMA contraction on)",
                    "parity": False, "value": round(r2 / dt2 / 1e6, 3), "frame_samples": n2, "seconds": round(dt2, 2)})
    # what the host gives this process: a cgroup CPU quota and the other tenants of the box decide how many of the `usable`
    # cores a thread team really gets -- cpu_seconds_per_second of a variant is the number of cores it was running on
    host = {"os_cpu_count": os.cpu_count(), "usable": usable, "cgroup_quota_cores": quota_cores, "cgroup_quota_from": quota_from,
            "scaling_note": ("the rate peaks at the container's CPU quota and falls beyond it (threads over the quota are throttled, not run): "
                             "cpu_seconds_per_second of each variant is the number of cores it actually ran on") if quota_cores else None}
    for name, path in (("cgroup_cpu_max", "/sys/fs/cgroup/cpu.max"), ("loadavg", "/proc/loadavg")):
        try:
            with open(path) as fh:
                host[name] = fh.read().strip()
        except OSError:
            host[name] = None
    host["omp_env"] = {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_NUM_THREADS")}
    return {"value": best["value"], "unit": "Mrays/s", "cores": best["threads"], "kind": "port", "host": host,
            "sample": f"{best['frame_samples']} whole frame-samples of this workload ({W}x{H}, every pixel once; a frame is {cfg['spp']} of "
                      f"them), {best['rays']} rays in {best['seconds']} s, oracle built -O2 -ffp-contract=off, OpenMP dynamic rows on "
                      f"{best['threads']} threads -- the fastest of {counts} (os.cpu_count() = {os.cpu_count()}, usable = {usable})",
            "variants": out}


def cgroup_cpu_quota(root="/sys/fs/cgroup"):
    """(cores, where it was read) of this container's CFS quota, or (None, None): cgroup v2's cpu.max ("<quota> <period>" or
    "max <period>"), else cgroup v1's cpu.cfs_quota_us / cpu.cfs_period_us (-1 = no limit)."""
    try:
        with open(os.path.join(root, "cpu.max")) as fh:
            q, per = fh.read().split()[:2]
        return (None, None) if q == "max" else (float(q) / float(per), "cgroup v2 cpu.max")
    except (OSError, ValueError):
        pass
    for d in ("cpu", "cpu,cpuacct", ""):
        try:
            with open(os.path.join(root, d, "cpu.cfs_quota_us")) as fh:
                q = float(fh.read().strip())
            with open(os.path.join(root, d, "cpu.cfs_period_us")) as fh:
                per = float(fh.read().strip())
            return (None, None) if q <= 0 or per <= 0 else (q / per, "cgroup v1 cpu.cfs_quota_us / cpu.cfs_period_us")
        except (OSError, ValueError):
            continue
    return None, None


def rank_environments(n, base_env, port):
    """The environment of each of the N ranks of one node -- what torch.distributed.run would set for them.

    HSA_ENABLE_IPC_MODE_LEGACY: whatever the launching environment says is passed on untouched.  Only where it says nothing
    do the ranks get "0" (dmabuf IPC), for EVERY exchange form: on this pool's host driver legacy-mode IPC handles do not
    exist, and without the variable both hipIpcGetMemHandle (--gather peer) and RCCL's own intra-node P2P set-up (--gather
    root / all) fail with "invalid argument" (the image exports it for that reason).
    VRC_KEEP_IPC_MODE=1 leaves an unset variable unset.  The value the ranks ran with is in the line: config.exchange."""
    envs = []
    for r in range(n):
        e = dict(base_env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "VRC_BENCH_LAUNCHED_BY": str(os.getpid())})
        if "HSA_ENABLE_IPC_MODE_LEGACY" not in base_env and base_env.get("VRC_KEEP_IPC_MODE") != "1":
            e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        envs.append(e)
    return envs


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def other_exchange_forms(headline, world, no_extras=False):
    """the exchange forms a run measures after its timed region (extra.exchange_forms), in order: the two that are not the
    headline form; none at N = 1 or with --no-extras"""
    return [] if world <= 1 or no_extras else [f for f in ("root", "all", "peer") if f != headline]


def launch_ranks(n, argv, popen=None, grace_s=20.0, plan_extra=None):
    """`python bench.py --gpus N` without a launcher: start N children (fresh processes -- never an exec of this one), one
    rank per GPU, and join them.  This process imports nothing that can initialise the GPU.  Rank 0 inherits stdout (its
    JSON line is the last line of ours); the other ranks' stdout goes to our stderr.  When a rank fails, the others get
    `grace_s` seconds to leave by themselves (a peer blocked in a collective would wait for ever) and are then ended BY
    PID.  Returns the worst exit code."""
    import signal
    import subprocess
    real = popen is None
    if real:                                                     # (a test passes its own popen from a process that has torch loaded)
        assert "torch" not in sys.modules, "the fork/join process must not have imported torch"
    popen = popen or subprocess.Popen
    port = int(os.environ.get("MASTER_PORT") or free_port())
    envs = rank_environments(n, os.environ, port)
    if os.environ.get("VRC_BENCH_LAUNCH_DRYRUN") == "1":         # tests/test_bench_launch.py: the plan, nothing started
        keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")
        print(json.dumps({"argv": [sys.executable, os.path.abspath(__file__)] + list(argv), "ranks": [{k: e.get(k) for k in keys} for e in envs],
                          "after_timed_region": plan_extra, "torch_imported": "torch" in sys.modules, "modules_touching_gpu": sorted(m for m in sys.modules if m.split(".")[0] in ("torch", "cpuvoxelraycaster_amd", "ctypes"))}))
        return 0
    procs = []
    for r, e in enumerate(envs):
        procs.append(popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                           stdout=(None if r == 0 else sys.stderr), stderr=None))

    def end_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)                            # this child's PID, nothing by pattern
                except OSError:
                    pass

    def on_signal(signum, _frame):
        end_all(signal.SIGTERM)
        sys.exit(128 + signum)
    for sg in (signal.SIGINT, signal.SIGTERM) if real else ():
        signal.signal(sg, on_signal)
    first_failure = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and first_failure is None:
            first_failure = time.monotonic()
            print(f"bench.py: rank {procs.index(bad[0])} exited with {bad[0].returncode}; the other ranks get {grace_s:.0f} s", file=sys.stderr, flush=True)
        if first_failure is not None and time.monotonic() - first_failure > grace_s:
            end_all(signal.SIGTERM)
            time.sleep(2.0)
            end_all(signal.SIGKILL)
        time.sleep(0.05)
    codes = [p.returncode for p in procs]
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr, flush=True)
    return min(worst, 255)


def oracle_frame(cfg, nodes, textures, cam, light, threads):
    """Frame 0 of the workload -- frame-samples 0..spp-1 of every pixel -- rendered by the oracle (the checker): the resolved
    image (samples_to_image, raycaster.hpp:94-103) and the ray / loop-iteration counts the timed renderer's frame 0 must equal."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    top, side = textures
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    acc, rays, sumc = None, 0, 0
    for s_ in range(cfg["spp"]):
        p = O.make_params(cfg["width"], cfg["height"], light, use_gi=cfg["use_gi"], use_samples=1, shadow_samples=1, frame_index=s_,
                          gi_bounces=cfg.get("gi_bounces", 1))
        _, acc, _, st = O.render_frame(nodes, cfg["depth"], top, side, ocam, p, accum=acc, threads=threads)
        rays += st.rays
        sumc += st.sum_complexity
    return O.samples_to_image(acc), rays, sumc


def exchange_facts(torch, dist, backend, world):
    """What the N > 1 exchange ran on, for the bench line: RCCL's version, the GPUs of the node and which of them can reach
    which directly (hipDeviceCanAccessPeer: xGMI links between the ranks' devices), and the IPC mode of the run."""
    facts = {"backend": backend, "world": world, "gpus_on_node": torch.cuda.device_count(),
             "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
             "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
    try:
        facts["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                                       # noqa: BLE001
        facts["rccl_version"] = f"unavailable: {e!r}"[:120]
    try:
        n = torch.cuda.device_count()
        facts["device_can_access_peer"] = [[(1 if i == j else int(torch.cuda.can_device_access_peer(i, j))) for j in range(n)] for i in range(n)]
        facts["device_names"] = sorted({torch.cuda.get_device_name(i) for i in range(n)})
    except Exception as e:                                       # noqa: BLE001
        facts["device_can_access_peer"] = f"unavailable: {e!r}"[:120]
    return facts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--pitch", type=float, default=-0.5)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of `steps` frames each (the median block is reported)")
    ap.add_argument("--blocks-min-seconds", type=float, default=8.0,
                    help="keep timing blocks until this much has been timed (max 4096 blocks): the timed region is one contiguous stretch of GPU "
                         "work several seconds long, so that an outside sampler of GPU activity sees it (round-3 verdict item 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-frame", action="store_true",
                    help="skip config.frame_equals_oracle (frame 0 of the timed renderers -- at N > 1 the gathered frame 0 -- against the oracle's)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight per GPU (renderers / streams used round-robin); 1 = one stream, each frame starts "
                         "when the previous one has ended; 0 = default: 3")
    ap.add_argument("--gather", default="root", choices=["all", "root", "peer"],
                    help="N > 1: how the RGBA8 row-block shards meet: 'root' (default) = RCCL gather to rank 0, the presenting rank "
                         "(north_star / SURVEY 8e 'RCCL gather of the RGBA8 framebuffer': every other rank sends its shard over one "
                         "direct xGMI hop, 1/N of the all-gather's bytes on the wire), 'all' = RCCL all-gather (every rank ends up "
                         "with the frame), 'peer' = no collective: every rank's frame kernel writes its rows straight into "
                         "rank 0's framebuffer (hipIpc mapping over xGMI, frame flags in shared memory; SURVEY 8e 'direct peer writes').  The two forms not "
                         "chosen are measured right after the timed region: extra.exchange_forms")
    ap.add_argument("--fused", default="auto", choices=["auto", "0", "1"],
                    help="resolve + pack + reset inside the frame kernel (vrc_render_frame_resolved) instead of a second kernel; "
                         "auto = whenever frames overlap (whole-spp work units: the fused path then resolves from registers, "
                         "never slower, 1 % faster on short shards; tools/shard_inflight.py, tools/resolve_cost.py)")
    ap.add_argument("--whole-spp", default="auto", choices=["auto", "0", "1"],
                    help="work unit of the frame kernel = a tile's WHOLE sample set (vrc_renderer_set_sample_chunk(spp)); auto = whenever "
                         "frames overlap (--inflight > 1).  `--inflight 1 --whole-spp 1 --fused 1 --lane-samples 1` runs the timed "
                         "region's kernel symbol one launch at a time: the form rocprofv3 --stats can time cleanly")
    ap.add_argument("--lane-samples", type=int, default=0, choices=[0, 1, 4],
                    help="lane <-> (pixel, sample) map of the frame kernel (vrc_renderer_set_lane_samples): 1 = 8 x 8 pixels per wave, "
                         "4 = 4 x 4 pixels x 4 samples abreast, 0 = the library's choice")
    ap.add_argument("--no-quad", action="store_true",
                    help="A/B: vrc_renderer_set_quad_walks(0) -- the pinhole kernel's sample-invariant walks in the plain 8 x 8 x 1 layout")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra measurements (2-bounce extension; N > 1: the other exchange forms)")
    ap.add_argument("--forms-deadline-s", type=float, default=240.0,
                    help="N > 1: the untimed block that measures the other exchange forms (extra.exchange_forms) may take this long; after it "
                         "rank 0 prints the headline line it already has and every rank leaves")
    ap.add_argument("--tuning", default="", help="blocks_per_cu: workgroups of the frame kernel per CU (= which waves-per-SIMD build), 0 = the library's choice")
    args = ap.parse_args()

    # the fork/join of main.cpp:139-154 across GPUs: plain `python bench.py --gpus N` starts its own ranks -- decided here,
    # before torch (or anything else that can touch the GPU) is imported
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], plan_extra={
            "headline_exchange": args.gather, "exchange_forms_measured_by_the_same_ranks": other_exchange_forms(args.gather, args.gpus, args.no_extras),
            "forms_deadline_s": args.forms_deadline_s, "processes_started_for_them": 0}))

    import numpy as np
    import torch
    import torch.distributed as dist

    import __graft_entry__ as g
    import cpuvoxelraycaster_amd as vrc
    import cpuvoxelraycaster_amd.dist as vdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}: they must agree")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # one process per GPU; VRC_DIST_BACKEND=gloo (+ ranks sharing a GPU) exists only to exercise the
    # N > 1 code path on a single-GPU box -- the real run is nccl (= RCCL over xGMI)
    backend = os.environ.get("VRC_DIST_BACKEND", "nccl")
    # VRC_FORCE_DIST=1 takes the N > 1 code path (process group, shard buffers, all-gather, unpack) with a world of one
    # rank: the only way to run the RCCL calls of this file on a single-GPU box
    dist_on = world > 1 or os.environ.get("VRC_FORCE_DIST") == "1"
    # ... and VRC_FORCE_FORMS=1 with it runs the N > 1 extras too (frame 0 through the exchange, extra.exchange_forms) with that
    # world of one: RCCL's gather <-> all-gather switch and the peer-write set-up on the one GPU a box has
    many = world > 1 or (dist_on and os.environ.get("VRC_FORCE_FORMS") == "1")
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > torch.cuda.device_count() and backend == "nccl":
        sys.exit(f"bench.py --gpus {world}: this node has {torch.cuda.device_count()} GPU(s) and RCCL needs one per rank "
                 "(VRC_DIST_BACKEND=gloo rehearses the N > 1 path with ranks sharing a GPU)")
    ranks_share_gpus = world > torch.cuda.device_count()
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        # the process group must span exactly the ranks the driver asked for
        assert dist.get_world_size() == max(args.gpus, 1) == world, (dist.get_world_size(), args.gpus, world)
    # the in-tree libraries travel with the snapshot; if they are missing or stale, ONE process per node builds
    # (N ranks writing the same .so at once would corrupt it) and the others wait
    if dist_on:
        if local_rank == 0:
            g.build()
        dist.barrier()
    else:
        g.build()
    if args.tuning:
        vrc.capi.check(vrc.capi.load().vrc_set_tuning(int(args.tuning.split(",")[0])))

    cfg = CONFIGS[args.config]
    depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
    textures = vrc.load_textures()
    # main.cpp:59-88 on the GPU: FastNoise heights -> LSVO, straight into HBM (csrc/vrc_build_gpu.hip);
    # the LNode[] is resident from here on
    # (a first, untimed build pays the one-time costs of the process: code-object load, first allocations)
    vrc.LSVO.fromFastNoiseTerrain(min(depth, 7), seed=1337, device=local_rank).close()
    t_build = time.perf_counter()
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, seed=1337, device=local_rank, textures=textures)
    build_call_ms = (time.perf_counter() - t_build) * 1e3        # whole call: allocations + noise + build + texture upload
    light = vrc.reference_light(depth)
    cam = vrc.reference_camera(depth, pitch=args.pitch, aperture=cfg["aperture"],
                               focal_length=(60.0 * (1 << depth) / 512.0 if cfg["aperture"] else 1.0))
    L = vrc.capi.load()

    # `--inflight F` frames in flight per GPU: frame i uses renderer / stream / buffers i % F.  With F > 1 the end of
    # one launch (its last work units finish at latency speed with most of the chip idle) overlaps the start of the
    # next frames' launches, like the exchange of frame i overlaps the renders after it.  Frames stay independent
    # and complete in order; F = 1 is the plain one-stream loop.
    F = args.inflight if args.inflight > 0 else 3
    # with frames overlapping, a launch's tail is covered by the next frame's launch: work units of ALL the samples of a tile
    # (one primary / shadow set-up per tile, no atomics, resolved from registers by the fused path) beat the short units the
    # library picks for a launch that has the chip to itself (tools/shard_inflight.py)
    whole_spp_units = F > 1 if args.whole_spp == "auto" else args.whole_spp == "1"
    fused = whole_spp_units if args.fused == "auto" else args.fused == "1"

    def make_renderer(whole=None):
        r = vrc.RayCaster(svo, (W, H))
        r.setLightPosition(light)
        r.use_gi, r.use_samples, r.shadow_samples = bool(cfg["use_gi"]), True, 1
        r.gi_bounces = cfg.get("gi_bounces", 1)
        if whole_spp_units if whole is None else whole:
            r.setSampleChunk(spp)
        if args.lane_samples:
            r.setLaneSamples(args.lane_samples)
        if args.no_quad:
            r.setQuadWalks(False)
        return r

    rcs = [make_renderer() for _ in range(F)]
    tstreams = [torch.cuda.current_stream()] if F == 1 else [torch.cuda.Stream() for _ in range(F)]
    shard_bytes = L.vrc_shard_bytes(W, H, ROW_BLOCK, world)
    if dist_on:
        # per slot: this rank's packed shard, the all-gathered shards, the composed frame
        mine = [torch.empty(shard_bytes, dtype=torch.uint8, device="cuda") for _ in range(2 if F == 1 else F + 2)]
        gathered = [torch.empty(world * shard_bytes, dtype=torch.uint8, device="cuda") for _ in range(2 if F == 1 else F + 2)]
        full = [torch.empty(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(2 if F == 1 else F + 2)]
    # more slots than frames in flight: a frame's all-gather is only waited for when its slot comes round again, so that
    # wait does not hold up the frame being issued (with frames in flight a frame's exchange only starts when its launch
    # has drained, i.e. while the following frames' launches run)
    n_slots = 2 if F == 1 else F + 2

    ev_pairs = []
    # ("base": the frame the current PeerFrames segment counts from -- its flags start at zero, the frame counter does not)
    state = {"i": 0, "pending": [None] * n_slots, "last_slot": 0, "gather": args.gather, "base": 0}
    peers = None
    if dist_on and args.gather == "peer":
        # rank 0's F renderers own the framebuffers; the other ranks' renderers write their rows into them
        # (cpuvoxelraycaster_amd/dist.py PeerFrames: hipIpc-mapped framebuffers, frame flags in shared memory; no collective)
        peers = vdist.PeerFrames(rcs, local_rank, rank, world, dist.group.WORLD, "/vrc_bench_peer_%s" % os.environ.get("MASTER_PORT", "0"))
        _PEER_EXCHANGE[:] = [peers, [ts.cuda_stream for ts in tstreams]]   # (a failing rank drains its streams before it leaves, see __main__)

    def finish_gather(slot):
        """wait (on the current stream) for the slot's outstanding all-gather and scatter it into the slot's frame"""
        if state["pending"][slot] is None:
            return
        work = state["pending"][slot]
        if work != "sync":
            work.wait()                                        # current stream waits for the collective
        state["pending"][slot] = None
        if state["gather"] == "root" and rank != 0:
            return                                             # only the presenting rank composes the frame
        vrc.capi.check(L.vrc_unpack_shards(vrc.capi.ptr(gathered[slot].data_ptr()), W, H, ROW_BLOCK, world,
                                           vrc.capi.ptr(full[slot].data_ptr()),
                                           vrc.capi.ptr(torch.cuda.current_stream().cuda_stream)))

    def step(timed, frame_override=None):
        # one frame = render (spp samples per pixel into the accumulators) + one pass that resolves this rank's
        # rows to RGBA8 (straight into the packed shard buffer for N > 1) and clears their accumulators for the
        # next frame (vrc_resolve_shard = samples_to_image + pack_shard + reset_samples)
        i = state["i"]
        rc, ts = rcs[i % F], tstreams[i % F]
        slot = i % n_slots
        with torch.cuda.stream(ts):
            stream = ts.cuda_stream
            pf = i - state["base"]                             # the frame's number in the peer-write flags (a multiple of F below i)
            if peers is not None:
                peers.before_render(pf, stream)                # rank 0 has consumed the frame this framebuffer held
            elif dist_on:
                finish_gather(slot)                            # the frame that used this slot's buffers last
            # (frame_override: which frame's samples are rendered -- the slot / flag sequence still follows i; the N > 1 parity bit
            # renders frame 0 again through the exchange after the timed region)
            rc.frame_index = (i if frame_override is None else frame_override) * spp
            use_ev = timed
            if use_ev:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(ts)
            if peers is not None and fused:
                rc.renderFrameResolved(cam, spp=spp, stream=stream, row_block=ROW_BLOCK, shard_index=rank, shard_count=world)
            elif dist_on and fused:
                rc.renderFrameResolved(cam, spp=spp, dst_ptr=mine[slot].data_ptr(), stream=stream, row_block=ROW_BLOCK, shard_index=rank, shard_count=world)
            elif dist_on:
                rc.renderFrame(cam, spp=spp, stream=stream, row_block=ROW_BLOCK, shard_index=rank, shard_count=world)
            elif fused:
                rc.renderFrameResolved(cam, spp=spp, stream=stream)
            else:
                rc.renderFrame(cam, spp=spp, stream=stream)
            if use_ev:
                e1.record(ts)
            if peers is not None:
                if not fused:
                    rc.resolveShard(ROW_BLOCK, rank, world, None, reset=True, stream=stream)
                peers.after_render(pf, stream)                 # rank 0: the frame is complete on this stream after the call
                peers.consumed_on(pf, stream)                  # (nothing reads it here: a presenter would, before this line)
                state["last_slot"] = i % F
            elif dist_on:
                if not fused:                                      # (fused: resolved, packed and reset by the frame kernel)
                    rc.resolveShard(ROW_BLOCK, rank, world, mine[slot].data_ptr(), reset=True, stream=stream)
                if backend == "nccl":
                    # RCCL over xGMI: RGBA8 row-block shards; enqueued behind the resolve, runs beside the next render.
                    # --gather all: all-gather; --gather root: rank 0 receives every shard straight into its place of
                    # the [shard][slot] buffer (one direct hop per peer), the other ranks only send
                    state["pending"][slot] = vdist.exchange_shards(mine[slot], gathered[slot], world, rank, state["gather"], async_op=True)
                else:
                    ts.synchronize()
                    parts = [torch.empty_like(mine[slot]) for _ in range(world)]
                    dist.all_gather(parts, mine[slot])
                    gathered[slot].copy_(torch.cat(parts))
                    state["pending"][slot] = "sync"
                state["last_slot"] = slot
            elif not fused:
                rc.resolveShard(reset=True, stream=stream)
        state["i"] += 1
        if use_ev:
            ev_pairs.append((e0, e1))

    def drain():
        if peers is not None:
            # streams that wait for flags are never synchronised blindly: a rank that died never writes its flag (watchdog:
            # cpuvoxelraycaster_amd/dist.py PeerFrames.wait -> vrc_ipc_stream_wait; raises when a peer is gone, after every
            # stream of this rank has drained with the release held)
            peers.wait([ts.cuda_stream for ts in tstreams])
        if dist_on and peers is None:
            torch.cuda.synchronize()                           # every frame stream has produced its shard
            for k in range(n_slots):
                finish_gather((state["i"] + k) % n_slots)      # oldest first; on the current stream
        torch.cuda.synchronize()

    for r in rcs:
        r.resetSamples(tstreams[0].cuda_stream)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(False)
    drain()
    for r in rcs:
        r.stats(reset=True, stream=tstreams[0].cuda_stream)
    # The timed region: exactly `steps` frames between a barrier + synchronize on both sides -- and that block repeated
    # (at least 5 times, and until --blocks-min-seconds, 8 s by default, have been timed) so that one run reports a spread: ms_per_step is the MEDIAN block,
    # extra.ms_per_step_blocks = [min, median, max] (round-2 verdict: a 26 ms window is one sample).
    blocks = []            # (elapsed seconds, host enqueue us per step, per-rank stats of the block, events of the block)
    n_blocks = max(1, args.blocks)
    while len(blocks) < n_blocks:
        for r in rcs:
            r.stats(reset=True, stream=tstreams[0].cuda_stream)
        del ev_pairs[:]
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        enq_us = (time.perf_counter() - t0) / args.steps * 1e6   # host time to issue a frame (no waiting on the GPU unless a slot is busy)
        t_dr = time.perf_counter()
        drain()                                                # the last frames' gathers are inside the timed region
        drain_ms = (time.perf_counter() - t_dr) * 1e3           # this rank's host wait for its own streams / exchanges to empty
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()
        el = el_local = time.perf_counter() - t0
        if dist_on:                                            # the slowest rank's clock decides, block by block
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        el_mine = time.perf_counter() - t0 if not dist_on else el_local
        blocks.append((el, enq_us, [r.stats(stream=tstreams[0].cuda_stream) for r in rcs], list(ev_pairs), drain_ms, el_mine))
        if len(blocks) == 1 and args.blocks_min_seconds > 0:   # from the first block's (rank-agreed) time: the same count on every rank
            n_blocks = min(4096, max(n_blocks, int(args.blocks_min_seconds / max(el, 1e-6)) + 1))
    order = sorted(range(len(blocks)), key=lambda k: blocks[k][0])
    med = order[len(order) // 2]                               # the median block is the one reported
    elapsed, host_enqueue_us, sts, med_events, med_drain_ms, med_elapsed_mine = blocks[med]
    del ev_pairs[:]
    ev_pairs.extend(med_events)
    block_ms = [b[0] / args.steps * 1e3 for b in blocks]
    stream = tstreams[0].cuda_stream

    class _Sum:
        rays = sum(x.rays for x in sts)
        sum_complexity = sum(x.sum_complexity for x in sts)
        pixels = sum(x.pixels for x in sts)
        iterations_not_executed = sum(x.iterations_not_executed for x in sts)
    st = _Sum
    # events around each launch: with one frame in flight that is the kernel's duration; with F launches sharing the chip
    # it spans ~F frame periods (and is what rocprofv3 --stats shows as the average duration), so the roofline prices the
    # kernel on the chip time per launch: first timed launch's start -> last launch's end (over the F streams) / K
    kernel_ms_bracketed = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(1, len(ev_pairs))
    if F > 1 and ev_pairs:
        kernel_ms = max(ev_pairs[0][0].elapsed_time(b) for _, b in ev_pairs[-F:]) / len(ev_pairs)
    else:
        kernel_ms = kernel_ms_bracketed

    rays, steps_sum, pixels = st.rays, st.sum_complexity, st.pixels
    if dist_on:
        c = torch.tensor([rays, steps_sum, pixels], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        rays, steps_sum, pixels = (int(v) for v in c.tolist())
        km = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        kernel_ms_max = float(km.item())
    else:
        kernel_ms_max = kernel_ms

    verified = None
    if dist_on and rank == 0:
        # not timed: the gathered frame must equal the unsharded render of the same samples
        ref = make_renderer()
        ref.frame_index = (state["i"] - 1) * spp                 # the last frame rendered
        ref.renderFrame(cam, spp=spp, stream=stream)
        ref.samples_to_image(stream)
        torch.cuda.synchronize()
        got = rcs[state["last_slot"]].readImage(stream) if peers is not None else full[state["last_slot"]].cpu().numpy().reshape(H, W, 4)
        verified = bool(np.array_equal(got, ref.readImage(stream)))

    # the parity bit at N > 1 (round-4 verdict, Next 2a): frame 0 once more THROUGH THE EXCHANGE -- every rank renders its rows of
    # frame 0 with the timed renderers and settings, the shards meet the timed way -- and rank 0 compares the gathered frame and
    # the ranks' summed counters with the oracle's frame 0 (below, where the line is put together)
    def frame0_through_exchange():
        for r in rcs:
            r.stats(reset=True, stream=tstreams[0].cuda_stream)
        dist.barrier()
        step(False, frame_override=0)
        drain()
        c0 = torch.tensor([sum(r.stats(stream=tstreams[0].cuda_stream).rays for r in rcs),
                           sum(r.stats(stream=tstreams[0].cuda_stream).sum_complexity for r in rcs)], dtype=torch.int64, device="cuda")
        dist.all_reduce(c0, op=dist.ReduceOp.SUM)
        got = None
        if rank == 0:
            got0 = rcs[state["last_slot"]].readImage(stream) if peers is not None else full[state["last_slot"]].cpu().numpy().reshape(H, W, 4)
            got = (got0.copy(), int(c0[0].item()), int(c0[1].item()))
        dist.barrier()                                         # (peer writes: nobody overwrites rank 0's framebuffer before it is read)
        return got

    frame0 = None
    if dist_on and many and not args.no_oracle_frame:
        frame0 = frame0_through_exchange()

    # N > 1: the chip time per launch above spans this rank's exchange / unpack kernels and waits as well: time the
    # frame kernel of this rank's shard on its own, untimed extra, same renderers / streams / unit policy, no exchange
    render_only_ms = None
    if dist_on:
        dist.barrier()                                         # rank 0 has compared the gathered frame (peer writes land in ITS framebuffers)
        n_ro = max(2 * F, min(args.steps, 30))
        evs = []
        for k in range(F + n_ro):
            rc, ts = rcs[k % F], tstreams[k % F]
            with torch.cuda.stream(ts):
                rc.frame_index = k * spp
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ts)
                if fused:
                    rc.renderFrameResolved(cam, spp=spp, dst_ptr=(None if peers is not None else mine[k % n_slots].data_ptr()), stream=ts.cuda_stream,
                                           row_block=ROW_BLOCK, shard_index=rank, shard_count=world)
                else:
                    rc.renderFrame(cam, spp=spp, stream=ts.cuda_stream, row_block=ROW_BLOCK, shard_index=rank, shard_count=world)
                    rc.resolveShard(ROW_BLOCK, rank, world, None if peers is not None else mine[k % n_slots].data_ptr(), reset=True, stream=ts.cuda_stream)
                e1.record(ts)
                if k >= F:
                    evs.append((e0, e1))
        torch.cuda.synchronize()
        render_only_ms = (max(evs[0][0].elapsed_time(b) for _, b in evs[-F:]) / len(evs)) if F > 1 else sum(a.elapsed_time(b) for a, b in evs) / len(evs)
        for r in rcs:
            r.stats(reset=True, stream=tstreams[0].cuda_stream)
        if peers is not None:                                  # those frames went to rank 0's framebuffers without flags:
            dist.barrier()                                     # nobody reads them, but keep the ranks together
    # N > 1: what each rank saw, so that a poor scaling curve explains itself (round-4 verdict, Next 2c)
    per_rank = None
    if dist_on:
        mine_diag = {"rank": rank, "device": local_rank, "kernel_ms": round(kernel_ms, 4),
                     "kernel_ms_render_only": round(render_only_ms, 4) if render_only_ms is not None else None,
                     "exchange_ms_per_step": (round(kernel_ms - render_only_ms, 4) if render_only_ms is not None else None),
                     "drain_wait_ms_per_block": round(med_drain_ms, 3), "block_ms_own_clock": round(med_elapsed_mine * 1e3, 3),
                     "host_enqueue_us_per_step": round(host_enqueue_us, 1),
                     "rays_per_frame": st.rays // args.steps, "pid": os.getpid()}
        gathered_diag = [None] * dist.get_world_size()
        dist.all_gather_object(gathered_diag, mine_diag)
        per_rank = gathered_diag
    # untimed extra, N = 1 only: BASELINE config 3's text says "GI 2 bounces"; the reference has ONE indirect bounce
    # (RayContext::gi_bounce is never read), which is what `value` measures -- the 2-bounce EXTENSION
    # (recursive getGlobalIllumination at the GI hit, defined by the oracle) is reported beside it
    extras = {}
    if rank == 0 and world == 1 and not dist_on and not args.no_extras and F > 1:
        # the same frames one at a time on one stream: a frame's latency, and the kernel's duration when it has the chip alone
        rc1 = make_renderer(whole=False)                        # the library's own unit policy for a launch alone on the chip
        ts = tstreams[0]
        n1 = max(10, min(args.steps, 50))
        runs = []                                               # (ms per step, kernel ms, rays per frame) of each block
        with torch.cuda.stream(ts):
            for i in range(3):                                  # untimed
                rc1.frame_index = i * spp
                rc1.renderFrame(cam, spp=spp, stream=ts.cuda_stream)
                rc1.resolveShard(reset=True, stream=ts.cuda_stream)
            for blk in range(max(3, args.blocks)):
                evs = []
                rc1.stats(reset=True, stream=ts.cuda_stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(n1):
                    rc1.frame_index = (3 + blk * n1 + i) * spp
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record(ts)
                    rc1.renderFrame(cam, spp=spp, stream=ts.cuda_stream)
                    eb.record(ts)
                    rc1.resolveShard(reset=True, stream=ts.cuda_stream)
                    evs.append((ea, eb))
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t1
                runs.append((dt1 / n1 * 1e3, sum(a.elapsed_time(b) for a, b in evs) / len(evs), rc1.stats(stream=ts.cuda_stream).rays / n1))
        runs.sort()
        m1 = runs[len(runs) // 2]
        extras["one_frame_in_flight"] = {
            "ms_per_step": round(m1[0], 4), "kernel_ms": round(m1[1], 4),
            "value": round(m1[2] / (m1[0] * 1e-3) / 1e6, 2), "unit": "Mrays/s", "steps": n1,
            "ms_per_step_blocks": [round(runs[0][0], 4), round(m1[0], 4), round(runs[-1][0], 4)], "blocks": len(runs),
            "kernel": rc1.lastKernel(),
            "note": "--inflight 1: one stream, each frame starts when the previous one has ended; the library's automatic work-unit "
                    "policy, resolve as a second kernel; median block of `blocks` (min / median / max in ms_per_step_blocks)"}
    if rank == 0 and world == 1 and not dist_on and not args.no_extras and F > 1:
        # the TIMED kernel symbol one launch at a time (round-4 verdict, Next 1): same unit policy, resolve and lane map as the
        # timed renderers, one stream, events around every launch -- the duration rocprofv3 --stats can reproduce
        # (`--inflight 1 --whole-spp 1 --fused 1 --lane-samples 1`, profiles/traffic.json[<config>_n1_inflight1_wholespp])
        rca = make_renderer()
        if not args.lane_samples:
            rca.setLaneSamples(1)                                 # (whole-spp units already make the library pick the 8 x 8 map)
        ts = tstreams[0]
        na = max(10, min(args.steps, 50))
        runs_a = []
        with torch.cuda.stream(ts):
            def frame_a(i):
                rca.frame_index = i * spp
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record(ts)
                if fused:
                    rca.renderFrameResolved(cam, spp=spp, stream=ts.cuda_stream)
                else:
                    rca.renderFrame(cam, spp=spp, stream=ts.cuda_stream)
                eb.record(ts)
                if not fused:
                    rca.resolveShard(reset=True, stream=ts.cuda_stream)
                return ea, eb
            for i in range(3):
                frame_a(i)
            for blk in range(max(3, args.blocks)):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                evs = [frame_a(3 + blk * na + i) for i in range(na)]
                torch.cuda.synchronize()
                runs_a.append(((time.perf_counter() - t1) / na * 1e3, sum(a.elapsed_time(b) for a, b in evs) / len(evs)))
        runs_a.sort(key=lambda r: r[1])
        ma = runs_a[len(runs_a) // 2]
        extras["timed_symbol_alone"] = {
            "kernel": rca.lastKernel(), "kernel_ms": round(ma[1], 4), "ms_per_step": round(ma[0], 4), "steps": na, "blocks": len(runs_a),
            "kernel_ms_blocks": [round(runs_a[0][1], 4), round(ma[1], 4), round(runs_a[-1][1], 4)],
            "note": "the timed region's kernel symbol and unit policy, one launch at a time on one stream (HIP events around each launch, "
                    "median block); this is what `rocprofv3 --stats -- python3 bench.py --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1` "
                    "reports as the symbol's average duration"}

    if rank == 0 and world == 1 and not dist_on and not args.no_extras and args.config == "c3":
        # the same way as the timed region: F renderers / streams round-robin, same unit policy, same resolve
        rcs2 = [make_renderer() for _ in range(F)]
        for r2 in rcs2:
            r2.gi_bounces = 2
        n2 = max(10, min(args.steps, 50))

        def frames2(i0, n):
            for i in range(i0, i0 + n):
                r2, ts2 = rcs2[i % F], tstreams[i % F]
                r2.frame_index = i * spp
                if fused:
                    r2.renderFrameResolved(cam, spp=spp, stream=ts2.cuda_stream)
                else:
                    r2.renderFrame(cam, spp=spp, stream=ts2.cuda_stream)
                    r2.resolveShard(reset=True, stream=ts2.cuda_stream)
        frames2(0, 2 * F)
        torch.cuda.synchronize()
        for r2 in rcs2:
            r2.stats(reset=True, stream=tstreams[0].cuda_stream)
        t2 = time.perf_counter()
        frames2(2 * F, n2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2

        class st2:
            rays = sum(r2.stats(stream=tstreams[0].cuda_stream).rays for r2 in rcs2)
            sum_complexity = sum(r2.stats(stream=tstreams[0].cuda_stream).sum_complexity for r2 in rcs2)
        extras["gi_2_bounce_extension"] = {
            "value": round(st2.rays / dt2 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt2 / n2 * 1e3, 4), "steps": n2,
            "rays_per_frame": st2.rays // n2, "loop_iterations_per_frame": st2.sum_complexity // n2,
            "note": "extension without reference behaviour: bounce 2 = getGlobalIllumination applied again at the GI hit "
                    "(oracle/vrc_oracle.c global_illumination, level 1); same frame, same frames in flight and unit policy as the timed "
                    "region, measured right after it"}

    if rank == 0 and world == 1 and not dist_on and not args.no_extras and cfg["aperture"] == 0.0 and cfg.get("gi_bounces", 1) <= 1:
        # untimed extra, the other way round: the same frames with every ray started at the root as the reference's loop does
        # (vrc_renderer_set_walk_from_root) instead of below it, next to the previous hit / the camera -- what the start below
        # the root is worth; same image, same rays, same sum of complexities
        rcs4 = [make_renderer() for _ in range(F)]
        for r4 in rcs4:
            r4.setWalkFromRoot(True)
        n4 = max(10, min(args.steps, 50))

        def frames4(i0, n):
            for i in range(i0, i0 + n):
                r4, ts4 = rcs4[i % F], tstreams[i % F]
                r4.frame_index = i * spp
                if fused:
                    r4.renderFrameResolved(cam, spp=spp, stream=ts4.cuda_stream)
                else:
                    r4.renderFrame(cam, spp=spp, stream=ts4.cuda_stream)
                    r4.resolveShard(reset=True, stream=ts4.cuda_stream)
        frames4(0, 2 * F)
        torch.cuda.synchronize()
        for r4 in rcs4:
            r4.stats(reset=True, stream=tstreams[0].cuda_stream)
        t4 = time.perf_counter()
        frames4(2 * F, n4)
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t4
        sts4 = [r4.stats(stream=tstreams[0].cuda_stream) for r4 in rcs4]
        extras["every_ray_from_the_root"] = {
            "ms_per_step": round(dt4 / n4 * 1e3, 4), "value": round(sum(x.rays for x in sts4) / dt4 / 1e6, 2), "unit": "Mrays/s", "steps": n4,
            "loop_iterations_not_executed": int(sum(x.iterations_not_executed for x in sts4)),
            "note": "measurement switch (vrc_renderer_set_walk_from_root): no ray starts below the root; same image, rays and "
                    "complexity sums as the timed region, same frames in flight and unit policy, measured right after it"}

    if rank == 0 and world == 1 and not dist_on and not args.no_extras and cfg["aperture"] == 0.0:
        # untimed extra, beyond the reference: the same frames with vrc_renderer_set_invariant_ray_reuse -- a pinhole camera's
        # primary ray and the shadow ray of its hit are the same for every sample, the reference (and `value`) walk them
        # once per sample, here once per pixel and frame.  Same image bit for bit (tests/test_gpu_reuse.py); its Mrays/s
        # counts the walks EXECUTED, so ms_per_step is the figure to compare
        rcs3 = [make_renderer() for _ in range(F)]
        for r3 in rcs3:
            r3.setInvariantRayReuse(True)
        n3 = max(10, min(args.steps, 50))

        def frames3(i0, n):
            for i in range(i0, i0 + n):
                r3, ts3 = rcs3[i % F], tstreams[i % F]
                r3.frame_index = i * spp
                if fused:
                    r3.renderFrameResolved(cam, spp=spp, stream=ts3.cuda_stream)
                else:
                    r3.renderFrame(cam, spp=spp, stream=ts3.cuda_stream)
                    r3.resolveShard(reset=True, stream=ts3.cuda_stream)
        frames3(0, 2 * F)
        torch.cuda.synchronize()
        for r3 in rcs3:
            r3.stats(reset=True, stream=tstreams[0].cuda_stream)
        t3 = time.perf_counter()
        frames3(2 * F, n3)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t3
        rays3 = sum(r3.stats(stream=tstreams[0].cuda_stream).rays for r3 in rcs3)
        extras["invariant_rays_walked_once"] = {
            "ms_per_step": round(dt3 / n3 * 1e3, 4), "value": round(rays3 / dt3 / 1e6, 2), "unit": "Mrays/s (walks executed)", "steps": n3,
            "rays_per_frame_executed": rays3 // n3, "rays_per_frame_reference": rays // args.steps,
            "note": "beyond the reference, off by default: pinhole camera, the sample-invariant primary and shadow walks done once per "
                    "pixel and frame instead of once per sample; same image bit for bit; same frames in flight and unit policy as the "
                    "timed region (whole-spp units make the reuse span the frame's samples), measured right after it"}

    out = oracle0 = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # roofline of the dominant kernel (k_render_sync*), this rank's launches:
        # algorithmic bytes per launch = 8 B per traversal-loop iteration (one LNode, lsvo.hpp:74)
        # + one 32 B accumulator read-modify-write per pixel of the launch (raycaster.hpp:87-90).  Only the iterations the kernel
        # EXECUTES are priced: a ray next to the previous hit starts below the root, and the descends from the root it leaves out
        # (counted in sum_complexity, as the reference counts them) fetch nothing
        my_steps_per_launch = (st.sum_complexity - st.iterations_not_executed) / args.steps
        my_pixels_per_launch = st.pixels / args.steps / spp
        alg_bytes = 8.0 * my_steps_per_launch + 32.0 * my_pixels_per_launch
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # measured HBM / fabric bytes and instruction-issue numbers of the same workload come from the rocprofv3 PMC
        # passes recorded in profiles/traffic.json -- only when that record was taken on THIS tree's kernel sources
        # (the stage-synchronous kernel has a specialisation for the pinhole camera, see csrc/vrc_kernels.hip launch_render)
        sync_kernel_name = rcs[0].lastKernel()       # the build the library chose for this kind of launch (vrc_renderer_last_kernel)
        rec, why = pmc_record(pmc_key(args.config, world, F, whole_spp_units), kernel=sync_kernel_name, fused=fused, inflight=F)
        traffic = rec["traffic_bytes"] if rec else None
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "kernel": sync_kernel_name, "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_is": ("events around each launch, averaged (launches do not overlap)" if F == 1 else
                                 f"chip time per launch: HIP events from the first timed launch's start to the last one's end on the {F} "
                                 "launch streams / steps" + (" (includes this rank's resolve / exchange kernels)" if dist_on else "")
                                 + "; kernel_ms_bracketed = events around one launch, which overlaps its neighbours "
                                 "(= the average duration rocprofv3 --stats reports for this command)"),
                "kernel_ms_bracketed": round(kernel_ms_bracketed, 4),
                "kernel_ms_max_rank": round(kernel_ms_max, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
                "loop_iterations_per_launch": {"counted": int(st.sum_complexity / args.steps),
                                               "executed": int(my_steps_per_launch),
                                               "note": "counted = sum of HitPoint::complexity (lsvo.hpp:73), equal to the reference's; executed = "
                                                       "without the descends from the root that rays starting next to a hit leave out"},
                "kernel_ms_render_only": (round(render_only_ms, 4) if render_only_ms is not None else None),
                "achieved_is": "ALGORITHMIC bytes (SURVEY 8d: 8 B per walk-loop iteration EXECUTED + 32 B per pixel) / kernel time -- nominal by "
                               "construction: the tree is L1/L2-resident, see traffic_frac_of_peak, and the binding limit is VALU "
                               "instruction issue, see issue"}
        if "timed_symbol_alone" in extras:
            ka = extras["timed_symbol_alone"]["kernel_ms"]
            roof["kernel_ms_alone"] = ka
            roof["frac_alone"] = round(alg_bytes / (ka * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)
            roof["kernel_ms_alone_is"] = ("the same kernel symbol, work units and resolve as the timed region, ONE launch at a time (HIP events around "
                                          "each launch, measured in this run right after the timed region: extra.timed_symbol_alone); frac_alone = "
                                          "algorithmic_bytes_per_launch / kernel_ms_alone / peak.  kernel_ms / frac are per frame PERIOD with "
                                          f"{F} launches sharing the chip")
            rec_a, why_a = pmc_record(pmc_key(args.config, world, 1, True), kernel=extras["timed_symbol_alone"]["kernel"], fused=fused, inflight=1)
            if rec_a and rec_a.get("profiled_kernel_ms"):
                pa = rec_a["profiled_kernel_ms"]
                roof["kernel_ms_alone_rocprof"] = {
                    "average_ms": pa, "profile": rec_a["profile"].replace("_pmc.txt", "_kernel_stats.csv"), "pmc": rec_a["profile"],
                    "frac_alone": round(alg_bytes / (pa * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                    "agrees_with_kernel_ms_alone_within_3pct": bool(abs(pa / ka - 1.0) <= 0.03),
                    "note": "rocprofv3 --kernel-trace --stats average duration of this symbol under `bench.py --inflight 1 --whole-spp 1 --fused 1 "
                            "--lane-samples 1` on these kernel sources (stamped by hash in profiles/traffic.json)"}
            else:
                roof["kernel_ms_alone_rocprof"] = {"average_ms": None, "note": why_a}
        if rec:
            roof["traffic_frac_of_peak"] = round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)
            roof["traffic_detail"] = {k: rec[k] for k in ("fetch_bytes", "write_bytes", "profile", "commit", "profiled_kernel_ms") if k in rec}
            roof["issue"] = {"valu_wave_instr_per_launch": rec.get("valu_wave_instr"),
                             "lane_utilisation": rec.get("lane_utilisation"),
                             "simd_cycles_per_valu_instr": rec.get("simd_cycles_per_valu_instr"),
                             "simd_cycles_per_valu_instr_timed": (round(kernel_ms * 1e-3 * 2.4e9 * 1024.0 / rec["valu_wave_instr"], 3)
                                                                  if rec.get("valu_wave_instr") else None),
                             "note": "rocprofv3 PMC, per launch (the profiler runs the launches one at a time): SQ_INSTS_VALU; "
                                     "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); GRBM_GUI_ACTIVE / 8 x 1024 SIMDs / SQ_INSTS_VALU; "
                                     "..._timed = this run's kernel_ms x 2.4 GHz x 1024 SIMDs / SQ_INSTS_VALU (launches overlapping as timed)"}
        else:
            roof["traffic_note"] = why
        # The binding limit is instruction issue, not bandwidth: price the launch against the issue floor as well
        # (tools/issue_model.py: instruction classes of the walk loop x block issue frequencies; counts from the PMC record
        # where there is one).  `bound` names it; frac / achieved / peak keep the HBM form SURVEY 8d asks for.
        issue = roof.setdefault("issue", {})
        try:
            model = json.load(open(os.path.join(ROOT, "profiles", "issue_model.json")))
        except Exception as e:                                   # noqa: BLE001
            model = None
            issue["floor_note"] = f"profiles/issue_model.json unreadable: {e}"
        if model is not None and model.get("source_hash") != kernel_source_hash():
            issue["floor_note"] = (f"profiles/issue_model.json was made from kernel sources {model.get('source_hash')}, this tree's are "
                                   f"{kernel_source_hash()}: re-run tools/issue_model.py")
            model = None
        walks_model = model is not None and args.config == "c3" and abs(args.pitch + 0.5) < 1e-9
        if model is not None and not walks_model:
            issue["floor_note"] = ("block issue frequencies of the walk exist for the c3 workload at pitch -0.5 only (tests/tools/path_schedule.py "
                                   "--dump): the slow-class resource is left out of this floor")
        if model is not None and (walks_model or (rec and rec.get("valu_wave_instr"))):
            cyc = model["cycles"]
            tile_samples = my_pixels_per_launch / 64.0 * spp      # 8 x 8 pixel tiles x samples of this rank's launch
            w = model["walks_per_tile_sample"] if walks_model else {}
            walk_valu = sum(w.get(c, 0.0) for c in "ABSQP") * tile_samples
            slow = w.get("Q", 0.0) * tile_samples
            pcls = w.get("P", 0.0) * tile_samples
            valu = float(rec["valu_wave_instr"]) if rec and rec.get("valu_wave_instr") else walk_valu
            scalar = (float(rec["sq_insts_salu"] + rec.get("sq_insts_branch", 0)) if rec and rec.get("sq_insts_salu")
                      else (w.get("salu", 0.0) + w.get("branch", 0.0)) * tile_samples)
            simds, cus, hz = 1024.0, 256.0, 2.4e9
            res = {"valu_issue": cyc["valu_arch"] * valu / simds, "scalar_unit": cyc["salu_per_cu"] * scalar / cus}
            if walks_model:
                res["slow_class_unit"] = (cyc["slow"] * slow + cyc["p"] * pcls) / simds
            binding = max(res, key=res.get)
            floor_ms = res[binding] / hz * 1e3
            issue.update({
                "floor_ms": round(floor_ms, 4), "issue_frac": round(floor_ms / kernel_ms, 4), "binding_resource": binding,
                "floor_ms_by_resource": {k: round(v / hz * 1e3, 4) for k, v in res.items()},
                "floor_ms_at_measured_ceiling": round(cyc["valu_measured_ceiling"] * valu / simds / hz * 1e3, 4),
                "valu_wave_instr_used": int(valu), "valu_count_from": "rocprofv3 SQ_INSTS_VALU (profiles/traffic.json)" if rec and rec.get("valu_wave_instr") else "model (walks only: a lower bound)",
                "floor_is": f"max over resources of the cycles below which the launch's instruction streams cannot issue: {cyc['valu_arch']} cycles per VALU "
                            f"wave-instruction per SIMD (a wave64 instruction is two passes of the SIMD-32, MI355X_MICROARCH.md), {cyc['slow']} per "
                            f"slow-class instruction (compares, min / max, shift-or ..., tools/micro/op_issue.hip), {cyc['salu_per_cu']} per scalar / branch "
                            "instruction of a CU's one scalar unit; 1024 SIMDs, 256 CUs, 2.4 GHz.  floor_ms_at_measured_ceiling prices VALU issue at "
                            f"{cyc['valu_measured_ceiling']} cycles, the fastest any plain-VALU stream issued in the microbenchmarks (profiles/r03/op_pairs.txt) -- "
                            "the kernel's own streams issue at about that rate, i.e. only fewer instructions (or fuller lanes, see lane_utilisation) make the frame faster"})
            if walks_model:
                issue.update({"valu_wave_instr_walks_model": int(walk_valu), "slow_class_wave_instr_walks_model": int(slow)})
                if model.get("lane_utilisation_by_stage"):
                    issue["lane_utilisation_by_stage"] = dict(model["lane_utilisation_by_stage"])
                    issue["lane_utilisation_by_stage_note"] = (
                        "live lanes per issued walk-loop header, stage by stage, from the oracle's per-iteration traces of this workload "
                        "(profiles/issue_model.json <- " + model["frequencies"] + "); lane_utilisation above is the PMC average over the whole "
                        "launch.  With the pinhole camera the reference casts a pixel's primary ray and the shadow ray of its hit once per "
                        "sample (raycaster.hpp:131,147-158; the jitter is commented out at :151): in the quadrant layout those identical walks sit "
                        f"side by side, so (spp - 1) / spp = {spp - 1}/{spp} of the primary / shadow stages' live lanes are lockstep copies -- see "
                        "config.rays_distinct_per_frame.  The GI stages' rays are all distinct")
            roof["bound"] = "valu_issue"
            roof["bound_note"] = ("instruction issue (roofline.issue) is what binds; achieved / peak / frac keep the nominal HBM form SURVEY 8d "
                                  "asks for (algorithmic bytes / kernel time), traffic is the measured HBM traffic")
        out = {
            "metric": "Mrays/s + ms/frame, 512^3 LSVO @1920x1080 4spp GI, 1/2/4/8 GPU",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": cfg["label"], "rays_per_frame": rays // args.steps,
                       "loop_iterations_per_frame": steps_sum // args.steps,
                       "camera_pitch": args.pitch, "lsvo_nodes": svo.n_nodes,
                       "lsvo_gpu_build_ms": round(svo.build_ms, 3), "lsvo_gpu_build_call_ms": round(build_call_ms, 3),
                       "ms_per_step_is": (f"frame PERIOD with {F} frames in flight (all `steps` frames complete inside the timed region); the LATENCY of one "
                                          "frame, rendered alone, is extra.one_frame_in_flight.ms_per_step" if F > 1 else
                                          "frame latency = period: one frame at a time on one stream"),
                       "frames_in_flight": F, "whole_spp_work_units": whole_spp_units, "resolve_fused_into_frame_kernel": fused, "host_enqueue_us_per_step": round(host_enqueue_us, 1), "sharding": f"{ROW_BLOCK}-row blocks round-robin over {world} GPU(s)"
                       + ((", direct peer writes into rank 0's framebuffers (hipIpc), no collective" if args.gather == "peer" else
                          (f", RCCL {'gather to rank 0' if args.gather == 'root' else 'all-gather'} of RGBA8 shards" if backend == "nccl" else
                           f", {backend} all_gather of RGBA8 shards staged through the host (REHEARSAL of the N > 1 control flow"
                           + (": the ranks share a GPU" if ranks_share_gpus else "") + "; the real exchange is RCCL)")) if dist_on else ""),
                       "ranks_in_process_group": (dist.get_world_size() if dist_on else 1),
                       "launched_by": ("bench.py itself (launch_ranks)" if os.environ.get("VRC_BENCH_LAUNCHED_BY") else
                                       "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "plain"),
                       "distinct_gpus": min(world, torch.cuda.device_count()),
                       "library": (os.environ.get("VRC_LIB") or "cpuvoxelraycaster_amd/libvrc_hip.so (the product)"),
                       "exchange_backend": (backend if dist_on else None),
                       "exchange": exchange_facts(torch, dist, backend, world) if dist_on else None,
                       "per_rank": per_rank,
                       "gathered_frame_equals_unsharded": verified,
                       "frame_equals_oracle": None},
            "roofline": roof,
        }
        extras["ms_per_step_blocks"] = {"min_median_max": [round(min(block_ms), 4), round(sorted(block_ms)[len(block_ms) // 2], 4), round(max(block_ms), 4)],
                                        "blocks": len(block_ms), "steps_per_block": args.steps,
                                        "note": "every block is `steps` frames between barrier + synchronize on both sides; ms_per_step, value and the "
                                                "roofline are those of the median block"}
        if "invariant_rays_walked_once" in extras:
            # which of the work is distinct (round-5 verdict, Next 3): `value` counts every ray the reference casts; a pinhole
            # camera's primary and shadow ray are the same for every sample of a pixel, so part of them are identical walks
            inv = extras["invariant_rays_walked_once"]
            out["config"]["rays_distinct_per_frame"] = inv["rays_per_frame_executed"]
            out["config"]["ms_per_identical_image"] = inv["ms_per_step"]
            out["config"]["rays_distinct_note"] = (
                f"of the {rays // args.steps} rays per frame the reference casts (and `value` counts, each walked by a lane of its own), "
                f"{rays // args.steps - inv['rays_per_frame_executed']} are repeats of a pixel's sample-invariant primary / shadow walk "
                "(raycaster.hpp:131,147-158); rays_distinct_per_frame = the walks left when each is done once per pixel and frame, "
                "ms_per_identical_image = the frame period of that mode (vrc_renderer_set_invariant_ray_reuse, off by default, bit-identical "
                "image; measured right after the timed region: extra.invariant_rays_walked_once)")
        if extras:
            out["extra"] = extras
        if frame0 is not None:
            # the parity bit at N > 1: the GATHERED frame 0 and the ranks' summed counters against the oracle's frame 0
            try:
                usable_cores = len(os.sched_getaffinity(0))
            except AttributeError:
                usable_cores = os.cpu_count() or 1
            t_or = time.perf_counter()
            ref_img, ref_rays, ref_sumc = oracle_frame(cfg, svo.downloadNodes(), textures, cam, light, min(64, usable_cores))
            got0, rays0, sumc0 = frame0
            frame_ok = bool(np.array_equal(got0, ref_img)) and (rays0, sumc0) == (ref_rays, ref_sumc)
            oracle0 = (ref_img, ref_rays, ref_sumc)                # (the other exchange forms' gathered frames meet the same one)
            out["config"]["frame_equals_oracle"] = frame_ok
            out["config"]["frame_equals_oracle_is"] = (
                f"frame 0 ({spp} samples of all {W * H} pixels) rendered once more after the timed region by the {world} ranks' timed renderers "
                f"({sync_kernel_name}), each its own row blocks, gathered the timed way ({out['config']['sharding']}) vs oracle/vrc_oracle.c: the "
                f"gathered RGBA8 image bit for bit, rays summed over ranks {rays0} vs {ref_rays}, loop iterations {sumc0} vs {ref_sumc}; oracle time "
                f"{time.perf_counter() - t_or:.1f} s on rank 0")
        if world == 1 and not args.no_cpu_baseline and not args.no_oracle_frame:
            # the parity bit of THIS run (round-3 verdict item 4): frame 0 once more on a renderer set up like the timed ones
            # (same unit policy, same resolve, same lane map) against the oracle's frame 0 -- image, rays, loop iterations
            host_nodes = svo.downloadNodes()
            chk = make_renderer()
            chk.frame_index = 0
            if fused:
                chk.renderFrameResolved(cam, spp=spp, stream=stream)
            else:
                chk.renderFrame(cam, spp=spp, stream=stream)
                chk.resolveShard(reset=True, stream=stream)
            torch.cuda.synchronize()
            got_img, got_st = chk.readImage(stream), chk.stats(stream=stream)
            try:
                usable_cores = len(os.sched_getaffinity(0))
            except AttributeError:
                usable_cores = os.cpu_count() or 1
            t_or = time.perf_counter()
            ref_img, ref_rays, ref_sumc = oracle_frame(cfg, host_nodes, textures, cam, light, min(64, usable_cores))
            frame_ok = bool(np.array_equal(got_img, ref_img)) and (got_st.rays, got_st.sum_complexity) == (ref_rays, ref_sumc)
            out["config"]["frame_equals_oracle"] = frame_ok
            out["config"]["frame_equals_oracle_is"] = (
                f"frame 0 ({spp} samples of all {W * H} pixels) rendered by {chk.lastKernel()} on a renderer configured like the timed ones vs "
                f"oracle/vrc_oracle.c: resolved RGBA8 image bit for bit, rays {got_st.rays} vs {ref_rays}, loop iterations {got_st.sum_complexity} vs "
                f"{ref_sumc}; oracle time {time.perf_counter() - t_or:.1f} s")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, svo.downloadNodes(), textures, cam, light)
    # N > 1, untimed extra (round-5 verdict, Next 2): ONE invocation reports all three exchange forms.  After the timed region
    # (the headline form, --gather) the SAME ranks, renderers, streams and buffers run a short block of frames (3 blocks of
    # `steps`, median) for each of the other forms -- RCCL gather to rank 0, RCCL all-gather, direct peer writes -- each with its
    # own gathered-frame-0-equals-the-oracle bit and every rank's host wait for its exchange to empty.  A form that cannot be
    # set up (hipIpc refused ...) records its error and the run goes on; nothing here can change `value`, and a deadline guards
    # the headline: if the extras have not finished after --forms-deadline-s seconds, rank 0 prints the line it already has
    # (with the reason in extra.exchange_forms) and every rank leaves by itself.  No process is started or replaced.
    if dist_on and many and not args.no_extras:
        forms = {}
        if rank == 0:
            forms[args.gather] = {
                "exchange": FORM_LABEL[args.gather], "headline": True, "ms_per_step": out["ms_per_step"], "value": out["value"],
                "steps": args.steps, "blocks": len(block_ms), "frame_equals_oracle": out["config"].get("frame_equals_oracle"),
                "gathered_frame_equals_unsharded": verified,
                "per_rank_exchange_wait_ms": [p_["drain_wait_ms_per_block"] for p_ in per_rank],
                "per_rank_exchange_ms_per_step": [p_["exchange_ms_per_step"] for p_ in per_rank]}
            out.setdefault("extra", {})["exchange_forms"] = forms
        dist.barrier()                                         # rank 0 has put the headline line together (oracle frame): the clock starts for everyone

        why = f"the other exchange forms did not finish within {args.forms_deadline_s:.0f} s; the headline measurement above is complete"
        fallback_line = None
        if rank == 0:                                              # the line as it stands, serialised NOW: the timer thread touches nothing the main thread writes
            fb = dict(out)
            fb["extra"] = dict(out["extra"], exchange_forms={args.gather: forms[args.gather], "deadline": why})
            fallback_line = json.dumps(fb)
        import threading
        finishing = threading.Lock()                               # whoever holds it prints the line: the timer thread or the main thread, never both

        def give_up():
            if not finishing.acquire(blocking=False):
                return                                             # the forms are done and the main thread is on its way out
            try:
                if rank == 0:
                    sys.stdout.flush()
                    print(fallback_line, flush=True)
                print(f"bench.py: rank {rank}: {why}", file=sys.stderr, flush=True)
                if _PEER_EXCHANGE:
                    _PEER_EXCHANGE[0].wait(_PEER_EXCHANGE[1], timeout_s=5.0)      # no stream-ordered flag wait stays behind on the device
            except BaseException:                                 # noqa: BLE001
                pass
            os._exit(0 if out is None or out["config"].get("frame_equals_oracle") is not False else 1)
        guard = threading.Timer(args.forms_deadline_s, give_up)
        guard.daemon = True
        guard.start()

        def close_peers():
            nonlocal peers
            if peers is not None:
                dist.barrier()
                peers.close()
                peers = None
                del _PEER_EXCHANGE[:]
                dist.barrier()

        n_f = max(3, min(args.steps, 100))
        for form in other_exchange_forms(args.gather, max(world, 2)):
            rec = {"exchange": FORM_LABEL[form], "headline": False, "steps": n_f}
            try:
                close_peers()
                state["gather"] = form
                if form == "peer":
                    state["i"] += (-state["i"]) % F                 # renderer / framebuffer i % F == flag slot (i - base) % F
                    state["base"] = state["i"]
                    peers = vdist.PeerFrames(rcs, local_rank, rank, world, dist.group.WORLD,
                                             "/vrc_bench_peer_%s_x" % os.environ.get("MASTER_PORT", "0"))
                    _PEER_EXCHANGE[:] = [peers, [ts.cuda_stream for ts in tstreams]]
                for _ in range(max(F, 3)):
                    step(False)
                drain()
                runs = []
                for _blk in range(3):
                    dist.barrier()
                    torch.cuda.synchronize()
                    t_f = time.perf_counter()
                    for _ in range(n_f):
                        step(False)
                    t_d = time.perf_counter()
                    drain()
                    wait_ms = (time.perf_counter() - t_d) * 1e3
                    dist.barrier()
                    torch.cuda.synchronize()
                    dt_f = torch.tensor([time.perf_counter() - t_f], dtype=torch.float64, device="cuda")
                    dist.all_reduce(dt_f, op=dist.ReduceOp.MAX)
                    runs.append((float(dt_f.item()) / n_f * 1e3, wait_ms))
                med_f = sorted(runs)[len(runs) // 2]
                waits = [None] * world
                dist.all_gather_object(waits, round(med_f[1], 3))
                rec.update({"ms_per_step": round(med_f[0], 4), "ms_per_step_blocks": [round(r_[0], 4) for r_ in sorted(runs)], "blocks": len(runs),
                            "per_rank_exchange_wait_ms": waits})
                if not args.no_oracle_frame:
                    got_f = frame0_through_exchange()
                    if rank == 0 and oracle0 is not None:
                        rec["frame_equals_oracle"] = bool(np.array_equal(got_f[0], oracle0[0])) and (got_f[1], got_f[2]) == (oracle0[1], oracle0[2])
                if rank == 0:
                    rec["value"] = round(out["config"]["rays_per_frame"] / (rec["ms_per_step"] * 1e-3) / 1e6, 2)
            except Exception as e:                                 # noqa: BLE001  (never lose the main result to an extra)
                rec["error"] = repr(e)[:400]
            if rank == 0:
                forms[form] = rec
        try:
            close_peers()
        except Exception as e:                                     # noqa: BLE001
            if rank == 0:
                forms["close_error"] = repr(e)[:300]
        state["gather"] = args.gather
        guard.cancel()
        if not finishing.acquire(blocking=False):                  # the deadline fired a moment ago: that thread prints and leaves
            time.sleep(60.0)
            os._exit(0)
        if rank == 0:
            forms["note"] = ("the headline form is the timed region (`value`); the others ran right after it on the same ranks, renderers and "
                             f"buffers: median of 3 blocks of {n_f} frames between barrier + synchronize, untimed extras.  per_rank_exchange_wait_ms = "
                             "each rank's host wait, after issuing a block, for its own streams and exchanges to empty; frame_equals_oracle = frame 0 "
                             "rendered once more through THAT exchange, gathered image and summed counters vs the oracle's"
                             + ("" if backend == "nccl" else f".  REHEARSAL on {backend}: 'root' and 'all' are the same host-staged all_gather here; "
                                                             "only the control flow and the parity bits mean anything"))
    if dist_on:
        dist.barrier()
    if peers is not None:
        peers.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to the C stdout buffer, which is flushed at exit, i.e. after anything Python
        # printed: flush it first so that the JSON line is the last line of stdout
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
        if out["config"].get("frame_equals_oracle") is False:
            sys.exit("bench.py: the timed renderer's frame 0 differs from the oracle's (config.frame_equals_oracle)")


_PEER_EXCHANGE = []     # [PeerFrames, stream pointers] of a --gather peer run


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        # direct peer writes: a rank that fails must not leave with stream-ordered flag waits pending on the device -- the
        # watchdog drains its streams with every wait released (and tells the other ranks to give up too)
        if _PEER_EXCHANGE and not isinstance(sys.exc_info()[1], SystemExit):
            try:
                _PEER_EXCHANGE[0].wait(_PEER_EXCHANGE[1], timeout_s=10.0)
            except Exception:                                     # noqa: BLE001
                pass
        raise
