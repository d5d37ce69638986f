#!/usr/bin/env python3
"""Camera-path replay driver: renders the ticks of a replay file in the
reference's format (include/replay.hpp: `t x y z view_x view_y` per tick, whitespace
separated) and reports per-frame time.  Without --replay it writes and uses a synthetic
orbit around the 512^3 terrain."""
import argparse
import json
import math
import os
import sys

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # more than three frame streams would otherwise share hardware queues

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _extract_float(text, pos):
    """One `file >> float` of the reference's reader (replay.hpp:24, libstdc++ num_get): skip whitespace, collect
    [sign] digits [. digits] [e [sign] digits] as far as the characters fit, convert; returns (value, new_pos) or
    (None, pos) when the extraction fails (nothing collected, incomplete exponent, float32 overflow)."""
    n = len(text)
    while pos < n and text[pos].isspace():
        pos += 1
    i = pos
    if i < n and text[i] in "+-":
        i += 1
    digits = 0
    while i < n and text[i].isdigit():
        i += 1; digits += 1
    if i < n and text[i] == ".":
        i += 1
        while i < n and text[i].isdigit():
            i += 1; digits += 1
    if digits and i < n and text[i] in "eE":
        i += 1
        if i < n and text[i] in "+-":
            i += 1
        while i < n and text[i].isdigit():
            i += 1
    try:
        v = float(text[pos:i])
    except ValueError:
        return None, pos
    v32 = float(np.float32(v)) if abs(v) < 3.5e38 else math.inf
    if math.isinf(v32):
        return None, pos
    return v32, i


def load_replay(path):
    """Same ticks as ReplayElements::loadFromFile (include/replay.hpp:18-33): values are read one `>>` at a time, so a
    tick may span lines and a number may be glued to following text; reading stops at the first failed extraction;
    a missing file gives no ticks.  Pinned against the reference's reader by tests/golden/replay_cases.json."""
    ticks = []
    try:
        text = open(path).read()
    except OSError:
        return ticks
    pos = 0
    while True:
        vals = []
        for _ in range(6):
            v, pos = _extract_float(text, pos)
            if v is None:
                return ticks
            vals.append(v)
        ticks.append(tuple(vals))


def synthetic_orbit(path, n=60, size=512.0):
    with open(path, "w") as f:
        for i in range(n):
            a = 2 * math.pi * i / n
            x, z = size / 2 + 100 * math.cos(a), size / 2 + 100 * math.sin(a)
            yaw = math.atan2(size / 2 - x, size / 2 - z)   # look at the centre
            f.write(f"{i / 30:.4f} {x:.3f} {size / 2 - 70:.3f} {z:.3f} {-yaw:.5f} -0.45\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replay", default="")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--inflight", type=int, default=0, help="F > 0: F renderers / streams take turns (whole-spp work units, fused "
                    "resolve), the orbit is rendered 5 times over and the figure is wall time per frame")
    ap.add_argument("--reuse", action="store_true", help="with --inflight: vrc_renderer_set_invariant_ray_reuse")
    args = ap.parse_args()
    import torch
    import cpuvoxelraycaster_amd as vrc
    path = args.replay
    if not path:
        path = "/tmp/vrc_orbit.replay"
        synthetic_orbit(path)
    ticks = load_replay(path)
    depth = 9
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
    if args.inflight:
        import time
        F = args.inflight
        rcs, streams = [], [torch.cuda.Stream() for _ in range(F)]
        for _ in range(F):
            r = vrc.RayCaster(svo, (args.width, args.height))
            r.setLightPosition(vrc.reference_light(depth))
            r.use_gi, r.use_samples, r.shadow_samples = True, True, 1
            r.setSampleChunk(args.spp)
            r.setInvariantRayReuse(args.reuse)
            rcs.append(r)
        cams = [vrc.make_camera((x, y, z), vrc.make_rotation(vx, vy)) for (t, x, y, z, vx, vy) in ticks]

        def orbit(i0):
            for k, cam in enumerate(cams):
                i = i0 + k
                rcs[i % F].frame_index = i * args.spp
                rcs[i % F].renderFrameResolved(cam, spp=args.spp, stream=streams[i % F].cuda_stream)
        orbit(0)
        torch.cuda.synchronize()
        for r in rcs:
            r.stats(reset=True)
        t0 = time.perf_counter()
        for rep in range(5):
            orbit((rep + 1) * len(cams))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays = sum(r.stats().rays for r in rcs)
        print(json.dumps({"replay": path, "ticks": len(ticks), "frames_in_flight": F, "invariant_ray_reuse": args.reuse,
                          "ms_per_frame": round(dt / (5 * len(cams)) * 1e3, 4), "Mrays_per_s": round(rays / dt / 1e6, 1),
                          "rays_per_frame": rays // (5 * len(cams))}))
        return
    rc = vrc.RayCaster(svo, (args.width, args.height))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
    stream = torch.cuda.current_stream().cuda_stream
    times = []
    for (t, x, y, z, vx, vy) in ticks:
        cam = vrc.make_camera((x, y, z), vrc.make_rotation(vx, vy))
        rc.resetSamples(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc.renderFrame(cam, spp=args.spp, stream=stream)
        rc.samples_to_image(stream)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    st = rc.stats()
    times.sort()
    print(json.dumps({"replay": path, "ticks": len(ticks), "ms_per_frame_median": round(times[len(times) // 2], 3),
                      "ms_per_frame_max": round(times[-1], 3), "Mrays_per_s": round(st.rays / sum(times) / 1e3, 1),
                      "rays_per_frame": st.rays // max(1, len(ticks))}))


if __name__ == "__main__":
    main()
