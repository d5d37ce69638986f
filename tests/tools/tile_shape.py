#!/usr/bin/env python3
"""Offline study (CPU, oracle only; round-3 verdict item 2): the lane <-> (pixel, sample) map of the stage-synchronous frame
kernel.  The image does not depend on it (the RNG is keyed by the global pixel, vrc_device.h counter_rand), so it is a free
parameter of the launch; what it changes is how long each stage of a wave lasts (a stage lasts as long as its longest ray)
and how many of the walk loop's blocks a round issues (a block is issued when ANY lane needs it, csrc/vrc_walk.h).

Maps studied: a wave = tw x th pixels x ns samples, tw * th * ns = 64.  ns = 1 is the shipped form (the wave loops over the
samples of its 64 pixels); ns = 4 puts the four samples of 16 pixels side by side in one wave (for a pinhole camera the
primary and the shadow walk of a pixel are the same for every sample, so the four lanes of a pixel stay together there).

From the oracle's per-iteration op traces of the same pixels (orc_render_frame_optrace_geo, samples 0..3) -- every map is
priced on exactly the same rays: 64 x 8-pixel super-tiles of a band of the frame, each ray started where the kernel starts it
(tests/tools/path_schedule.py).  Cost of a wave-stage = sum over rounds of header 13 (15 with the LOD test) + descend 25 if
any lane descends + advance 15 if any advances + pop 22 if any pops (VALU instructions, vrc_walk.h).
ANALYSIS INFRASTRUCTURE: uses oracle/ only.

    python tests/tools/tile_shape.py [--band 16] [--out profiles/r04/tile_shape.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

from path_schedule import H_, D_, A_, P_, camera_path_scale, start_scale_next_to   # noqa: E402

MAPS = [(8, 8, 1), (16, 4, 1), (32, 2, 1), (64, 1, 1), (4, 8, 2), (8, 4, 2), (16, 2, 2), (32, 1, 2),
        (4, 4, 4), (8, 2, 4), (16, 1, 4), (2, 8, 4)]
NAMES = ["primary", "shadow", "GI", "GI shadow"]


def wave_costs(ops, length, skip, lod):
    """ops (n, 64, L) u8, length / skip (n, 64): the shipped policy (every active lane does one iteration per round).
    Returns per wave: cost (VALU instructions issued), rounds, lane-iterations executed."""
    n, _, L = ops.shape
    start = np.minimum(skip, length)
    todo = length - start
    rounds = todo.max(axis=1)
    cost = np.zeros(n)
    h = H_ + (2.0 if lod else 0.0)
    idx = start.copy()
    ar = np.arange(n)[:, None], np.arange(64)[None, :]
    for r in range(int(rounds.max()) if n else 0):
        active = (r < todo)
        if not active.any():
            break
        cur = ops[ar[0], ar[1], np.minimum(idx, L - 1)]
        d = (active & (cur == 0)).any(axis=1)
        a = (active & ((cur == 1) | (cur == 2))).any(axis=1)
        p = (active & (cur == 2)).any(axis=1)
        cost += active.any(axis=1) * h + d * D_ + a * A_ + p * P_
        idx += active
    return cost, rounds, todo.sum(axis=1)


def trace_pose(L, O, vrc, nodes, tex, depth, W, Hh, pitch, aperture, band, threads, spp, LEN=192):
    cam = vrc.reference_camera(depth, pitch=pitch, aperture=aperture, focal_length=(60.0 * (1 << depth) / 512.0 if aperture else 1.0),
                               make_rotation=lambda x, y: O.make_rotation(x, y))
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    f = np.float32
    cam_pos = (np.asarray(cam.position, f) * (f(1.0) / f(1 << depth)) + f(1.0)).astype(f)
    cam_scales = np.array([camera_path_scale(nodes, depth, cam_pos, v, centre_planes=(aperture == 0.0)) for v in range(8)], np.int64)
    blocks = [b for b in range(Hh // 8) if b % band == band // 2]
    rows = np.concatenate([np.arange(b * 8, b * 8 + 8) for b in blocks])
    out = []
    for s in range(spp):
        trace = np.zeros((Hh, W, 8), np.uint16)
        ops = np.zeros((Hh, W, 8, LEN), np.uint8)
        geo = np.zeros((Hh, W, 8, 8), np.float32)
        p = O.make_params(W, Hh, vrc.reference_light(depth), use_gi=1, use_samples=1, shadow_samples=1, row_block=8,
                          shard_index=band // 2, shard_count=band, frame_index=s)
        L.orc_render_frame_optrace_geo(O._ptr(nodes), depth, O._ptr(tex[0]), O._ptr(tex[1]), C.byref(ocam), C.byref(p), threads,
                                       O._ptr(trace), O._ptr(ops), LEN, O._ptr(geo))
        trace, ops, geo = trace[rows, :, :4], ops[rows, :, :4], geo[rows, :, :4]
        org, hpos, hscale = geo[..., 0:3], geo[..., 3:6], geo[..., 6].astype(np.int64)
        hkind, signs = geo[..., 7].astype(np.int64) % 16, geo[..., 7].astype(np.int64) // 16
        skip = np.zeros(trace.shape, np.int64)
        if aperture == 0.0:     # every camera ray starts at the camera, on the path of its direction signs
            skip[:, :, 0] = np.where(trace[:, :, 0] > 0, 22 - cam_scales[signs[:, :, 0]], 0)
        else:                   # the lens point is next to the camera position: it starts where their paths from the root part
            cs = int(cam_scales.min())   # (a camera on a centre plane of the cube: 22, its lens rays start at the root)
            s0 = start_scale_next_to(org[:, :, 0], np.broadcast_to(cam_pos, org[:, :, 0].shape), np.full(trace.shape[:2], cs))
            skip[:, :, 0] = np.where((s0 < 22) & (trace[:, :, 0] > 0), 22 - s0, 0)
        for ray, prev in ((1, 0), (2, 0), (3, 2)):
            s0 = start_scale_next_to(org[:, :, ray], hpos[:, :, prev], hscale[:, :, prev])
            skip[:, :, ray] = np.where((s0 < 22) & (hkind[:, :, prev] != 0) & (trace[:, :, ray] > 0), 22 - s0, 0)
        out.append((trace.astype(np.int64), ops, skip))
    # (sample, block, row in block, x, stage ...)
    nb = len(blocks)
    tr = np.stack([o[0] for o in out]).reshape(spp, nb, 8, W, 4)
    op = np.stack([o[1] for o in out]).reshape(spp, nb, 8, W, 4, LEN)
    sk = np.stack([o[2] for o in out]).reshape(spp, nb, 8, W, 4)
    return tr, op, sk


def price(tr, op, sk, spp):
    """every map on the same pixels x samples; returns {map: per-stage [cost, rounds, lane iterations]} + the wave count"""
    S, nb, _, W, _ = tr.shape
    LEN = op.shape[-1]
    res = {}
    for (tw, th, ns) in MAPS:
        if ns > spp or W % tw:
            continue
        # lanes: (sample-in-group, y-in-tile, x-in-tile); waves: (sample group, block, tile row, tile column)
        def lanes(a, extra):
            a = a.reshape((spp // ns, ns, nb, 8 // th, th, W // tw, tw) + extra)
            a = np.moveaxis(a, (1, 4, 6), (4, 5, 6))           # -> (groups, nb, 8/th, W/tw, ns, th, tw, ...)
            return a.reshape((-1, 64) + extra)
        stage = []
        for st in range(4):
            c, r, it = wave_costs(lanes(op[..., st, :], (LEN,)), lanes(tr[..., st], ()), lanes(sk[..., st], ()), lod=st >= 2)
            stage.append([float(c.sum()), float(r.sum()), float(it.sum())])
        res[f"{tw}x{th}x{ns}spp"] = stage
    n_waves = spp * nb * 8 * W // 64
    return res, n_waves


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--band", type=int, default=16, help="trace 1 / band of the 8-row blocks")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import cpuvoxelraycaster_amd as vrc
    import oracle_lib as O
    depth, W, Hh, spp = 9, 1920, 1080, 4
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    tex = O.load_textures()
    L = O.lib()
    L.orc_render_frame_optrace_geo.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(O.Camera),
                                               C.POINTER(O.FrameParams), C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    poses = [("c3 pitch -0.5 (the metric's pose)", -0.5, 0.0), ("c3 pitch 0", 0.0, 0.0), ("c3 pitch -1.2", -1.2, 0.0),
             ("c4 pitch -0.5, aperture 0.5", -0.5, 0.5)]
    doc = {"what": "VALU wave-instructions of the walk loop per wave-stage (64 lanes x one stage of one sample) under each lane <-> "
                   "(pixel, sample) map tw x th x ns; same rays for every map; utilisation = lane-iterations / (64 x rounds)",
           "band": a.band, "costs": {"header": H_, "header_lod": H_ + 2, "descend": D_, "advance": A_, "pop": P_}, "poses": {}}
    for name, pitch, aperture in poses:
        tr, op, sk = trace_pose(L, O, vrc, nodes, tex, depth, W, Hh, pitch, aperture, a.band, a.threads, spp)
        res, n_waves = price(tr, op, sk, spp)
        base = sum(s[0] for s in res["8x8x1spp"])
        print(f"\n{name}: {n_waves} wave-samples; walk VALU per wave-sample by stage [rounds] (lane utilisation)")
        rows = {}
        for k, stage in res.items():
            tot = sum(s[0] for s in stage)
            cells = "  ".join(f"{NAMES[i]} {stage[i][0] / n_waves:6.0f} [{stage[i][1] / n_waves:5.1f}] ({stage[i][2] / max(1.0, 64.0 * stage[i][1]):.2f})" for i in range(4))
            print(f"  {k:12s} {cells}   total {tot / n_waves:7.0f} ({(tot / base - 1) * 100:+5.1f} %)")
            rows[k] = {"total_per_wave_sample": round(tot / n_waves, 1), "vs_8x8": round(tot / base - 1, 4),
                       "stages": {NAMES[i]: {"valu": round(stage[i][0] / n_waves, 1), "rounds": round(stage[i][1] / n_waves, 2),
                                             "lane_utilisation": round(stage[i][2] / max(1.0, 64.0 * stage[i][1]), 3)} for i in range(4)}}
        doc["poses"][name] = {"wave_samples": n_waves, "maps": rows}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
