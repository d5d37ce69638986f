#!/usr/bin/env python3
"""Offline study (CPU, oracle only): how many wave-iterations the stage-synchronous frame kernel needs per 8x8 tile
under different ways of grouping a work unit's rays, from the oracle's per-ray loop-iteration counts
(orc_render_frame_trace).  TEST/ANALYSIS INFRASTRUCTURE: uses oracle/, never the product path.

    python tests/tools/stage_balance.py [--width 1920 --height 1080 --depth 9 --spp 4]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def trace_frame(nodes, depth, tex, cam, light, W, H, sample, threads):
    import oracle_lib as O
    L = O.lib()
    p = O.make_params(W, H, light, use_gi=1, use_samples=1, shadow_samples=1, frame_index=sample)
    trace = np.zeros((H, W, 8), np.uint16)
    img = np.zeros((H, W, 4), np.uint8)
    acc = np.zeros((H, W, 4), np.uint32)
    st = O.FrameStats()
    L.orc_render_frame_trace.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(O.Camera),
                                         C.POINTER(O.FrameParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(O.FrameStats), C.c_int, C.c_void_p]
    L.orc_render_frame_trace(O._ptr(nodes), depth, O._ptr(tex[0]), O._ptr(tex[1]), C.byref(cam), C.byref(p),
                             O._ptr(img), O._ptr(acc), None, C.byref(st), threads, O._ptr(trace))
    return trace[..., :4].astype(np.int32)   # primary, shadow, GI, GI shadow


def tiles(a, H, W):
    """(H, W, ...) -> (tiles, 64, ...) for 8x8 tiles (frame cropped to multiples of 8)"""
    h, w = H // 8 * 8, W // 8 * 8
    a = a[:h, :w]
    rest = a.shape[2:]
    a = a.reshape(h // 8, 8, w // 8, 8, *rest)
    a = np.moveaxis(a, 2, 1)
    return a.reshape((h // 8) * (w // 8), 64, *rest)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=9)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--pitch", type=float, default=-0.5)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--save", default="")
    a = ap.parse_args()
    import cpuvoxelraycaster_amd as vrc
    import oracle_lib as O
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), a.depth)
    tex = O.load_textures()
    cam = vrc.reference_camera(a.depth, pitch=a.pitch, make_rotation=lambda x, y: O.make_rotation(x, y))
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    light = vrc.reference_light(a.depth)
    tr = np.stack([tiles(trace_frame(nodes, a.depth, tex, ocam, light, a.width, a.height, s, a.threads), a.height, a.width)
                   for s in range(a.spp)], axis=2)           # (tiles, 64 lanes, spp, 4 stages)
    if a.save:
        np.save(a.save, tr.astype(np.uint16))
    T = tr.shape[0]
    useful = tr.sum() / (T * 64.0 * a.spp)
    names = ["primary", "shadow", "GI", "GI shadow"]
    print(f"{T} tiles, {a.spp} samples; useful iterations per pixel-sample {useful:.2f}")
    cur = tr.max(axis=1)                                     # (tiles, spp, stage): a stage lasts as long as its longest ray
    for k, n in enumerate(names):
        print(f"  {n:10s} mean/lane {tr[..., k].mean():6.2f}   per-tile max {cur[..., k].mean():6.2f}   ratio {tr[..., k].mean() / cur[..., k].mean():.3f}")
    base = cur.sum(axis=2).mean()
    print(f"stage-synchronous now: {base:.2f} wave-iterations per tile-sample (lane utilisation in the walks {useful / base:.3f})")

    def report(label, per_sample):
        print(f"  {label:58s} {per_sample:7.2f}  ({(per_sample / base - 1) * 100:+5.1f} %)")

    for K in (2, 4):
        if a.spp % K:
            continue
        g = tr.reshape(T, 64, a.spp // K, K, 4)
        for stages, lab in (((2, 3), "GI + GI shadow"), ((0, 1, 2, 3), "all four stages")):
            tot = 0.0
            for k in range(4):
                if k in stages:
                    tot += g[..., k].sum(axis=3).max(axis=1).sum(axis=1).mean() / a.spp * 1.0   # max over lanes of the lane's K rays
                else:
                    tot += cur[..., k].mean()
            report(f"K={K} rays of a lane back to back in one walk, {lab}", tot)
        # work stealing inside the wave: 64*K rays shared by 64 lanes (longest-processing-time bound: sum/64 + last ray)
        tot = 0.0
        for k in range(4):
            if k in (2, 3):
                s = g[..., k].reshape(T, 64 * K, a.spp // K)   # wrong axis order guard below
            tot += 0
        # greedy list scheduling in pixel order
        tot = 0.0
        for k in range(4):
            if k not in (2, 3):
                tot += cur[..., k].mean()
                continue
            acc = 0.0
            gk = g[..., k]                                    # (T, 64, groups, K)
            for grp in range(a.spp // K):
                rays = np.concatenate([gk[:, :, grp, j] for j in range(K)], axis=1)   # (T, 64K): sample-major
                lanes = rays[:, :64].astype(np.int64).copy()
                for j in range(64, 64 * K):
                    i = lanes.argmin(axis=1)
                    lanes[np.arange(T), i] += rays[:, j]
                acc += lanes.max(axis=1).mean()
            tot += acc / a.spp
        report(f"K={K} with work stealing inside the wave (greedy), GI + GI shadow", tot)
    report("perfect balance inside a tile (lower bound) = the useful iterations", useful)

    # K = 2 with what the swap costs: a lane whose first ray has ended gets its second one at a swap event (save the ended
    # ray's result, load the parked ray: ~25-35 instructions for the whole wave = 0.3-0.5 of a 75-instruction iteration);
    # events at fixed iteration counts plus a last one when every first ray has ended
    def k2_events(first, second, times, cswap):
        tot = 0.0
        for t in range(0, first.shape[0], 4):                  # every 4th tile is plenty
            a1, b1 = first[t].astype(float), second[t].astype(float)
            cur = np.where(a1 > 0, a1, b1)
            sec = np.where((a1 > 0) & (b1 > 0), b1, 0.0)
            end, started = cur.copy(), sec <= 0
            for tt in list(times) + [None]:
                if tt is None:
                    tt = end[~started].max() if (~started).any() else 0.0
                sw = (~started) & (end <= tt)
                if not sw.any():
                    continue
                end[end > tt] += cswap
                end[sw] = tt + cswap + sec[sw]
                started |= sw
            tot += end.max()
        return tot / len(range(0, first.shape[0], 4))

    if a.spp >= 2:
        print("K=2 on the GI / GI-shadow stages with the cost of the swap events (wave-iterations for two samples' stage):")
        for k, n in ((2, "GI"), (3, "GI shadow")):
            first, second = tr[:, :, 0, k], tr[:, :, 1, k]
            now = (first.max(axis=1) + second.max(axis=1))[::4].mean()
            free = (first + second).max(axis=1)[::4].mean()
            line = f"  {n:10s} now {now:6.2f}  free swaps {free:6.2f} ({(free / now - 1) * 100:+.1f} %)"
            for times in ([], [16], [20], [16, 22], [12, 16, 20, 24]):
                for cs in (0.3, 0.5):
                    r = k2_events(first, second, times, cs)
                    line += f" | events {times or 'end only'} c={cs}: {(r / now - 1) * 100:+.1f} %"
            print(line)


if __name__ == "__main__":
    main()
