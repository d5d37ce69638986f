#!/usr/bin/env python3
"""Offline study (CPU, oracle only): instruction issue of the walk loop under different ways of scheduling its three
divergent paths inside a wave, from the oracle's per-iteration op traces (orc_render_frame_optrace_geo) of a strided band of
the C3 frame.  A wave issues a block at full cost whenever ANY of its lanes needs it; per round (vrc_walk.h):
header 13 VALU (15 with the LOD test), descend 25, advance 15, pop 22.  Every ray starts where the kernel starts it: below
the root, at the scale vrc_device.h's start_scale_next_to / camera_path give for it (the iterations before that are not
executed), computed here from the trace's ray origins and hit positions.  ANALYSIS INFRASTRUCTURE: uses oracle/ only.

    python tests/tools/path_schedule.py [--band 16]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

H_, D_, A_, P_ = 13.0, 25.0, 15.0, 22.0
MASKS = None     # --dump-masks: [(block, lanes in EXEC)] in issue order, of the shipped policy
MASK_MULT = 1


def highbit(x):
    """index of the highest set bit of uint32 values, -1 for 0"""
    x = np.asarray(x, np.uint64)
    out = np.full(x.shape, -1, np.int64)
    for b in range(32):
        out[(x >> np.uint64(b)) & np.uint64(1) == 1] = b
    return out


def start_scale_next_to(o, h, hit_scale):
    """numpy restatement of vrc_device.h's start_scale_next_to; o, h: (..., 3) float32"""
    a = np.ascontiguousarray(o, np.float32).view(np.uint32).astype(np.uint64)
    b = np.ascontiguousarray(h, np.float32).view(np.uint32).astype(np.uint64)
    m32 = np.uint64(0xffffffff)
    db = np.bitwise_or.reduce(a ^ b, axis=-1)
    mb = np.bitwise_or.reduce(((a - np.uint64(8)) & m32) ^ ((a + np.uint64(8)) & m32), axis=-1)
    return np.maximum(highbit((db | (mb + mb)) & m32), np.asarray(hit_scale, np.int64))


def camera_path_scale(nodes, depth, cam, variant=0, centre_planes=True):
    """numpy restatement of camera_path: the scale at which camera rays start (22 = at the root); variant bit i = the ray goes
    towards +axis i, which matters for a coordinate exactly on the cube's centre plane 1.5 (vrc_device.h: camera_path)"""
    cam = np.asarray(cam, np.float32)
    a = cam.view(np.uint32).astype(np.uint64)
    on = [bool(centre_planes and float(c) == 1.5) for c in cam]
    m32 = np.uint64(0xffffffff)
    mb = 0
    for i in range(3):
        if not on[i]:
            mb |= int(((a[i] - np.uint64(8)) & m32) ^ ((a[i] + np.uint64(8)) & m32))
    if not all((int(v) >> 23) == 127 for v in a):
        return 22
    a = np.array([int(a[i]) - (1 if on[i] and not (variant >> i) & 1 else 0) for i in range(3)], np.uint64)
    lowest = max(int(highbit(np.array([(mb + mb) & 0xffffffff]))[0]), 23 - depth, 23 - 3 * (depth // 3))
    raw = nodes.view(np.uint32).reshape(-1, 2)
    s, parent = 22, 0
    while s > lowest:
        nd = raw[parent]
        cs = 7 ^ (((int(a[0]) >> s) & 1) | (((int(a[1]) >> s) & 1) << 1) | (((int(a[2]) >> s) & 1) << 2))
        masks = int(nd[0]) >> cs
        if (masks & 0x10100) != 0x100:
            break
        parent += int(nd[1]) + cs
        s -= 1
    return s


def simulate(ops, length, policy, lod, start=None):
    """ops: (64, L) uint8 op codes, length: (64,) iterations per lane (0 = no ray), start: (64,) iterations a lane does not
    execute (it starts below the root).  Returns issued VALU instructions."""
    n = len(length)
    pos = np.zeros(n, np.int64) if start is None else np.minimum(start.astype(np.int64), length)   # next iteration of each lane
    waiting_pop = np.zeros(n, bool)       # header + advance done, pop outstanding
    cost = 0.0
    blocks = np.zeros(4, np.int64)        # times the header / descend / advance / pop block was issued
    h_cost = H_ + (2.0 if lod else 0.0)
    k = policy.get("pop_every", 1)
    thr = policy.get("pop_threshold", 1)
    rnd = 0
    while True:
        active = (pos < length) & ~waiting_pop
        if not active.any() and not waiting_pop.any():
            break
        rnd += 1
        if active.any():
            cur = ops[np.arange(n), np.minimum(pos, ops.shape[1] - 1)]
            cost += h_cost
            blocks[0] += 1
            if MASKS is not None and k == 1 and thr == 1:
                MASKS.append((0, int(active.sum()) * MASK_MULT))
            d = active & (cur == 0)
            a = active & ((cur == 1) | (cur == 2))
            if d.any():
                cost += D_
                blocks[1] += 1
                if MASKS is not None and k == 1 and thr == 1:
                    MASKS.append((1, int(d.sum()) * MASK_MULT))
            if a.any():
                cost += A_
                blocks[2] += 1
                if MASKS is not None and k == 1 and thr == 1:
                    MASKS.append((2, int(a.sum()) * MASK_MULT))
            need_pop = active & (cur == 2)
            waiting_pop |= need_pop
            pos[active & ~need_pop] += 1        # descend / plain advance / exit complete this round
        # pop block: by policy
        nwait = int(waiting_pop.sum())
        run_pop = False
        if nwait:
            others = ((pos < length) & ~waiting_pop).any()
            if k > 1:
                run_pop = (rnd % k == 0) or not others
            else:
                run_pop = nwait >= thr or not others
        if run_pop:
            cost += P_
            blocks[3] += 1
            if MASKS is not None and k == 1 and thr == 1:
                MASKS.append((3, int(waiting_pop.sum()) * MASK_MULT))
            pos[waiting_pop] += 1
            waiting_pop[:] = False
    policy["_blocks"] = policy.get("_blocks", np.zeros(4, np.int64)) + blocks
    return cost, rnd


def simulate_work_list(ops, length, T, lod, start=None):
    """A work list inside the wave (round 5): lanes wait in the set of the block they need next (descend / advance / pop) and a
    round issues the header for the lanes that need none, then every pending set with >= T lanes and always the fullest one.
    T = 1 is the shipped loop.  Returns issued VALU instructions, rounds."""
    n = len(length)
    pos = np.zeros(n, np.int64) if start is None else np.minimum(start.astype(np.int64), length)
    pend = np.zeros(n, np.int8)                 # 0 none, 1 descend, 2 advance, 3 advance then pop, 4 pop
    cost, rounds = 0.0, 0
    h_cost = H_ + (2.0 if lod else 0.0)
    lane = np.arange(n)
    while True:
        alive = pos < length
        if not alive.any():
            return cost, rounds
        rounds += 1
        fresh = alive & (pend == 0)
        if fresh.any():
            cur = ops[lane, np.minimum(pos, ops.shape[1] - 1)]
            cost += h_cost
            pend[fresh & (cur == 0)] = 1
            pend[fresh & (cur == 1)] = 2
            pend[fresh & (cur == 2)] = 3
            pos[fresh & (cur > 2)] += 1         # an exit the header decides
        cnt = {1: int((alive & (pend == 1)).sum()), 2: int((alive & ((pend == 2) | (pend == 3))).sum()), 4: int((alive & (pend == 4)).sum())}
        big = max(cnt, key=lambda k: cnt[k])
        for kind in (1, 2, 4):
            if cnt[kind] and (cnt[kind] >= T or kind == big):
                if kind == 1:
                    m = alive & (pend == 1); cost += D_; pos[m] += 1; pend[m] = 0
                elif kind == 2:
                    m2, m3 = alive & (pend == 2), alive & (pend == 3)
                    cost += A_; pos[m2] += 1; pend[m2] = 0; pend[m3] = 4
                else:
                    m = alive & (pend == 4); cost += P_; pos[m] += 1; pend[m] = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--band", type=int, default=32, help="render 1 / band of the 8-row blocks")
    ap.add_argument("--tiles", type=int, default=600)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--from-root", action="store_true", help="every ray from the root (the reference's walk, the kernels before round 3)")
    ap.add_argument("--dump", default="", help="write the block issue frequencies of the shipped policy as JSON (tools/issue_model.py reads it)")
    ap.add_argument("--quad", action="store_true",
                    help="the pinhole kernel's quadrant walks (round 5): the primary and the shadow stage are walked one 4 x 4 quadrant of "
                         "the tile at a time, each pixel's ray in four lanes (per tile-sample: four walks of 16 distinct rays instead of one of 64)")
    ap.add_argument("--work-list", action="store_true",
                    help="also price a work list inside the wave: lanes wait in the set of the block they need next, a round issues the "
                         "sets with >= T lanes and always the fullest (simulate_work_list)")
    ap.add_argument("--dump-masks", default="",
                    help="write the sequence of (block, lanes in EXEC) the shipped policy issues, uint32 = block << 8 | lanes, in issue order "
                         "(a unit's four GI / GI-shadow walks per four quadrant walks); tools/micro/mask_replay.hip replays it")
    a = ap.parse_args()
    global MASKS, MASK_MULT
    if a.dump_masks:
        MASKS = []
    import cpuvoxelraycaster_amd as vrc
    import oracle_lib as O
    depth, W, Hh = 9, 1920, 1080
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    tex = O.load_textures()
    cam = vrc.reference_camera(depth, pitch=-0.5, make_rotation=lambda x, y: O.make_rotation(x, y))
    ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
    L = O.lib()
    LEN = 192
    trace = np.zeros((Hh, W, 8), np.uint16)
    ops = np.zeros((Hh, W, 8, LEN), np.uint8)
    geo = np.zeros((Hh, W, 8, 8), np.float32)
    p = O.make_params(W, Hh, vrc.reference_light(depth), use_gi=1, use_samples=1, shadow_samples=1, row_block=8,
                      shard_index=a.band // 2, shard_count=a.band)
    L.orc_render_frame_optrace_geo.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(O.Camera),
                                               C.POINTER(O.FrameParams), C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.orc_render_frame_optrace_geo(O._ptr(nodes), depth, O._ptr(tex[0]), O._ptr(tex[1]), C.byref(ocam), C.byref(p), a.threads,
                                   O._ptr(trace), O._ptr(ops), LEN, O._ptr(geo))
    # where the kernel starts each ray (iterations not executed = 22 - start scale): camera rays next to the camera position
    # on the camera cell's path, shadow and GI rays next to the primary hit, GI shadow rays next to the GI hit
    f = np.float32
    cam_pos = (np.asarray(cam.position, f) * (f(1.0) / f(1 << depth)) + f(1.0)).astype(f)
    cam_scales = np.array([camera_path_scale(nodes, depth, cam_pos, v) for v in range(8)], np.int64)
    cam_scale = int(cam_scales.min())
    org, hpos, hscale = geo[..., 0:3], geo[..., 3:6], geo[..., 6].astype(np.int64)
    hkind, signs = geo[..., 7].astype(np.int64) % 16, geo[..., 7].astype(np.int64) // 16
    skip = np.zeros((Hh, W, 8), np.int64)
    # pinhole camera: every camera ray starts at the camera, on the path of its direction signs
    skip[:, :, 0] = np.where(trace[:, :, 0] > 0, 22 - cam_scales[signs[:, :, 0]], 0)
    for ray, prev in ((1, 0), (2, 0), (3, 2)):
        s0 = start_scale_next_to(org[:, :, ray], hpos[:, :, prev], hscale[:, :, prev])
        skip[:, :, ray] = np.where((s0 < 22) & (hkind[:, :, prev] != 0) & (trace[:, :, ray] > 0), 22 - s0, 0)
    if a.from_root:
        skip[:] = 0
    rows = [y for y in range(Hh) if (y // 8) % a.band == a.band // 2 and y // 8 * 8 + 8 <= Hh]
    blocks = sorted({y // 8 for y in rows})
    tiles = [(b, tx) for b in blocks for tx in range(W // 8)]
    rng = np.random.default_rng(0)
    rng.shuffle(tiles)
    tiles = tiles[:a.tiles]
    policies = {"now (pop whenever a lane needs it)": {},
                "pop every 2nd round": {"pop_every": 2}, "pop every 3rd round": {"pop_every": 3},
                "pop when >= 8 lanes wait": {"pop_threshold": 8}, "pop when >= 16 lanes wait": {"pop_threshold": 16},
                "pop when >= 24 lanes wait": {"pop_threshold": 24}, "pop when >= 32 lanes wait": {"pop_threshold": 32}}
    names = ["primary", "shadow", "GI", "GI shadow"]
    WL = {T: (np.zeros(4), np.zeros(4)) for T in (1, 4, 8, 12, 16, 24, 32)}
    tot = {k: np.zeros(4) for k in policies}
    rounds = {k: np.zeros(4) for k in policies}
    shipped = "now (pop whenever a lane needs it)"
    stage_blocks = np.zeros((4, 4), np.float64)     # [stage][header, descend, advance, pop] of the shipped policy
    lane_iters = np.zeros(4, np.int64)
    lane_skipped = np.zeros(4, np.int64)
    for (b, tx) in tiles:
        for st in range(4):
            o = ops[b * 8:b * 8 + 8, tx * 8:tx * 8 + 8, st].reshape(64, LEN)
            ln = trace[b * 8:b * 8 + 8, tx * 8:tx * 8 + 8, st].reshape(64).astype(np.int64)
            sk = skip[b * 8:b * 8 + 8, tx * 8:tx * 8 + 8, st].reshape(64)
            lane_iters[st] += (ln - np.minimum(sk, ln)).sum()
            lane_skipped[st] += np.minimum(sk, ln).sum()
            # the lanes of one walk: the whole tile, or (quadrant walks, sample-invariant stages) one quadrant -- its 16 rays stand
            # for 64 lanes, four identical copies each, which changes neither the rounds nor which blocks a round issues
            if a.quad and st < 2:
                lanes = np.arange(64).reshape(8, 8)
                groups = [lanes[qy * 4:qy * 4 + 4, qx * 4:qx * 4 + 4].reshape(-1) for qy in range(2) for qx in range(2)]
            else:
                groups = [np.arange(64)]
            # (a quadrant walk covers 16 pixels x 4 samples = one tile-sample's worth of rays: per tile-sample the stage costs the
            # MEAN of the four quadrants' walks)
            if a.work_list:
                for T in WL:
                    for g in groups:
                        c, r = simulate_work_list(o[g], ln[g], T, st >= 2, sk[g])
                        WL[T][0][st] += c / len(groups)
                        WL[T][1][st] += r / len(groups)
            for k, pol in policies.items():
                pol.pop("_blocks", None)
                for g in groups:
                    MASK_MULT = 64 // len(g)
                    n0 = len(MASKS) if (MASKS is not None and k == shipped) else 0
                    c, r = simulate(o[g], ln[g], pol, lod=st >= 2, start=sk[g])
                    if MASKS is not None and k == shipped and a.quad and len(g) == 64:
                        MASKS.extend(MASKS[n0:] * 3)
                    tot[k][st] += c / len(groups)
                    rounds[k][st] += r / len(groups)
                if k == shipped:
                    stage_blocks[st] += pol["_blocks"] / float(len(groups))
    if a.dump:
        import json
        doc = {"workload": "c3: 512^3 terrain, 1920x1080, GI, pitch -0.5, one sample of every pixel of the sampled tiles"
                           + ("; primary and shadow stage as quadrant walks (4 walks of a 4 x 4 quadrant per tile-sample)" if a.quad else ""),
               "tiles": len(tiles), "band": a.band,
               "stages": {names[i]: {"header": int(round(stage_blocks[i][0])), "descend": int(round(stage_blocks[i][1])),
                                     "advance": int(round(stage_blocks[i][2])), "pop": int(round(stage_blocks[i][3])),
                                     "starts": len(tiles), "lane_iterations": int(lane_iters[i]),
                                     "lane_iterations_not_executed": int(lane_skipped[i])} for i in range(4)},
               "camera_start_scale": int(cam_scale),
               "note": "times each block of the walk loop (csrc/vrc_walk.h) is issued by a wave, summed over the sampled tiles: a "
                       "block is issued when any of the wave's 64 lanes needs it; from the oracle's per-iteration op traces "
                       "(tests/tools/path_schedule.py --dump)"}
        json.dump(doc, open(a.dump, "w"), indent=1)
    if a.dump_masks:
        seq = np.array([(b << 8) | min(n, 64) for b, n in MASKS], np.uint32)
        seq.tofile(a.dump_masks)
        cls = np.array([14, 24, 16, 21])[seq >> 8]
        print(f"{len(seq)} block issues written to {a.dump_masks}; with <= 8 lanes in EXEC: {np.mean((seq & 255) <= 8) * 100:.1f} % of the issues, "
              f"{(cls * ((seq & 255) <= 8)).sum() / cls.sum() * 100:.1f} % of the loop's VALU instructions")
        for i, nm in enumerate(("header", "descend", "advance", "pop")):
            m = (seq >> 8) == i
            print(f"  {nm:8s} {m.mean() * 100:5.1f} % of the issues, mean {(seq[m] & 255).mean():5.1f} lanes, <= 8 lanes in {np.mean((seq[m] & 255) <= 8) * 100:5.1f} %")
    base = tot["now (pop whenever a lane needs it)"]
    print(f"{len(tiles)} tiles; issued VALU instructions per tile-sample and stage (rounds in brackets)")
    for k in policies:
        line = "  ".join(f"{names[i]} {tot[k][i] / len(tiles):7.0f} [{rounds[k][i] / len(tiles):5.1f}]" for i in range(4))
        print(f"{k:36s} {line}   total {tot[k].sum() / len(tiles):7.0f} ({(tot[k].sum() / base.sum() - 1) * 100:+5.1f} %)")
    if a.work_list:
        for T, (c, r) in WL.items():
            line = "  ".join(f"{names[i]} {c[i] / len(tiles):7.0f} [{r[i] / len(tiles):5.1f}]" for i in range(4))
            print(f"{'work list, sets with >= %d lanes' % T:36s} {line}   total {c.sum() / len(tiles):7.0f} ({(c.sum() / WL[1][0].sum() - 1) * 100:+5.1f} %)")


if __name__ == "__main__":
    main()
