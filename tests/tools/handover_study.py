#!/usr/bin/env python3
"""Offline study (CPU, oracle only; round-4 verdict, Next 4): the one family of schedules DESIGN.md section 9 had not priced --
handing the STRAGGLERS of a stage to somebody else instead of letting 64 lanes wait for them.

The frame kernel is stage-synchronous: a wave-stage lasts as long as its longest ray, and the GI / GI-shadow stages run at
0.44 / 0.51 of their lanes.  Schemes priced here on the executed per-ray iteration counts of the C3 frame (every ray started
where the kernel starts it, tests/tools/tile_shape.py: trace_pose), all bit-exact by construction (a ray's walk does not depend
on which lane runs it):

  wg(T)      workgroup-level hand-over (the verdict's sketch): the four waves of a workgroup run the same stage on four tiles;
             a wave whose live lanes drop to <= T writes their Ray state (~60 B each) into an LDS pool and stops; ONE collector
             wave walks the pooled rays compacted (64 at a time), writes the hits back, and the four waves go on to the next
             stage together.  Rounds = sum of the four cut stages + the collector's walks.
  unit(T)    the same inside ONE wave over the four samples of its whole-spp unit (no barrier): the GI stages of samples 0..3
             are each cut at <= T live lanes, the survivors parked, then walked together; likewise the GI-shadow stages.
  ideal(wg)  continuous perfect compaction over the 256 rays of a workgroup-stage at zero cost: sum over rounds of
             ceil(live / 64) -- the bound for EVERY scheme that regroups rays inside a workgroup.

Costs in VALU wave-instructions: a round = R (62: the kernel's average over the GI stages); a hand-over EVENT (one wave parking
its survivors, or the collector loading 64 rays, or writing / fetching results) = E for the whole wave, whatever the number of
lanes -- the same rule that made every refill scheme lose (a transition issued for a few lanes costs what it costs for 64).
E = 40 is the optimistic figure (~18 state words per ray moved with ds_write_b128 / ds_read_b128, ballot + prefix, addresses);
DESIGN.md section 4 prices a swap at 25-35 without the stack, and a moved ray's stack rows (up to 9) have to follow it or be
addressed with bank conflicts.  The barrier a workgroup-level scheme needs (three waves idle while the collector walks) is NOT
priced: idle waves issue nothing, but they lower the occupancy that hides latency (5 instead of 6 waves per SIMD: +3 %).
ANALYSIS INFRASTRUCTURE: uses oracle/ only.

    python tests/tools/handover_study.py [--band 32] [--out profiles/r05/handover_study.txt]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

NAMES = ["primary", "shadow", "GI", "GI shadow"]
R = 62.0                      # VALU wave-instructions per round (GI stages, measured average)
KERNEL_VALU_PER_TILE_SAMPLE = 0.934e9 / (32400 * 4)      # rocprofv3 SQ_INSTS_VALU per C3 launch / (tiles x samples)


def cut_stage(ln, T):
    """ln (..., 64) executed iterations of one wave-stage.  The wave walks until at most T lanes are live.
    Returns (rounds walked, remaining iterations of the survivors (..., 64), survivors per wave)."""
    srt = -np.sort(-ln, axis=-1)                       # descending
    # live lanes after r rounds = #(ln > r); the wave stops at the first r with live <= T: r = (T+1)-th largest length
    cut = srt[..., T] if T < 64 else np.zeros(ln.shape[:-1], ln.dtype)
    if T == 0:
        cut = srt[..., 0]
    rem = np.maximum(ln - cut[..., None], 0)
    return cut, rem, (rem > 0).sum(axis=-1)


def pool_walks(rem):
    """rem (groups, n) remaining iterations of the pooled rays of each group (zeros = no ray).  The collector walks them
    compacted, 64 at a time, longest first (it cannot know the lengths; any order costs at least this): returns (rounds,
    walks)."""
    srt = -np.sort(-rem, axis=-1)
    n = srt.shape[-1]
    rounds = np.zeros(srt.shape[0])
    walks = np.zeros(srt.shape[0])
    for k in range(0, n, 64):
        head = srt[:, k]
        rounds += head
        walks += head > 0
    return rounds, walks


def pool_walks_random(rem, rng):
    """the same with the rays in arrival order (what a real pool gives): 64 at a time as they come"""
    g, n = rem.shape
    rounds = np.zeros(g)
    walks = np.zeros(g)
    for i in range(g):
        r = rem[i][rem[i] > 0]
        rng.shuffle(r)
        for k in range(0, len(r), 64):
            rounds[i] += r[k:k + 64].max()
            walks[i] += 1
    return rounds, walks


def walk_blocks(ops, start, length, lod, T=-1):
    """The walk loop priced block by block (csrc/vrc_walk.h; tests/tools/path_schedule.py: header 13 VALU, 15 with the LOD
    test; descend 25, advance 15, pop 22 -- a block is issued when ANY live lane needs it).  ops (n, 64, L) op codes per
    iteration, start / length (n, 64): first iteration to execute / iterations of the ray (0 = no ray).  A wave stops when at
    most T lanes are live (T = -1: runs to the end).  Returns (VALU cost per wave, rounds per wave, next iteration per lane)."""
    from path_schedule import H_, D_, A_, P_
    n, _, L = ops.shape
    idx = np.minimum(start, length).astype(np.int64)
    cost = np.zeros(n)
    rounds = np.zeros(n)
    h = H_ + (2.0 if lod else 0.0)
    rows = np.arange(n)[:, None], np.arange(64)[None, :]
    running = np.ones(n, bool)
    while True:
        live = (idx < length) & running[:, None]
        nlive = live.sum(axis=1)
        running &= nlive > max(T, 0)
        live &= running[:, None]
        if not running.any():
            break
        cur = ops[rows[0], rows[1], np.minimum(idx, L - 1)]
        d = (live & (cur == 0)).any(axis=1)
        a = (live & ((cur == 1) | (cur == 2))).any(axis=1)
        p = (live & (cur == 2)).any(axis=1)
        cost += running * h + d * D_ + a * A_ + p * P_
        rounds += running
        idx += live
    return cost, rounds, idx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--band", type=int, default=32)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--pitch", type=float, default=-0.5)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import tile_shape as TS
    import cpuvoxelraycaster_amd as vrc
    import oracle_lib as O
    depth, W, Hh, spp = 9, 1920, 1080, 4
    nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
    tex = O.load_textures()
    L = O.lib()
    L.orc_render_frame_optrace_geo.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.POINTER(O.Camera), C.POINTER(O.FrameParams),
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    tr, _op, sk = TS.trace_pose(L, O, vrc, nodes, tex, depth, W, Hh, a.pitch, 0.0, a.band, a.threads, spp)
    ln = np.maximum(tr - np.minimum(sk, tr), 0)                    # (spp, nb, 8, W, 4) executed iterations
    S, nb = ln.shape[0], ln.shape[1]
    t = ln.reshape(S, nb, 8, W // 8, 8, 4)
    t = np.moveaxis(t, 3, 2).reshape(S, nb * (W // 8), 64, 4)     # (sample, tile, lane, stage)
    ntiles = t.shape[1] // 4 * 4
    t = t[:, :ntiles]
    lines = []

    def say(s=""):
        print(s)
        lines.append(s)

    base = t.max(axis=2).astype(np.float64)                         # (sample, tile, stage) rounds now
    per_ts = base.mean(axis=(0, 1))
    total_rounds = per_ts.sum()
    say(f"C3 frame, pitch {a.pitch}, band 1/{a.band}: {ntiles} tiles x {S} samples; rounds per tile-sample now: "
        + ", ".join(f"{NAMES[k]} {per_ts[k]:.1f}" for k in range(4)) + f"; all {total_rounds:.1f}")
    util = [t[..., k].sum() / (base[..., k].sum() * 64.0) for k in range(4)]
    say("lane utilisation of the stages: " + ", ".join(f"{NAMES[k]} {util[k]:.2f}" for k in range(4)))
    say(f"a round = {R:.0f} VALU wave-instructions; the kernel issues {KERNEL_VALU_PER_TILE_SAMPLE:.0f} per tile-sample in all (rocprofv3), "
        f"so 1 % of the frame = {KERNEL_VALU_PER_TILE_SAMPLE / 100 / R:.2f} rounds per tile-sample")
    say()

    def pct(saved_rounds_per_ts, events_per_ts, E):
        return (saved_rounds_per_ts * R - events_per_ts * E) / KERNEL_VALU_PER_TILE_SAMPLE * 100.0

    rng = np.random.default_rng(5)
    # ---- ideal: continuous perfect compaction over the 4 waves of a workgroup, per stage
    say("ideal(wg): continuous perfect compaction over a workgroup's 4 wave-stages, zero cost (bound for any regrouping inside a workgroup)")
    ideal_saved = 0.0
    for k in (2, 3):
        g = t[..., k].reshape(S, ntiles // 4, 256)                  # 4 consecutive tiles = one workgroup's stage
        mx = int(g.max())
        rounds = np.zeros(g.shape[:2])
        for r in range(mx):
            rounds += np.ceil((g > r).sum(axis=-1) / 64.0)
        now = base[..., k].reshape(S, ntiles // 4, 4).sum(axis=-1)
        saved = (now - rounds).mean() / 4.0
        ideal_saved += saved
        say(f"  {NAMES[k]:10s} {now.mean() / 4:.2f} -> {rounds.mean() / 4:.2f} rounds per tile-sample (-{saved:.2f})")
    say(f"  both GI stages: -{ideal_saved:.2f} rounds per tile-sample = {pct(ideal_saved, 0, 0):.1f} % of the frame's VALU instructions, gross, at zero cost")
    say()

    for scheme in ("wg", "unit"):
        say({"wg": "wg(T): four waves of a workgroup (4 consecutive tiles, same sample) hand their stragglers to one collector wave",
             "unit": "unit(T): one wave parks the stragglers of its unit's 4 samples and walks them together (GI stages first, then GI-shadow stages)"}[scheme])
        say(f"  {'T':>3s}  {'rounds saved / tile-sample':>27s}  {'events / tile-sample':>20s}  " + "  ".join(f"net % at E={E:>3d}" for E in (0, 40, 80, 120)))
        for T in (2, 4, 8, 12, 16, 24, 32):
            saved_ts = 0.0
            events_ts = 0.0
            for k in (2, 3):
                x = t[..., k]                                       # (sample, tile, lane)
                if scheme == "wg":
                    g = x.reshape(S, ntiles // 4, 4, 64)
                else:
                    g = np.moveaxis(x, 0, 1).reshape(ntiles, 1, S, 64)          # group = the 4 samples of one tile
                    g = g.reshape(1, ntiles, S, 64)
                cut, rem, nsurv = cut_stage(g, T)                   # (.., 4), (.., 4, 64), (.., 4)
                pooled = rem.reshape(-1, 4 * 64)
                prounds, pwalks = pool_walks_random(pooled, rng)
                now = g.max(axis=-1).sum(axis=-1).reshape(-1)
                new = cut.sum(axis=-1).reshape(-1) + prounds
                parked = (nsurv > 0).sum(axis=-1).reshape(-1)       # waves (or samples) that had survivors to hand over
                # events: each parker writes its survivors and later fetches their hits (2 each), the collector loads and
                # writes back per walk (2 each)
                ev = 2.0 * parked + 2.0 * pwalks
                saved_ts += (now - new).mean() / 4.0
                events_ts += ev.mean() / 4.0
            say(f"  {T:3d}  {saved_ts:27.2f}  {events_ts:20.2f}  " + "  ".join(f"{pct(saved_ts, events_ts, E):+13.1f}" for E in (0, 40, 80, 120)))
        say()
    # ---- the same unit(T) scheme priced block by block: the rounds a cut saves are TAIL rounds (few live lanes want few of the
    # loop's blocks), the rounds a pool walk adds are rounds of 64 unrelated rays (every block, every round)
    say("unit(T) priced block by block (walk VALU wave-instructions per tile-sample, GI + GI-shadow stages; header 15, descend 25, advance 15, pop 22):")
    LEN = _op.shape[-1]
    o = _op.reshape(S, nb, 8, W // 8, 8, 4, LEN)
    o = np.moveaxis(o, 3, 2).reshape(S, nb * (W // 8), 64, 4, LEN)[:, :ntiles]
    skt = sk.reshape(S, nb, 8, W // 8, 8, 4)
    skt = np.moveaxis(skt, 3, 2).reshape(S, nb * (W // 8), 64, 4)[:, :ntiles]
    trt = tr.reshape(S, nb, 8, W // 8, 8, 4)
    trt = np.moveaxis(trt, 3, 2).reshape(S, nb * (W // 8), 64, 4)[:, :ntiles]
    base_cost = 0.0
    per_stage_now = {}
    for k in (2, 3):
        c, r, _ = walk_blocks(o[..., k, :].reshape(-1, 64, LEN), skt[..., k].reshape(-1, 64), trt[..., k].reshape(-1, 64), lod=True)
        per_stage_now[k] = (c.sum() / (S * ntiles), r.sum() / (S * ntiles))
        base_cost += c.sum() / (S * ntiles)
    say("  now: " + ", ".join(f"{NAMES[k]} {per_stage_now[k][0]:.0f} VALU in {per_stage_now[k][1]:.1f} rounds ({per_stage_now[k][0] / per_stage_now[k][1]:.1f} per round)" for k in (2, 3))
        + f"; both {base_cost:.0f}")
    say(f"  {'T':>3s}  {'walk VALU / tile-sample':>24s}  {'saved':>7s}  {'pool walks / unit':>17s}  {'VALU per pool round':>19s}  " + "  ".join(f"net % at E={E:>3d}" for E in (0, 40, 80, 120)))
    for T in (4, 8, 12, 16, 24, 32):
        tot = 0.0
        events_ts = 0.0
        pw_n = 0.0
        pool_cost_sum = pool_round_sum = 0.0
        for k in (2, 3):
            ok = np.moveaxis(o[..., k, :], 0, 1)                    # (tile, sample, 64, LEN): a unit = the 4 samples of a tile
            st = np.moveaxis(skt[..., k], 0, 1)
            lk = np.moveaxis(trt[..., k], 0, 1)
            c, r, idx = walk_blocks(ok.reshape(-1, 64, LEN), st.reshape(-1, 64), lk.reshape(-1, 64), lod=True, T=T)
            tot += c.sum()
            idx = idx.reshape(ntiles, S * 64)
            okf = ok.reshape(ntiles, S * 64, LEN)
            lkf = lk.reshape(ntiles, S * 64)
            surv = idx < lkf
            parked = surv.reshape(ntiles, S, 64).any(axis=2).sum()
            # the unit's survivors, 64 at a time in arrival order
            pools_o, pools_s, pools_l = [], [], []
            for ti in range(ntiles):
                w = np.flatnonzero(surv[ti])
                for b in range(0, len(w), 64):
                    sel = w[b:b + 64]
                    po = np.zeros((64, LEN), np.uint8); ps = np.zeros(64, np.int64); pl = np.zeros(64, np.int64)
                    po[:len(sel)] = okf[ti, sel]; ps[:len(sel)] = idx[ti, sel]; pl[:len(sel)] = lkf[ti, sel]
                    pools_o.append(po); pools_s.append(ps); pools_l.append(pl)
            if pools_o:
                pc, pr, _ = walk_blocks(np.stack(pools_o), np.stack(pools_s), np.stack(pools_l), lod=True)
                tot += pc.sum()
                pool_cost_sum += pc.sum(); pool_round_sum += pr.sum()
            pw_n += len(pools_o)
            events_ts += (2.0 * parked + 2.0 * len(pools_o)) / (S * ntiles)
        tot /= S * ntiles
        saved = base_cost - tot
        say(f"  {T:3d}  {tot:24.0f}  {saved:7.0f}  {pw_n / ntiles:17.2f}  {pool_cost_sum / max(1.0, pool_round_sum):19.1f}  "
            + "  ".join(f"{(saved - events_ts * E) / KERNEL_VALU_PER_TILE_SAMPLE * 100.0:+13.1f}" for E in (0, 40, 80, 120)))
    say()
    say("Reading: net % = (rounds saved x 62 - events x E) / the kernel's VALU instructions per tile-sample.  Not priced, all against the scheme:")
    say("  the barrier of wg(T) (idle waves lower the occupancy that hides latency), the stack rows that have to follow a moved ray or be read")
    say("  with bank conflicts, the registers the hand-over code needs in a kernel at its budget (80 VGPRs, 6 waves per SIMD), and, for unit(T),")
    say("  the GI hits of four samples parked until their shadow rays run (the GI-shadow ray then cannot start on the GI walk's stack).")
    if a.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, a.out)), exist_ok=True)
        with open(os.path.join(ROOT, a.out), "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
