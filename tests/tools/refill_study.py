#!/usr/bin/env python3
"""Offline study (CPU, oracle only): would a GI-only refill scheduler beat the stage-synchronous GI stages at the walk's CURRENT
cost?  Jobs = (GI ray, GI-shadow ray) iteration counts of the C3 frame from the oracle's traces (orc_render_frame_trace).  A
persistent wave of 64 lanes runs the walk loop in lockstep rounds; a lane that ends a ray waits; when at least T lanes wait (or
nobody runs) the wave leaves the loop and issues the transition code for the waiting lanes -- GI hit -> shadow ray (T1
instructions) and / or shadow result -> next job's GI ray (T2) -- at full wave cost each, then re-enters the loop.  Costs in VALU
wave-instructions: a round = R (the GI stages' measured average), T1 / T2 = the transitions' sizes in the shipped kernel.
ANALYSIS INFRASTRUCTURE: uses oracle/ only.

    python tests/tools/refill_study.py [--trace /tmp/trace.npy]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def simulate(jobs, T, R, T1, T2, queue_cost=0.0, waves=200, seed=0):
    """jobs: (n, 2) iteration counts (GI, GI shadow; shadow 0 = the GI ray missed).  Returns VALU per job."""
    rng = np.random.default_rng(seed)
    total_cost = 0.0
    done_jobs = 0
    for w in range(waves):
        stream = jobs[rng.permutation(len(jobs))[:64 * 40]]        # 40 jobs per lane of this wave
        nxt = 64
        cur = stream[:64].copy()                                    # per lane (gi, gish) of its current job
        phase = np.zeros(64, np.int8)                               # 0 walking GI, 1 walking GI shadow
        rem = cur[:, 0].copy()
        wait1 = np.zeros(64, bool)
        wait2 = np.zeros(64, bool)
        alive = np.ones(64, bool)
        cost = 0.0
        cost += T2                                                  # the first jobs' set-up
        while alive.any():
            running = alive & ~wait1 & ~wait2
            nwait = int((wait1 | wait2).sum())
            if running.any() and nwait < T:
                cost += R
                rem[running] -= 1
                ended = running & (rem <= 0)
                e0 = ended & (phase == 0)
                has_shadow = cur[:, 1] > 0
                wait1 |= e0 & has_shadow
                wait2 |= (e0 & ~has_shadow) | (ended & (phase == 1))
                continue
            # transition event
            if wait1.any():
                cost += T1
                phase[wait1] = 1
                rem[wait1] = cur[wait1, 1]
                wait1[:] = False
            if wait2.any():
                cost += T2 + queue_cost
                idx = np.flatnonzero(wait2)
                done_jobs += len(idx)
                take = min(len(idx), len(stream) - nxt)
                if take:
                    cur[idx[:take]] = stream[nxt:nxt + take]
                    phase[idx[:take]] = 0
                    rem[idx[:take]] = cur[idx[:take], 0]
                    nxt += take
                alive[idx[take:]] = False
                wait2[:] = False
        total_cost += cost
    return total_cost / done_jobs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", default="")
    a = ap.parse_args()
    if a.trace and os.path.exists(a.trace):
        tr = np.load(a.trace).astype(np.int32)
    else:
        import stage_balance as sb
        import cpuvoxelraycaster_amd as vrc
        import oracle_lib as O
        depth, W, H = 9, 1920, 1080
        nodes = vrc.build_terrain_lsvo(O.load_terrain_heights(), depth)
        tex = O.load_textures()
        cam = vrc.reference_camera(depth, pitch=-0.5, make_rotation=lambda x, y: O.make_rotation(x, y))
        ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
        tr = np.stack([sb.tiles(sb.trace_frame(nodes, depth, tex, ocam, vrc.reference_light(depth), W, H, s, os.cpu_count() or 8), H, W)
                       for s in range(2)], axis=2)
        if a.trace:
            np.save(a.trace, tr.astype(np.uint16))
    T_, L, S, _ = tr.shape
    gi, gish = tr[..., 2], tr[..., 3]
    has = gi > 0                                                  # pixel-samples with a primary hit cast a GI ray
    jobs = np.stack([gi[has], gish[has]], axis=1)
    R, T1, T2 = 62.0, 200.0, 250.0
    # the shipped kernel: per tile-sample max over lanes per stage, transitions once per stage at full utilisation
    cur = tr.max(axis=1)
    rounds = (cur[..., 2] + cur[..., 3]).mean()
    jobs_per_tile_sample = has.sum() / (T_ * S)
    base = (rounds * R + T1 + T2) / jobs_per_tile_sample
    print(f"{len(jobs)} jobs; GI {jobs[:, 0].mean():.1f} + shadow {jobs[:, 1].mean():.1f} iterations per job; jobs per tile-sample {jobs_per_tile_sample:.1f}")
    print(f"stage-synchronous now: {rounds:.1f} rounds per tile-sample -> {base:.1f} VALU wave-instructions per job (round {R:.0f}, transitions {T1:.0f} + {T2:.0f})")
    for T in (1, 8, 16, 24, 32, 40, 48, 56):
        for qc in (0.0, 60.0):
            c = simulate(jobs, T, R, T1, T2, qc)
            print(f"  refill when >= {T:2d} lanes wait, queue / hand-over cost {qc:3.0f} per event: {c:6.1f} per job ({(c / base - 1) * 100:+5.1f} %)")


if __name__ == "__main__":
    main()
