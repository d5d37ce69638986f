#!/usr/bin/env python3
"""Long differential fuzz on the GPU box: random scenes / rays / frame configurations, HIP path
vs the oracle, for a given number of seconds.  Any mismatch is printed and makes the exit code 1."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--quad-bias", type=float, default=0.0,
                    help="probability that a frame case is steered to a launch the quadrant walks apply to (tree of 8+ levels, pinhole camera, "
                         "sample mode, units of a multiple of four samples, no reuse / checkerboard): the unbiased draw gets there in ~0.1 %% of the frames")
    ap.add_argument("--depth10", type=float, default=0.0,
                    help="probability that a steered terrain case runs on the 1024^3 tree (10 levels, BASELINE config 5's: the deepest stack "
                         "the frame kernels are built for; generated once and kept) instead of 256^3 / 512^3 -- no draw at 0: old seeds replay")
    args = ap.parse_args()
    deep = {}                                                    # the 1024^3 terrain and its 324 MB of nodes on the host, made once
    import collections
    kernels = collections.Counter()
    import cpuvoxelraycaster_amd as vrc
    import oracle_lib as O
    import raygen
    L = vrc.capi.load()
    top, side = vrc.load_textures()
    heights = O.load_terrain_heights()
    t0 = time.time()
    it = bad = rays_total = frames = 0
    while time.time() - t0 < args.seconds:
        rng = np.random.default_rng(args.seed * 100003 + it)
        it += 1
        if it % 3:
            # random volume, random rays with LOD parameters
            depth = int(rng.integers(2, 8))
            S = 1 << depth
            vol = (rng.random((S, S, S)) < float(rng.choice([0.001, 0.01, 0.05, 0.2, 0.6, 1.0]))).astype(np.uint8)
            svo = vrc.LSVO.fromVolume(vol, depth)
            nodes = svo.downloadNodes()
            ref_nodes = O.compile_voxels(depth, np.argwhere(vol)) if depth <= 5 else nodes
            if not np.array_equal(nodes.view(np.uint64), ref_nodes.view(np.uint64)):
                print("BUILDER MISMATCH", it, depth); bad += 1
            n = int(rng.integers(1, 60000))
            org, d = raygen.mixed_rays(max(n, 16), seed=int(rng.integers(1 << 30)))
            coef = rng.choice([0.0, 0.0, 0.3, 0.5, 2.0], size=len(org)).astype(np.float32)
            bias = rng.choice([0.0, 0.0, 0.0, 0.01], size=len(org)).astype(np.float32)
            got = svo.castRays(org, d, coef=coef, bias=bias)
            ref = O.cast_rays(nodes, depth, org, d, coef=coef, bias=bias, threads=16)
            rays_total += len(org)
            if got.tobytes() != ref.tobytes():
                print("RAY MISMATCH", it, depth, len(org)); bad += 1
        else:
            # random frame configuration on the terrain, random lane map
            quad_case = bool(args.quad_bias > 0 and rng.random() < args.quad_bias)     # (no draw at bias 0: old seeds replay)
            if rng.integers(0, 2):
                depth = int(rng.choice([8, 9] if quad_case else [7, 8, 9]))
                if quad_case and args.depth10 > 0 and rng.random() < args.depth10:
                    depth = 10
                    if not deep:
                        deep["svo"] = vrc.LSVO.fromFastNoiseTerrain(10, textures=(top, side))
                        deep["nodes"] = deep["svo"].downloadNodes()
                    svo = deep["svo"]
                else:
                    svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=(top, side))
            else:
                # random blobs / slabs / noise volume: frames through arbitrary geometry, camera anywhere
                depth = int(rng.integers(4, 8))
                S0 = 1 << depth
                g = np.indices((S0, S0, S0)).astype(np.float32)
                vol = np.zeros((S0, S0, S0), np.uint8)
                for _ in range(int(rng.integers(1, 6))):
                    c = rng.uniform(0, S0, 3)
                    r = rng.uniform(1, S0 / 3)
                    vol |= (((g[0] - c[0]) ** 2 + (g[1] - c[1]) ** 2 + (g[2] - c[2]) ** 2) < r * r).astype(np.uint8)
                vol |= (rng.random((S0, S0, S0)) < 0.01).astype(np.uint8)
                if rng.integers(0, 2):
                    vol[:, int(rng.integers(0, S0)), :] = 1
                if quad_case:
                    # the same geometry somewhere in the middle of a 256^3 cube: a tree of 8 levels
                    big = np.zeros((256, 256, 256), np.uint8)
                    o = [int(rng.integers(max(0, 128 - S0), min(256 - S0, 128) + 1)) for _ in range(3)]
                    big[o[0]:o[0] + S0, o[1]:o[1] + S0, o[2]:o[2] + S0] = vol
                    vol, depth = big, 8
                svo = vrc.LSVO.fromVolume(vol, depth, textures=(top, side))
            nodes = deep["nodes"] if svo is deep.get("svo") else svo.downloadNodes()
            W, H = int(rng.integers(1, 200)), int(rng.integers(1, 120))
            spp = int(rng.choice([4, 4, 8, 12, 16])) if quad_case else 0
            chunk = int(rng.choice([4, 8, spp])) if quad_case else int(rng.integers(0, 9))
            vrc.capi.check(L.vrc_set_sample_chunk(chunk))                       # 0 = automatic ... whole-spp units
            rc = vrc.RayCaster(svo, (W, H))
            lane_samples = int(rng.choice([0, 1])) if quad_case else int(rng.choice([0, 1, 4]))                          # the lane <-> (pixel, sample) map (4 applies to spp % 4 == 0)
            rc.setLaneSamples(lane_samples)
            # a quarter of the cases with the invariant-ray reuse (pinhole camera + stage-synchronous kernel: fewer walks
            # executed, same frame; anything else: no effect at all)
            reuse = bool(rng.integers(0, 4) == 0) and not quad_case
            rc.setInvariantRayReuse(reuse)
            # the quadrant walks of the pinhole kernels (on by default where a launch allows it: trees of 8+ levels, units of a
            # multiple of four samples): a third of the cases with the plain layout instead
            if rng.integers(0, 3) == 0 and not quad_case:
                rc.setQuadWalks(False)
            # the 6 / 7 / 8 waves-per-SIMD builds of the stage-synchronous kernel (0 = the library's choice)
            blocks = int(rng.choice([0, 0, 6, 7, 8]))
            if blocks:
                rc.setTuning(blocks_per_cu=blocks)
            S = float(1 << depth)
            light = vrc.reference_light(depth)
            rc.setLightPosition(light)
            use_samples = 1 if quad_case else int(rng.integers(0, 2))
            rc.use_gi, rc.use_samples = bool(rng.integers(0, 2)), bool(use_samples)
            rc.shadow_samples = int(rng.choice([0, 1, 2]))
            rc.gi_bounces = int(rng.choice([1, 1, 2]))
            rc.seed = int(rng.integers(1 << 31))
            pos = (S / 2 + rng.uniform(-S / 8, S / 8), S / 2 - rng.uniform(56, 56 + S / 8), S / 2 + rng.uniform(-S / 8, S / 8))
            cam = vrc.make_camera(pos, vrc.make_rotation(float(rng.uniform(-3, 3)), float(rng.uniform(-1.3, 0.3))),
                                  fov=float(rng.uniform(0.5, 1.5)), aperture=0.0 if quad_case else float(rng.choice([0.0, 0.3, 2.0])),
                                  focal_length=float(rng.uniform(1, 120)))
            # the light of main.cpp:124 most of the time; sometimes anywhere -- inside the cube, far away, on a voxel
            # boundary -- so that the shadow / GI-shadow directions and their normalisation leave the usual range
            lk = int(rng.integers(0, 8))
            if lk == 0:
                light = rng.uniform(1.0, 2.0, 3).astype(np.float32)
            elif lk == 1:
                light = (rng.normal(size=3) * 10.0 ** rng.uniform(0, 30)).astype(np.float32)
            elif lk == 2:
                light = (np.round(rng.uniform(1.0, 2.0, 3) * S) / S).astype(np.float32)
            if lk <= 2:
                rc.setLightPosition(light)
            if not quad_case:
                spp = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 6, 8, 8, 12]))
            nfr = int(rng.integers(1, 3))
            checker = bool(rng.integers(0, 2)) and not quad_case
            # a third of the frames are one rank's shard of a multi-GPU frame
            shard_count = int(rng.choice([1, 1, 2, 3, 5]))
            shard_index = int(rng.integers(0, shard_count))
            row_block = int(rng.choice([8, 8, 16, 24])) if shard_count > 1 else 0
            ocam = O.make_camera(tuple(cam.position), tuple(cam.rot), cam.fov, cam.aperture, cam.focal_length)
            oimg = oacc = None
            rr = ss = 0
            fi = 0
            # half of the sample-mode frames without checkerboard go through the fused call (render + resolve + reset in
            # one launch): every call's image must be the oracle's resolve of THAT call's samples, accumulators zero after
            fused = bool(use_samples and not checker and rng.integers(0, 2))
            ok = True
            for f in range(nfr):
                par = (f % 2) if checker else -1
                if fused:
                    rc.renderFrameResolved(cam, spp=spp, row_block=row_block, shard_index=shard_index, shard_count=shard_count)
                    oacc = None
                else:
                    rc.renderFrame(cam, spp=spp, checker_parity=par, row_block=row_block, shard_index=shard_index, shard_count=shard_count)
                for s in range(spp):
                    p = O.make_params(W, H, light, use_gi=int(rc.use_gi), use_samples=use_samples, shadow_samples=rc.shadow_samples,
                                      gi_bounces=rc.gi_bounces, checker_parity=par, seed=rc.seed, frame_index=fi,
                                      row_block=row_block, shard_index=shard_index, shard_count=shard_count)
                    oimg, oacc, _, st = O.render_frame(nodes, depth, top, side, ocam, p, image=oimg, accum=oacc, threads=16)
                    rr += st.rays; ss += st.sum_complexity; fi += 1
                if fused:
                    mine = oacc[..., 3] > 0                                 # this shard's pixels
                    ok = ok and np.array_equal(rc.readImage()[mine], O.samples_to_image(oacc)[mine]) and not rc.readAccum().any()
            st = rc.stats()
            kernels[rc.lastKernel() + (" @ depth 10" if depth == 10 else "")] += 1
            if reuse and cam.aperture == 0.0:
                ok = ok and st.rays <= rr and st.sum_complexity <= ss
            else:
                ok = ok and (st.rays, st.sum_complexity) == (rr, ss)
            if not fused:
                ok = ok and (np.array_equal(rc.readAccum(), oacc) if use_samples else np.array_equal(rc.readImage(), oimg))
            frames += 1
            rays_total += rr
            if not ok:
                print("FRAME MISMATCH", it, dict(depth=depth, W=W, H=H, lane_samples=lane_samples, kernel=rc.lastKernel(), spp=spp, gi=rc.use_gi, samples=use_samples,
                                                 checker=checker, fused=fused, reuse=reuse, blocks=blocks, shard=(row_block, shard_index, shard_count))); bad += 1
            vrc.capi.check(L.vrc_set_sample_chunk(0))
    print("frame kernels of the last launch of each frame case:", dict(sorted(kernels.items())))
    print(f"fuzz: {it} cases ({frames} frames), {rays_total} rays compared, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
