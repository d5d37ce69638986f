#!/usr/bin/env python3
"""In-process sweep of the k_render tuning knobs (blocks/CU, refill threshold) on one GPU, interleaved rounds (cdna guide rule 24).  Prints a table."""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--bpc", default="4,8")
    ap.add_argument("--thr", default="1,8,16,32,48")
    ap.add_argument("--chunk", default="64")
    ap.add_argument("--pitch", type=float, default=-0.5)
    ap.add_argument("--kernel", default="2", help="frame kernels to sweep: 2 sync, 0 tile-phased, 1 mixed")
    args = ap.parse_args()
    import torch
    import bench
    import cpuvoxelraycaster_amd as vrc
    cfg = bench.CONFIGS[args.config]
    depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
    nodes = vrc.build_terrain_lsvo(vrc.terrain_heights(), depth)
    svo = vrc.LSVO(nodes, depth, textures=vrc.load_textures())
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples = bool(cfg["use_gi"]), True, 1
    cam = vrc.reference_camera(depth, pitch=args.pitch, aperture=cfg["aperture"])
    stream = torch.cuda.current_stream().cuda_stream
    L = vrc.capi.load()
    combos = list(itertools.product([int(v) for v in args.kernel.split(",")], [int(v) for v in args.bpc.split(",")],
                                    [int(v) for v in args.thr.split(",")], [int(v) for v in args.chunk.split(",")]))
    times = {c: [] for c in combos}
    for r in range(args.rounds + 1):
        for c in combos:
            rc.setKernel(c[0])
            rc.setTuning(*c[1:])
            for f in range(args.frames):
                rc.resetSamples(stream)
                rc.frame_index = 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc.renderFrame(cam, spp=spp, stream=stream)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[c].append(e0.elapsed_time(e1))
    st = rc.stats(reset=True)
    print(f"config {args.config}: rays/frame ~ {st.rays // ((args.rounds + 1) * args.frames * len(combos))}")
    print("krn bpc thr chunk   median_ms  min_ms")
    for c in combos:
        t = sorted(times[c])
        print(f"{c[0]:3d} {c[1]:3d} {c[2]:3d} {c[3]:5d}   {t[len(t) // 2]:8.3f} {t[0]:8.3f}")


if __name__ == "__main__":
    main()
