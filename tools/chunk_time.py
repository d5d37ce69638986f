#!/usr/bin/env python3
"""Kernel time of a frame against the sample-chunk size of k_render_sync's work units (tail-granularity study)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--chunks", default="0,1,2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--tails", default="", help="with --chunks 0: units per wave that get half the chunk at the end of the launch")
    args = ap.parse_args()
    import torch
    import bench
    import cpuvoxelraycaster_amd as vrc
    cfg = bench.CONFIGS[args.config]
    depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples = bool(cfg["use_gi"]), True, 1
    cam = vrc.reference_camera(depth, pitch=-0.5, aperture=cfg["aperture"])
    stream = torch.cuda.current_stream().cuda_stream
    L = vrc.capi.load()
    chunks = [int(c) for c in args.chunks.split(",")]
    if args.tails:
        chunks = [("tail", int(t)) for t in args.tails.split(",")]
    times = {c: [] for c in chunks}
    for r in range(args.rounds + 1):
        for c in chunks:
            if isinstance(c, tuple):
                rc.setSampleChunk(0)
                rc.setSampleChunk(0xffff0000 | c[1])
            else:
                rc.setSampleChunk(c)
            for f in range(3):
                rc.resetSamples(stream)
                rc.frame_index = 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc.renderFrame(cam, spp=spp, stream=stream)
                e1.record()
                torch.cuda.synchronize()
                if r > 0:
                    times[c].append(e0.elapsed_time(e1))
    for c in chunks:
        t = sorted(times[c])
        print(f"sample_chunk {c}: median {t[len(t) // 2]:.3f} ms  min {t[0]:.3f} ms")


if __name__ == "__main__":
    main()
