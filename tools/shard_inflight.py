#!/usr/bin/env python3
"""Frame period of ONE rank's shard of the C3 frame (VRC_CONFIG=c4 / c5: of that configuration's; no collective), emulated on one GPU for N = 1, 2, 4, 8, against the
number of frames in flight (renderers / streams used round-robin) and the work-unit policy: the library's automatic sample
chunk (tuned for one launch at a time) with the resolve as a second kernel, or whole-spp units with the fused resolve
(what bench.py uses when frames overlap).  Each figure is the best of three timed blocks."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # four frame streams + the default stream would share hardware queues
import torch
import bench
import cpuvoxelraycaster_amd as vrc

CONFIG = os.environ.get("VRC_CONFIG", "c3")                     # c4: BASELINE config 4, the lens frame north_star shards over 8 GPUs
cfg = bench.CONFIGS[CONFIG]
depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
cam = vrc.reference_camera(depth, pitch=-0.5, aperture=cfg["aperture"], focal_length=(60.0 * (1 << depth) / 512.0 if cfg["aperture"] else 1.0))   # as bench.py
L = vrc.capi.load()
BLOCKS = int(os.environ.get("VRC_BLOCKS", "0"))
LANE_SAMPLES = int(os.environ.get("VRC_LANE_SAMPLES", "0"))     # A/B of the lane <-> (pixel, sample) map (0 = the library's choice)


def period(N, F, whole):
    rcs = []
    for _ in range(F):
        rc = vrc.RayCaster(svo, (W, H))
        rc.setLightPosition(vrc.reference_light(depth))
        rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
        if whole:
            rc.setSampleChunk(spp)
        if LANE_SAMPLES:
            rc.setLaneSamples(LANE_SAMPLES)
        if os.environ.get("VRC_QUAD_OFF") == "1":
            rc.setQuadWalks(False)                  # A/B of the quadrant walks (round 5)
        if BLOCKS:
            rc.setTuning(blocks_per_cu=BLOCKS)      # force the 6 / 7 / 8 waves-per-SIMD build (A/B of sync_waves_for's choice)
        rcs.append(rc)
    streams = [torch.cuda.Stream() for _ in range(F)]
    bufs = [torch.empty(L.vrc_shard_bytes(W, H, 8, N), dtype=torch.uint8, device="cuda") for _ in range(F)]

    def frame(i):
        rc, s = rcs[i % F], streams[i % F]
        rc.frame_index = i * spp
        if whole:
            rc.renderFrameResolved(cam, spp=spp, dst_ptr=bufs[i % F].data_ptr(), stream=s.cuda_stream, row_block=8, shard_index=0, shard_count=N)
        else:
            rc.renderFrame(cam, spp=spp, stream=s.cuda_stream, row_block=8, shard_index=0, shard_count=N)
            rc.resolveShard(8, 0, N, bufs[i % F].data_ptr(), reset=True, stream=s.cuda_stream)

    best = None
    i = 0
    for block in range(4):
        n = 60 if block == 0 else 200
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            frame(i)
            i += 1
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / n * 1e6
        if block and (best is None or t < best):
            best = t
    return best


period(1, 2, True)      # warm-up
print(f"config {CONFIG}: {cfg['label']}")
print("N  frames in flight: 1 2 3 4 | automatic chunk + resolve kernel  ||  whole-spp units + fused resolve      (us per frame)")
for N in (1, 2, 4, 8):
    a = [period(N, F, False) for F in (1, 2, 3, 4)]
    b = [period(N, F, True) for F in (1, 2, 3, 4)]
    print(f"{N}  " + " ".join(f"{v:7.1f}" for v in a) + "  ||  " + " ".join(f"{v:7.1f}" for v in b))
