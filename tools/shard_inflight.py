#!/usr/bin/env python3
"""Frame period of ONE rank's shard of the C3 frame (render + resolve_shard, no collective) for N = 1, 2, 4, 8 with
1 and 2 frames in flight (renderers / streams used round-robin), emulated on one GPU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import cpuvoxelraycaster_amd as vrc

cfg = bench.CONFIGS["c3"]
depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
cam = vrc.reference_camera(depth, pitch=-0.5)
L = vrc.capi.load()
print("N  inflight  us/frame")
for N in (1, 2, 4, 8):
    for F in (1, 2, 3, 4):
        rcs = []
        for _ in range(F):
            rc = vrc.RayCaster(svo, (W, H))
            rc.setLightPosition(vrc.reference_light(depth))
            rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
            rcs.append(rc)
        streams = [torch.cuda.Stream() for _ in range(F)]
        bufs = [torch.empty(L.vrc_shard_bytes(W, H, 8, N), dtype=torch.uint8, device="cuda") for _ in range(F)]

        def frame(i):
            rc, s = rcs[i % F], streams[i % F]
            rc.frame_index = i * spp
            rc.renderFrame(cam, spp=spp, stream=s.cuda_stream, row_block=8, shard_index=0, shard_count=N)
            rc.resolveShard(8, 0, N, bufs[i % F].data_ptr(), reset=True, stream=s.cuda_stream)

        for i in range(20):
            frame(i)
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for i in range(n):
            frame(20 + i)
        torch.cuda.synchronize()
        print(f"{N}  {F}  {(time.perf_counter() - t0) / n * 1e6:8.1f}")
