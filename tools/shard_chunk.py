#!/usr/bin/env python3
"""Frame period of one rank's shard (emulated, render + resolve_shard, frames in flight as bench.py uses them) against the
sample chunk of the work units: with the pinhole kernel's hoisted set-up a unit of two samples shares one primary /
shadow set-up, a unit of one sample does not -- but shorter units end a short launch better."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import cpuvoxelraycaster_amd as vrc

cfg = bench.CONFIGS["c3"]
depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
cam = vrc.reference_camera(depth, pitch=-0.5)
L = vrc.capi.load()
# warm the device up (clocks, allocator, code objects) before anything is timed
_w = vrc.RayCaster(svo, (W, H)); _w.setLightPosition(vrc.reference_light(depth)); _w.use_gi, _w.use_samples, _w.shadow_samples = True, True, 1
for _i in range(300):
    _w.renderFrame(cam, spp=spp)
    _w.resolveShard(reset=True)
torch.cuda.synchronize()
for N, F in ((1, 2), (2, 2), (4, 4), (8, 4)):
    for chunk in (0, 1, 2, 4):
        rcs = []
        for _ in range(F):
            rc = vrc.RayCaster(svo, (W, H)); rc.setLightPosition(vrc.reference_light(depth)); rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
            rc.setSampleChunk(chunk)
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
        n = 300
        t0 = time.perf_counter()
        for i in range(n):
            frame(20 + i)
        torch.cuda.synchronize()
        print(f"N={N} F={F} sample_chunk={chunk} ({'auto' if chunk == 0 else 'fixed'}): {(time.perf_counter() - t0) / n * 1e6:8.1f} us/frame")
