#!/usr/bin/env python3
"""Kernel time of ONE rank's shard of the C3 frame for N = 1, 2, 4, 8 (emulated on one GPU)
versus workgroups per CU: how well does the per-GPU work pack when the frame is split?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import cpuvoxelraycaster_amd as vrc

cfg = bench.CONFIGS["c3"]
depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
rc = vrc.RayCaster(svo, (W, H))
rc.setLightPosition(vrc.reference_light(depth))
rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
cam = vrc.reference_camera(depth)
stream = torch.cuda.current_stream().cuda_stream
L = vrc.capi.load()
print("N  bpc  median_ms   (ideal = t(N=1)/N)")
base = None
for N in (1, 2, 4, 8):
    for bpc in [int(v) for v in os.environ.get('BPCS', '3,5').split(',')]:
        rc.setTuning(bpc)
        ts = []
        for it in range(8):
            rc.resetSamples(stream)
            rc.frame_index = 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc.renderFrame(cam, spp=spp, stream=stream, row_block=8, shard_index=0, shard_count=N)
            e1.record()
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(e0.elapsed_time(e1))
        ts.sort()
        t = ts[len(ts) // 2]
        if N == 1 and base is None:
            base = t
        print(f"{N}  {bpc}   {t:8.3f}" + (f"   ideal {base / N:6.3f}  eff {base / N / t:5.2f}" if base else ""))
