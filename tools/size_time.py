#!/usr/bin/env python3
"""Kernel time vs amount of work (frame height, spp) for the default frame kernel: where does the
efficiency of small launches go?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cpuvoxelraycaster_amd as vrc
depth = 9
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
cam = vrc.reference_camera(depth)
stream = torch.cuda.current_stream().cuda_stream
print("W     H    spp   ms      ns_per_pixel_sample  rays")
for (W, H, spp) in [(1920, 8, 4), (1920, 16, 4), (1920, 32, 4), (1920, 64, 4), (1920, 136, 4), (1920, 136, 1), (1920, 272, 1), (1920, 1080, 4)]:
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
    ts = []
    for it in range(8):
        rc.resetSamples(stream); rc.frame_index = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc.renderFrame(cam, spp=spp, stream=stream); e1.record()
        torch.cuda.synchronize()
        if it >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort(); t = ts[len(ts)//2]
    st = rc.stats()
    print(f"{W:5d} {H:5d} {spp:3d} {t:8.3f} {t*1e6/(W*H*spp):10.2f}   {st.rays//8}")
