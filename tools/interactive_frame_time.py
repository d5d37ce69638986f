#!/usr/bin/env python3
"""Kernel time of the reference's own interactive frame (src/main.cpp:131-158 as shipped): 960x540, checkerboard (every
other 4x4 tile group per frame), one sample per pixel, temporal blend into the image (no sample accumulation), GI on/off."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cpuvoxelraycaster_amd as vrc

depth, W, H = 9, 960, 540
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
stream = torch.cuda.current_stream().cuda_stream
for use_gi in (False, True):
    rc = vrc.RayCaster(svo, (W, H))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples = use_gi, False
    cam = vrc.reference_camera(depth, pitch=-0.5)
    ts = []
    for f in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc.renderFrame(cam, spp=1, checker_parity=f % 2, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        if f >= 8:
            ts.append(e0.elapsed_time(e1))
    st = rc.stats()
    ts.sort()
    print(f"960x540 checkerboard frame, GI {'on' if use_gi else 'off'}: median {ts[len(ts) // 2] * 1e3:.0f} us, "
          f"{st.rays / 40 / 1e6:.2f} M rays per frame")
