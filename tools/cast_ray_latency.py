#!/usr/bin/env python3
"""Latency of the single-ray operator vrc_cast_ray (Camera::getClosestPoint, main.cpp:115: once per frame before the
frame is dispatched), idle and with a C3 frame in flight on another stream."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import cpuvoxelraycaster_amd as vrc
    depth = 9
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
    cam = vrc.reference_camera(depth)
    S = np.float32(1 << depth)
    org = (np.array(cam.position[:], np.float32) / S + np.float32(1)).astype(np.float32)
    d = np.array([cam.rot[2], cam.rot[5], cam.rot[8]], np.float32)
    for _ in range(50):
        svo.castRay(org, d)

    def measure(n=2000):
        t = []
        for _ in range(n):
            t0 = time.perf_counter()
            svo.castRay(org, d)
            t.append((time.perf_counter() - t0) * 1e6)
        t.sort()
        return t[len(t) // 2], t[int(len(t) * 0.99)]

    med, p99 = measure()
    print(f"vrc_cast_ray idle: median {med:.1f} us, p99 {p99:.1f} us (python + ctypes call included)")
    rc = vrc.RayCaster(svo, (1920, 1080))
    rc.setLightPosition(vrc.reference_light(depth))
    rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
    st = torch.cuda.Stream()
    for _ in range(20):
        rc.renderFrame(cam, spp=4, stream=st.cuda_stream)
    med, p99 = measure(300)
    torch.cuda.synchronize()
    print(f"vrc_cast_ray beside C3 frames on another stream: median {med:.1f} us, p99 {p99:.1f} us")


if __name__ == "__main__":
    main()
