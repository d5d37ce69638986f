#!/usr/bin/env python3
"""Frame period of one rank's shard (emulated on one GPU) with the resolve as a second kernel, fused into the frame
kernel (vrc_render_frame_resolved), and without any resolve."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
import cpuvoxelraycaster_amd as vrc
cfg = bench.CONFIGS["c3"]
depth, W, H, spp = cfg["depth"], cfg["width"], cfg["height"], cfg["spp"]
svo = vrc.LSVO.fromFastNoiseTerrain(depth, textures=vrc.load_textures())
cam = vrc.reference_camera(depth, pitch=-0.5)
L = vrc.capi.load()
for N, F in ((8, 4), (4, 4), (2, 2), (1, 2)):
    for with_resolve in ("two calls", "fused", "render only"):
        rcs = []
        for _ in range(F):
            rc = vrc.RayCaster(svo, (W, H)); rc.setLightPosition(vrc.reference_light(depth)); rc.use_gi, rc.use_samples, rc.shadow_samples = True, True, 1
            rcs.append(rc)
        streams = [torch.cuda.Stream() for _ in range(F)]
        bufs = [torch.empty(L.vrc_shard_bytes(W, H, 8, N), dtype=torch.uint8, device="cuda") for _ in range(F)]
        def frame(i):
            rc, s = rcs[i % F], streams[i % F]
            rc.frame_index = i * spp
            if with_resolve == "fused":
                rc.renderFrameResolved(cam, spp=spp, dst_ptr=bufs[i % F].data_ptr(), stream=s.cuda_stream, row_block=8, shard_index=0, shard_count=N)
                return
            rc.renderFrame(cam, spp=spp, stream=s.cuda_stream, row_block=8, shard_index=0, shard_count=N)
            if with_resolve == "two calls":
                rc.resolveShard(8, 0, N, bufs[i % F].data_ptr(), reset=True, stream=s.cuda_stream)
        for i in range(20): frame(i)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for i in range(n): frame(20 + i)
        torch.cuda.synchronize()
        print(f"N={N} F={F} {with_resolve:12s}: {(time.perf_counter() - t0) / n * 1e6:8.1f} us/frame")
