import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import cpuvoxelraycaster_amd as vrc
cfgs = {"c3": (9, 1920, 1080, 4, 0.0), "c4": (9, 1920, 1080, 4, 0.5), "c5": (10, 3840, 2160, 16, 0.0)}
which = sys.argv[1:] or ["c3", "c4"]
tex = vrc.load_textures()
for name in which:
    depth, W, H, spp, ap = cfgs[name]
    svo = vrc.LSVO.fromFastNoiseTerrain(depth, seed=1337, device=0, textures=tex)
    light = vrc.reference_light(depth)
    for pitch in (-0.5, 0.0, -1.2):
        cam = vrc.reference_camera(depth, pitch=pitch, aperture=ap, focal_length=(60.0 * (1 << depth) / 512.0 if ap else 1.0))
        for F in (3, 1):
            for ns in (1, 4):
                rcs = []
                for _ in range(F):
                    r = vrc.RayCaster(svo, (W, H)); r.setLightPosition(light); r.use_gi, r.use_samples, r.shadow_samples = True, True, 1
                    r.setLaneSamples(ns)
                    if F > 1: r.setSampleChunk(spp)
                    rcs.append(r)
                streams = [torch.cuda.Stream() for _ in range(F)]
                def frames(i0, n):
                    for i in range(i0, i0 + n):
                        rc, ts = rcs[i % F], streams[i % F]
                        rc.frame_index = i * spp
                        rc.renderFrameResolved(cam, spp=spp, stream=ts.cuda_stream)
                frames(0, 6); torch.cuda.synchronize()
                n = 60 if name != "c5" else 6
                best = 1e9
                for rep in range(5):
                    t0 = time.perf_counter(); frames(6, n); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
                print(f"{name} pitch {pitch:5.1f} inflight {F} lane_samples {ns}: {best:.4f} ms  {rcs[0].lastKernel()}", flush=True)
                for r in rcs: r.close()
    svo.close()
