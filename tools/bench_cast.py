#!/usr/bin/env python3
"""Throughput of the batch per-ray operator (vrc_cast_rays = Volumetric::castRay,
T1) with ray and hit buffers resident in HBM: coherent camera rays and
incoherent random rays, 512^3 terrain.  One JSON line per ray set."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import cpuvoxelraycaster_amd as vrc
    import raygen
    depth = 9
    svo = vrc.LSVO.fromTerrain(vrc.terrain_heights(), depth)
    stream = torch.cuda.current_stream().cuda_stream
    sets = {}
    o, d = raygen.camera_rays(depth, 1920, 1080, -0.5)
    sets["camera 1920x1080 pitch -0.5 (row-major pixels)"] = (o, d)
    rng = np.random.default_rng(0)
    n = 1 << 21
    o2 = rng.uniform(1.0, 2.0, (n, 3)).astype(np.float32)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    sets["2M random rays inside the cube"] = (o2, d2)
    for name, (o, d) in sets.items():
        n = o.shape[0]
        to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
        out = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            svo.castRaysDevice(n, to.data_ptr(), td.data_ptr(), out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            svo.castRaysDevice(n, to.data_ptr(), td.data_ptr(), out.data_ptr(), stream=stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        hits = np.frombuffer(out.cpu().numpy().tobytes(), dtype=vrc.HIT_DTYPE)
        steps = int(hits["complexity"].sum())
        alg = 8 * steps + n * (24 + 48)   # one LNode per loop iteration + ray in + HitPoint out
        print(json.dumps({"op": "vrc_cast_rays", "rays": name, "n": n, "ms": round(ms, 4), "Mrays_per_s": round(n / ms / 1e3, 1),
                          "loop_iterations": steps, "algorithmic_GBps": round(alg / ms / 1e6, 1),
                          "hbm_frac": round(alg / ms / 1e6 / 8000.0, 4), "hit_rate": round(float((hits["hit"] != 0).mean()), 3)}))


if __name__ == "__main__":
    main()
