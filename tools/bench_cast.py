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

    # the chained form (vrc_cast_ray_chains): the camera rays as ray A, the shadow ray of every hit (raycaster.hpp:139,151-153)
    # as ray B, started below the root next to A's hit -- against the same pairs cast as two independent batches
    o, d = sets["camera 1920x1080 pitch -0.5 (row-major pixels)"]
    n = o.shape[0]
    to, td = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda()
    out_a = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    svo.castRaysDevice(n, to.data_ptr(), td.data_ptr(), out_a.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    ha = np.frombuffer(out_a.cpu().numpy().tobytes(), dtype=vrc.HIT_DTYPE)
    f = np.float32
    inv = f(1.0) / f(1 << depth)
    so = (ha["position"] + (ha["normal"] * inv) * f(0.001)).astype(f)
    light = np.asarray(vrc.reference_light(depth), f)
    sd = (light[None, :] - so).astype(f)
    sd = (sd * (f(1.0) / np.sqrt((sd * sd).sum(axis=1, dtype=f)).astype(f))[:, None]).astype(f)
    tso, tsd = torch.from_numpy(np.ascontiguousarray(so)).cuda(), torch.from_numpy(np.ascontiguousarray(sd)).cuda()
    out_b = torch.empty(n * 48, dtype=torch.uint8, device="cuda")
    skipped = torch.zeros(n, dtype=torch.int32, device="cuda")

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms_chain = timed(lambda: svo.castRayChainsDevice(n, to.data_ptr(), td.data_ptr(), tso.data_ptr(), tsd.data_ptr(), 0.0, out_a.data_ptr(),
                                                     out_b.data_ptr(), skipped.data_ptr(), stream=stream))
    hb = np.frombuffer(out_b.cpu().numpy().tobytes(), dtype=vrc.HIT_DTYPE).copy()
    sk = int(skipped.sum().item())

    def two():
        svo.castRaysDevice(n, to.data_ptr(), td.data_ptr(), out_a.data_ptr(), stream=stream)
        svo.castRaysDevice(n, tso.data_ptr(), tsd.data_ptr(), out_b.data_ptr(), stream=stream)
    ms_two = timed(two)
    hb2 = np.frombuffer(out_b.cpu().numpy().tobytes(), dtype=vrc.HIT_DTYPE)
    print(json.dumps({"op": "vrc_cast_ray_chains", "rays": "camera rays + the shadow ray of every hit, started below the root", "pairs": n,
                      "ms": round(ms_chain, 4), "Mrays_per_s": round(2 * n / ms_chain / 1e3, 1),
                      "ms_as_two_vrc_cast_rays_batches": round(ms_two, 4), "shadow_iterations": int(hb["complexity"].sum()),
                      "shadow_iterations_not_executed": sk, "chained_equals_independent": bool(hb.tobytes() == hb2.tobytes())}))


if __name__ == "__main__":
    main()
