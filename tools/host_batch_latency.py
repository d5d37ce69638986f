#!/usr/bin/env python3
"""Wall time of vrc_cast_rays with HOST buffers (staged through the library) for a few batch sizes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpuvoxelraycaster_amd as vrc

svo = vrc.LSVO.fromFastNoiseTerrain(9)
rng = np.random.default_rng(0)
for n in (1, 256, 4096, 65536, 1 << 20):
    org = rng.uniform(1.0, 2.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    for _ in range(5):
        svo.castRays(org, d)
    t = []
    for _ in range(50):
        t0 = time.perf_counter()
        svo.castRays(org, d)
        t.append((time.perf_counter() - t0) * 1e6)
    t.sort()
    print(f"n = {n:8d}: median {t[len(t) // 2]:9.1f} us  ({n / t[len(t) // 2]:.1f} Mrays/s through host buffers)")
