#!/usr/bin/env python3
"""N1 measurement: LSVO build time, GPU builder vs host builder vs the oracle's
pointer-tree restatement (what the reference does), per scene size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cpuvoxelraycaster_amd as vrc
import oracle_lib as O

h = O.load_terrain_heights()
print("depth nodes      bytes(MiB)  gpu_kernels_ms  gpu_call_ms  host_builder_s  oracle_setcell+compile_s  GB/s(out)")
for depth in (7, 8, 9, 10):
    vrc.LSVO.fromTerrain(h, depth)  # warm
    t = time.perf_counter(); svo = vrc.LSVO.fromTerrain(h, depth); tc = time.perf_counter() - t
    t = time.perf_counter(); vrc.build_terrain_lsvo(h, depth); th = time.perf_counter() - t
    to = float("nan")
    if depth <= 9:
        t = time.perf_counter(); O.compile_terrain(depth, h); to = time.perf_counter() - t
    mib = svo.n_nodes * 8 / 2**20
    print(f"{depth:5d} {svo.n_nodes:10d} {mib:10.1f} {svo.build_ms:14.3f} {tc * 1e3:12.2f} {th:15.2f} {to:25.2f} {svo.n_nodes * 8 / svo.build_ms / 1e6:9.1f}")
