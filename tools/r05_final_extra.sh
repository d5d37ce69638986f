#!/bin/bash
# round 5, after the quadrant-walk kernels: the self-launched world-4 rehearsals on the C3 frame again (the gathered frame 0 of the
# sharded `_q` kernels against the oracle), and ten minutes of differential fuzz with both layouts of the invariant walks
mkdir -p gpurun_out/r05
VRC_DIST_BACKEND=gloo python3 bench.py --gpus 4 --steps 5 > gpurun_out/r05/self_launch_gloo4_root.json.log 2> gpurun_out/r05/self_launch_gloo4_root.err; echo "gloo4 root rc=$?"
VRC_DIST_BACKEND=gloo python3 bench.py --gpus 4 --steps 5 --gather peer > gpurun_out/r05/self_launch_gloo4_peer.json.log 2> gpurun_out/r05/self_launch_gloo4_peer.err; echo "gloo4 peer rc=$?"
VRC_DIST_BACKEND=gloo python3 bench.py --gpus 8 --steps 5 > gpurun_out/r05/self_launch_gloo8_root.json.log 2> gpurun_out/r05/self_launch_gloo8_root.err; echo "gloo8 root rc=$?"
timeout 700 python tests/tools/fuzz_gpu.py --seconds 600 --seed 5055 > gpurun_out/r05/fuzz_quad_10min.log 2>&1; tail -2 gpurun_out/r05/fuzz_quad_10min.log
