#!/bin/bash
# round-5 evidence, first part (verdict Next 1 + 2): GPU tests, the default bench line with roofline.kernel_ms_alone, the clean
# rocprofv3 record of the TIMED symbol one launch at a time, and the self-launched world-4 rehearsal
mkdir -p gpurun_out/r05
python -m pytest tests -q -m gpu -x > gpurun_out/r05/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r05/gputests.log
python bench.py > gpurun_out/r05/bench_default.json.log 2> gpurun_out/r05/bench_default.err; echo "bench rc=$?"
tools/pmc_roofline.sh c3_alone_wholespp --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 > gpurun_out/r05/pmc_c3_alone_wholespp.log 2>&1; tail -12 gpurun_out/r05/pmc_c3_alone_wholespp.log
VRC_DIST_BACKEND=gloo python3 bench.py --gpus 4 --steps 5 > gpurun_out/r05/self_launch_gloo4_root.json.log 2> gpurun_out/r05/self_launch_gloo4_root.err; echo "gloo4 rc=$?"
tail -c 600 gpurun_out/r05/self_launch_gloo4_root.err
VRC_DIST_BACKEND=gloo python3 bench.py --gpus 4 --steps 5 --gather peer > gpurun_out/r05/self_launch_gloo4_peer.json.log 2> gpurun_out/r05/self_launch_gloo4_peer.err; echo "gloo4 peer rc=$?"
tail -c 600 gpurun_out/r05/self_launch_gloo4_peer.err
