#!/bin/bash
# round-5 final evidence: the GPU suite, the bench lines of every configuration (kernel sources unchanged since round 4: the PMC
# records are carried forward by hash, profiles/traffic.json), the timed symbol one launch at a time, a capped fuzz
mkdir -p gpurun_out/r05
python -m pytest tests -q -m gpu > gpurun_out/r05/gputests_final.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r05/gputests_final.log
python bench.py > gpurun_out/r05/final_bench.json.log 2> gpurun_out/r05/final_bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/final_bench_steps20.json.log 2>/dev/null; echo "steps20 rc=$?"
for c in c2 c4 c5; do python bench.py --config $c --blocks-min-seconds 2 > gpurun_out/r05/final_bench_$c.json.log 2>/dev/null; echo "$c rc=$?"; done
python bench.py --inflight 1 --blocks-min-seconds 2 --no-cpu-baseline > gpurun_out/r05/final_bench_inflight1.json.log 2>/dev/null; echo "inflight1 rc=$?"
python bench.py --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 --blocks-min-seconds 2 --no-cpu-baseline > gpurun_out/r05/final_bench_timed_symbol_alone.json.log 2>/dev/null; echo "alone rc=$?"
VRC_FORCE_DIST=1 python bench.py --steps 20 --blocks-min-seconds 1 --no-cpu-baseline > gpurun_out/r05/force_dist_rccl_world1.json.log 2>/dev/null; echo "rccl world1 rc=$?"
timeout 380 python tests/tools/fuzz_gpu.py --seconds 300 --seed 5005 > gpurun_out/r05/fuzz_final_5min.log 2>&1; tail -2 gpurun_out/r05/fuzz_final_5min.log
