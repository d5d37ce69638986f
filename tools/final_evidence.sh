#!/bin/bash
# round-4 final evidence on the box
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q > gpurun_out/r04/gpu_tests.log 2>&1; tail -3 gpurun_out/r04/gpu_tests.log
python bench.py > gpurun_out/r04/final_bench.json.log 2> gpurun_out/r04/final_bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/final_bench_steps20.json.log 2>/dev/null; echo "bench steps20 rc=$?"
bash tools/pmc_roofline.sh c3 > gpurun_out/r04/pmc_c3.log 2>&1
bash tools/pmc_roofline.sh c3_inflight1 --inflight 1 > gpurun_out/r04/pmc_c3_inflight1.log 2>&1
bash tools/pmc_roofline.sh c4 --config c4 > gpurun_out/r04/pmc_c4.log 2>&1
tail -3 gpurun_out/r04/pmc_c3.log | cut -c1-300
