#!/bin/bash
# round-4 evidence, second part: the tests that failed once, the remaining configurations, a capped fuzz on the final sources
mkdir -p gpurun_out/r04
for i in 1 2 3; do timeout 400 python -m pytest tests/test_gpu_peer_write.py tests/test_gpu_rccl_cpp.py -q -x > gpurun_out/r04/peer_tests_$i.log 2>&1; tail -2 gpurun_out/r04/peer_tests_$i.log; done
python bench.py > gpurun_out/r04/final_bench.json.log 2> gpurun_out/r04/final_bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/final_bench_steps20.json.log 2>/dev/null; echo "steps20 rc=$?"
for c in c2 c4 c5; do python bench.py --config $c --blocks-min-seconds 2 > gpurun_out/r04/final_bench_$c.json.log 2>/dev/null; echo "$c rc=$?"; done
python bench.py --inflight 1 --blocks-min-seconds 2 --no-cpu-baseline > gpurun_out/r04/final_bench_inflight1.json.log 2>/dev/null
timeout 560 python tests/tools/fuzz_gpu.py --seconds 480 --seed 4004 > gpurun_out/r04/fuzz_final_8min.log 2>&1; tail -2 gpurun_out/r04/fuzz_final_8min.log
