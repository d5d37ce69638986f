#!/bin/bash
# Round 5, closing run on the final kernel sources: the whole GPU suite, smoke(), and the differential fuzzer steered
# to launches the quadrant walks apply to (tests/tools/fuzz_gpu.py --quad-bias).  Output under gpurun_out/r05/.
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > gpurun_out/r05/gputests_closing.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r05/gputests_closing.log
cat gpurun_out/r05/gputests_closing.log
timeout 1000 python tests/tools/fuzz_gpu.py --seconds ${FUZZ_SECONDS:-900} --seed 5115 --quad-bias 0.7 > gpurun_out/r05/fuzz_quad_bias_15min.log 2>&1
tail -3 gpurun_out/r05/fuzz_quad_bias_15min.log
