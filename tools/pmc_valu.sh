#!/bin/bash
# quick: VALU instruction count and lane utilisation of the frame kernel for one bench configuration
# usage: tools/pmc_valu.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_valu
rm -rf $OUT; mkdir -p $OUT
# (the profiler starts the HIP runtime before bench.py can set this: set it here so that profiled and timed runs map
# their streams onto hardware queues the same way)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 --inflight 1 "$@" > $OUT/sq.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT k_render
