#!/bin/bash
# One SQ pass (instruction counts + lane utilisation) for bench.py. Usage: tools/pmc_quick.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcq_$TAG
mkdir -p $OUT
# (the profiler starts the HIP runtime before bench.py can set this: set it here so that profiled and timed runs map
# their streams onto hardware queues the same way)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 "$@" > $OUT/sq1.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT | cut -c1-75
