#!/bin/bash
# Instruction-cache counters of the frame kernel (one rocprofv3 --pmc pass, kernel trace only): requests, hits, misses,
# fetches, and the cycles waves waited for an instruction.   usage: tools/pmc_icache.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/icache_$TAG
mkdir -p $OUT
# (the profiler starts the HIP runtime before bench.py can set this: set it here so that profiled and timed runs map
# their streams onto hardware queues the same way)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ --output-format csv -d $OUT/ic -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 "$@" > $OUT/ic.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $OUT/w -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 "$@" > $OUT/w.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT k_render > $OUT/summary.txt 2>&1
rm -rf $OUT/ic $OUT/w
cat $OUT/summary.txt
