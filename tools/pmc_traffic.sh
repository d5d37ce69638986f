#!/bin/bash
# HBM/fabric traffic of the frame kernel, per launch: FETCH_SIZE and WRITE_SIZE in separate
# --pmc passes (MI355X_MICROARCH.md, HBM section), plus the request-size breakdown that
# FETCH_SIZE is derived from.  Usage: tools/pmc_traffic.sh <tag> [bench args]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
# (the profiler starts the HIP runtime before bench.py can set this: set it here so that profiled and timed runs map
# their streams onto hardware queues the same way)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1; }
BENCH_ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
cd $R
python3 tools/pmc_summary.py $OUT k_render | cut -c1-100
