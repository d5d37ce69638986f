#!/bin/bash
# PMC counter passes for bench.py on the GPU box (each pass its own run, with
# --kernel-trace only, as the pool requires).  Usage: tools/pmc_passes.sh <tag> [bench args...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
# (the profiler starts the HIP runtime before bench.py can set this: set it here so that profiled and timed runs map
# their streams onto hardware queues the same way)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1
}
BENCH_ARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH
run sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_IFETCH SQ_CYCLES
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
