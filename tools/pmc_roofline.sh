#!/bin/bash
# The PMC record behind bench.py's roofline.traffic / roofline.issue, taken on the GPU box: one rocprofv3 run per
# counter group (--kernel-trace only, as the pool requires; FETCH_SIZE and WRITE_SIZE in passes of their own, as
# MI355X_MICROARCH.md prescribes) plus a --stats run of the same command.
#   usage: tools/pmc_roofline.sh <tag> [bench args, e.g. --config c5]
# Writes gpurun_out/roofline_<tag>/{summary.txt,source_hash.txt,stats/...}; turn it into profiles/traffic.json with
#   python tools/update_traffic.py gpurun_out/roofline_<tag> <key, e.g. c3_n1> <profiles/rNN/file-to-keep.txt>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/roofline_$TAG
mkdir -p $OUT
BENCH_ARGS=("$@")
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_hash())" > $OUT/source_hash.txt
# (the profiler starts the HIP runtime before bench.py can set this: set it here so that profiled and timed runs map
# their streams onto hardware queues the same way)
export GPU_MAX_HW_QUEUES=8
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 "${BENCH_ARGS[@]}" > $OUT/$name.log 2>&1; }
run sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VSKIPPED SQ_INSTS_SMEM
run sq4 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run grbm GRBM_GUI_ACTIVE
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 "${BENCH_ARGS[@]}" > $OUT/stats.log 2>&1
cd $R
python3 tools/pmc_summary.py $OUT k_render > $OUT/summary.txt 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/sq4 $OUT/fetch $OUT/write $OUT/tcc $OUT/grbm $OUT/stats    # raw traces are large; the summaries are what is kept
cat $OUT/summary.txt; head -5 $OUT/kernel_stats.csv; tail -1 $OUT/stats.log | cut -c1-400
