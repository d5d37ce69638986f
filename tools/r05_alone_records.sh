#!/bin/bash
# round 5: the timed kernel symbol of C5, C4 and C2 one launch at a time (rocprofv3 --stats + the PMC passes), as
# tools/r05_pmc_restamp.sh took it for C3 -- the records behind roofline.kernel_ms_alone_rocprof of those configurations' lines
mkdir -p gpurun_out/r05
for c in c5 c4 c2; do
  bash tools/pmc_roofline.sh ${c}_alone_wholespp --config $c --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 > gpurun_out/r05/pmc_${c}_alone.log 2>&1
  echo "== $c"; head -3 gpurun_out/roofline_${c}_alone_wholespp/kernel_stats.csv | cut -c1-160
done
