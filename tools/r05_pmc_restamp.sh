#!/bin/bash
# round 5: the PMC + --stats records of every configuration on the final kernel sources (quadrant walks)
mkdir -p gpurun_out/r05
bash tools/pmc_roofline.sh c3 > gpurun_out/r05/pmc_c3.log 2>&1; tail -2 gpurun_out/r05/pmc_c3.log | cut -c1-200
bash tools/pmc_roofline.sh c3_alone_wholespp --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 > gpurun_out/r05/pmc_c3_alone.log 2>&1
bash tools/pmc_roofline.sh c3_inflight1 --inflight 1 > gpurun_out/r05/pmc_c3_inflight1.log 2>&1
bash tools/pmc_roofline.sh c4 --config c4 > gpurun_out/r05/pmc_c4.log 2>&1
bash tools/pmc_roofline.sh c2 --config c2 > gpurun_out/r05/pmc_c2.log 2>&1
bash tools/pmc_roofline.sh c3b2 --config c3b2 > gpurun_out/r05/pmc_c3b2.log 2>&1
bash tools/pmc_roofline.sh c5 --config c5 > gpurun_out/r05/pmc_c5.log 2>&1
for t in c3 c3_alone_wholespp c3_inflight1 c4 c2 c3b2 c5; do echo "== $t"; head -3 gpurun_out/roofline_$t/kernel_stats.csv | cut -c1-160; done
