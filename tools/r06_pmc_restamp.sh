#!/bin/bash
# round 6: every PMC + --stats record once more.  The device code did not change (profiles/r06/code_objects.txt equals round
# 5's, byte for byte), but the records are stamped with a hash of the kernel SOURCES, which holds the host-side launch policy
# too, and that changed (quad_available, the unit policy planned for the build that is launched): bench.py refuses a record
# of other sources, so they are taken again rather than carried over by hand.
#   on the box: bash tools/r06_pmc_restamp.sh          here, afterwards: bash tools/r06_pmc_restamp.sh apply
if [ "$1" = "apply" ]; then
  for c in c3 c4 c2 c3b2 c5; do python3 tools/update_traffic.py gpurun_out/roofline_$c ${c}_n1 profiles/r06/final_$c > /dev/null || echo "FAILED $c"; done
  python3 tools/update_traffic.py gpurun_out/roofline_c3_inflight1 c3_n1_inflight1 profiles/r06/final_c3_inflight1 > /dev/null || echo "FAILED c3_inflight1"
  for c in c3 c5 c4 c2; do python3 tools/update_traffic.py gpurun_out/roofline_${c}_alone_wholespp ${c}_n1_inflight1_wholespp profiles/r06/final_${c}_timed_symbol_alone > /dev/null || echo "FAILED ${c}_alone"; done
  python3 - <<'PY'
import json
d = json.load(open("profiles/traffic.json"))["entries"]
for k, v in d.items():
    print(f"{k:28s} {v['kernel']:36s} {v['profiled_kernel_ms']:9.4f} ms  traffic {v['traffic_bytes'] / 1e6:8.1f} MB  VALU {v['valu_wave_instr'] / 1e9:6.3f} G  lanes {v['lane_utilisation']:.3f}  {v['source_hash']}  {v['profile']}")
PY
  exit 0
fi
mkdir -p gpurun_out/r06
bash tools/pmc_roofline.sh c3 > gpurun_out/r06/pmc_c3.log 2>&1; tail -2 gpurun_out/r06/pmc_c3.log | cut -c1-200
bash tools/pmc_roofline.sh c3_alone_wholespp --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 > gpurun_out/r06/pmc_c3_alone.log 2>&1
bash tools/pmc_roofline.sh c3_inflight1 --inflight 1 > gpurun_out/r06/pmc_c3_inflight1.log 2>&1
for c in c4 c2 c3b2 c5; do bash tools/pmc_roofline.sh $c --config $c > gpurun_out/r06/pmc_$c.log 2>&1; done
for c in c5 c4 c2; do bash tools/pmc_roofline.sh ${c}_alone_wholespp --config $c --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 > gpurun_out/r06/pmc_${c}_alone.log 2>&1; done
for t in c3 c3_alone_wholespp c3_inflight1 c4 c2 c3b2 c5 c5_alone_wholespp c4_alone_wholespp c2_alone_wholespp; do echo "== $t"; head -3 gpurun_out/roofline_$t/kernel_stats.csv | cut -c1-160; done
