#!/bin/bash
# round 5: the hand-over kernel -- parity first, then A/B against the plain kernel on the default bench command
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_handover.py -q -x > gpurun_out/r05/handover_tests.log 2>&1; echo "handover tests rc=$?"; tail -5 gpurun_out/r05/handover_tests.log
for h in 0 16 8 12 0 16; do
  timeout 300 python bench.py --no-cpu-baseline --no-oracle-frame --no-extras --blocks-min-seconds 2 --handover $h > gpurun_out/r05/ab_handover_$h.json.log 2> gpurun_out/r05/ab_handover_$h.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r05/ab_handover_$h.json.log").read().strip().splitlines()[-1])
    print("handover $h:", d["ms_per_step"], "ms", d["roofline"]["kernel"], d["extra"]["ms_per_step_blocks"]["min_median_max"])
except Exception as e:
    print("handover $h: failed", e)
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --blocks-min-seconds 1 --handover 16 > gpurun_out/r05/handover_16_oracle.json.log 2>/dev/null; echo "oracle-checked rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-oracle-frame --no-extras --blocks-min-seconds 2 --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 --handover 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alone ho16', d['ms_per_step'], d['roofline']['kernel'])"
timeout 300 python bench.py --no-cpu-baseline --no-oracle-frame --no-extras --blocks-min-seconds 2 --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 --handover 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alone ho0', d['ms_per_step'], d['roofline']['kernel'])"
for i in 1 2; do timeout 400 python -m pytest tests/test_gpu_peer_write.py -q -x > gpurun_out/r05/peer_tests_$i.log 2>&1; tail -2 gpurun_out/r05/peer_tests_$i.log; done
