#!/bin/bash
# A/B of kernel build variants on the GPU box: for every "name=flags" argument rebuild libvrc_hip.so with
# VRC_EXTRA_FLAGS=flags (cpuvoxelraycaster_amd/build.py notices the changed flag line) and run bench.py.
#   usage: tools/ab_variants.sh <outdir> "<bench args>" name1="-DFOO" name2="" ...
OUT=$1; shift
BARGS=$1; shift
mkdir -p $OUT
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  VRC_EXTRA_FLAGS="$flags" python3 bench.py --no-cpu-baseline --no-extras $BARGS > $OUT/$name.log 2>&1
  python3 - "$OUT/$name.log" "$name" "$flags" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:8.4f}  kernel_ms {d['roofline']['kernel_ms']:8.4f}  Mrays/s {d['value']:10.1f}   [{sys.argv[3]}]")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
