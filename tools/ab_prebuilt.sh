#!/bin/bash
# A/B of PREBUILT library variants (tools/build_variant.py -> gpurun_variants/var_<name>.so) on the GPU box: bench.py runs with
# VRC_LIB pointing at each (cpuvoxelraycaster_amd/capi.py loads it in the product's place and says so; the product library is
# never touched; the bench line records the library in config.library).  The name `product` runs the product library.
#   usage: tools/ab_prebuilt.sh <outdir> "<bench args>" name1 name2 ...
OUT=$1; shift
BARGS=$1; shift
mkdir -p $OUT
for name in "$@"; do
  if [ "$name" = product ]; then unset VRC_LIB; else export VRC_LIB=$(pwd)/gpurun_variants/var_$name.so; fi
  python3 bench.py --no-cpu-baseline --no-extras --blocks-min-seconds 0.5 $BARGS > $OUT/$name.log 2>$OUT/$name.err
  python3 - "$OUT/$name.log" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:8.4f}  kernel_ms {d['roofline']['kernel_ms']:8.4f}  Mrays/s {d['value']:10.1f}  {d['roofline'].get('kernel')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
unset VRC_LIB
