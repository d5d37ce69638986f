#!/bin/bash
# A/B of PREBUILT library variants (tools/build_variant.py -> gpurun_variants/var_<name>.so) on the GPU box: each is copied
# over the package's libvrc_hip.so (stamped with the default flag line so that build.py does not rebuild it) and bench.py runs.
#   usage: tools/ab_prebuilt.sh <outdir> "<bench args>" name1 name2 ...     (the product library is restored at the end)
OUT=$1; shift
BARGS=$1; shift
PKG=cpuvoxelraycaster_amd
mkdir -p $OUT
cp $PKG/libvrc_hip.so /tmp/libvrc_hip.product.so
for name in "$@"; do
  cp gpurun_variants/var_$name.so $PKG/libvrc_hip.so
  python3 - <<'PY'
from cpuvoxelraycaster_amd import build
open(build.STAMP, "w").write(build._flag_line() + "\n")
PY
  touch $PKG/libvrc_hip.so
  python3 bench.py --no-cpu-baseline --no-extras $BARGS > $OUT/$name.log 2>&1
  python3 - "$OUT/$name.log" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:24s} ms/step {d['ms_per_step']:8.4f}  kernel_ms {d['roofline']['kernel_ms']:8.4f}  Mrays/s {d['value']:10.1f}  {d['roofline'].get('kernel')}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
cp /tmp/libvrc_hip.product.so $PKG/libvrc_hip.so; touch $PKG/libvrc_hip.so
