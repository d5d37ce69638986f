#!/bin/bash
# Runs a command with a prebuilt library variant (gpurun_variants/var_<name>.so) in the product library's place, then restores it.
#   usage: tools/with_variant.sh <name> <command ...>
NAME=$1; shift
PKG=cpuvoxelraycaster_amd
cp $PKG/libvrc_hip.so /tmp/libvrc_hip.product.so
cp gpurun_variants/var_$NAME.so $PKG/libvrc_hip.so
python3 - <<'PY'
from cpuvoxelraycaster_amd import build
open(build.STAMP, "w").write(build._flag_line() + "\n")
PY
touch $PKG/libvrc_hip.so
"$@"
RC=$?
cp /tmp/libvrc_hip.product.so $PKG/libvrc_hip.so; touch $PKG/libvrc_hip.so
exit $RC
