#!/bin/bash
# Runs a command with a prebuilt library variant (gpurun_variants/var_<name>.so) loaded in the product library's place
# (VRC_LIB, cpuvoxelraycaster_amd/capi.py); the product library is never touched.
#   usage: tools/with_variant.sh <name> <command ...>
NAME=$1; shift
VRC_LIB=$(pwd)/gpurun_variants/var_$NAME.so "$@"
