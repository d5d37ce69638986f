#!/bin/bash
# experiment (round 4): unit policies for a launch alone on the chip, 8 x 8 map: head chunk, tail chunk, tail units per wave
export VRC_LIB=$(pwd)/gpurun_variants/var_expunits.so
for pol in "" "4,1,2" "4,1,4" "4,1,8" "4,1,12" "4,1,16" "4,2,4" "4,2,8" "2,1,4" "2,1,8"; do
  export VRC_EXP_UNITS=$pol
  python3 bench.py --inflight 1 --lane-samples 1 --no-cpu-baseline --no-extras --steps 100 --blocks-min-seconds 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('policy [%s]  ms/step %.4f  kernel %s' % ('$pol', d['ms_per_step'], d['roofline']['kernel']))"
done
unset VRC_EXP_UNITS
python3 bench.py --inflight 1 --lane-samples 4 --no-cpu-baseline --no-extras --steps 100 --blocks-min-seconds 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('samples abreast      ms/step %.4f  kernel %s' % (d['ms_per_step'], d['roofline']['kernel']))"
