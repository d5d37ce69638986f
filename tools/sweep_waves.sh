#!/bin/bash
# Which build of the stage-synchronous frame kernel (6 / 7 / 8 waves per SIMD) wins for which launch: the table behind
# sync_waves_for (csrc/vrc_kernels.hip), measured off the tuning pose as well (round-2 verdict item 8).
#   usage: tools/sweep_waves.sh <outfile> [configs, default "c2 c3 c4 c5"]
# For every config x camera pitch {0, -0.5, -1.2} x {three frames in flight (whole-spp units, fused resolve), one frame at a
# time (automatic chunk)}: ms per frame of the library's own choice and of each forced build.
OUT=$1; shift
CONFIGS=${1:-"c2 c3 c4 c5"}
: > $OUT
for cfg in $CONFIGS; do
  steps=60; [ $cfg = c5 ] && steps=12; [ $cfg = c2 ] && steps=200
  for pitch in 0 -0.5 -1.2; do
    for fl in 3 1; do
      line="$cfg pitch $pitch inflight $fl:"
      for bpc in auto 6 7 8; do
        targ=""; [ $bpc != auto ] && targ="--tuning $bpc,0,0"
        python3 bench.py --config $cfg --pitch=$pitch --inflight $fl --steps $steps --blocks 3 --blocks-min-seconds 0 --no-cpu-baseline --no-extras $targ > /tmp/sweep.json 2>/tmp/sweep.err
        r=$(python3 - <<'PY'
import json
try:
    d = json.loads(open("/tmp/sweep.json").read().strip().splitlines()[-1])
    k = d["roofline"]["kernel"].replace("k_render_sync", "")
    print(f"{d['ms_per_step']:.4f} ({k or 'plain'})")
except Exception as e:
    print("FAILED")
PY
)
        line="$line  $bpc: $r"
      done
      echo "$line" | tee -a $OUT
    done
  done
done
