#!/bin/bash
# round 6: frames in flight per GPU (renderers / streams taking turns) 3 (the default) against 2, 4, 5, 6 on one box, interleaved,
# three repetitions: is the default still the right one with 8 hardware queues?  (profiles/r05/shard_inflight_final.txt had 4 ahead
# of 3 by 0.8 % on the whole C3 frame and behind it on the N = 2 shard.)
OUT=${1:-gpurun_out/r06/inflight}; mkdir -p $OUT
for rep in 1 2 3; do
  for c in c3 c4 c5; do
    for fl in 3 4 2 5 6; do
      python3 bench.py --config $c --inflight $fl --whole-spp 1 --fused 1 --no-cpu-baseline --no-extras --no-oracle-frame --steps 100 --blocks 5 --blocks-min-seconds 3 > $OUT/${c}_f${fl}_$rep.log 2>&1
    done
  done
done
python3 - $OUT <<'PY' | tee $OUT/table.txt
import glob, json, os, sys
print("frame period (ms, median block of each run) against frames in flight; whole-spp units, fused resolve; three interleaved repetitions")
for c in ("c3", "c4", "c5"):
    for fl in (2, 3, 4, 5, 6):
        v = []
        for f in sorted(glob.glob(os.path.join(sys.argv[1], f"{c}_f{fl}_*.log"))):
            try:
                v.append(json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"])
            except Exception:
                v.append(float("nan"))
        print(f"  {c}  frames in flight {fl}   " + "  ".join(f"{x:8.4f}" for x in v) + f"   median {sorted(v)[len(v)//2]:8.4f}")
PY
