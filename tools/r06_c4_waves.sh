#!/bin/bash
# round 6 (verdict r5, Next 4): the lens (DOF) kernel at 6 waves per SIMD (k_render_sync_resolved: 80 VGPRs, 10 spills, 44 B
# scratch since the round-5 kernel-argument-block load) against the shipped 7-wave build (k_render_sync_resolved_w7: 72 VGPRs,
# 17 spills, 72 B) on BASELINE config 4, frames in flight as bench.py times them and one launch at a time.  No rebuild:
# `--tuning N` is vrc_set_tuning(blocks_per_cu), which sync_waves_for rounds to a build that exists.  A B A B A B.
OUT=${1:-gpurun_out/r06/c4_waves}; mkdir -p $OUT
for rep in 1 2 3; do
  for w in 6 7; do
    for fl in 3 1; do
      python3 bench.py --config c4 --tuning $w --inflight $fl --no-cpu-baseline --no-extras --no-oracle-frame --steps 100 --blocks 5 --blocks-min-seconds 4 > $OUT/w${w}_f${fl}_$rep.log 2>&1
    done
  done
done
python3 - $OUT <<'PY' | tee $OUT/table.txt
import glob, json, os, sys
print("lens kernel (C4: 512^3, 1920x1080, 4 spp GI + DOF), waves per SIMD x frames in flight; ms per frame (median block of each run)")
for fl in (3, 1):
    for w in (6, 7):
        v, k = [], None
        for f in sorted(glob.glob(os.path.join(sys.argv[1], f"w{w}_f{fl}_*.log"))):
            try:
                d = json.loads(open(f).read().strip().splitlines()[-1]); v.append(d["ms_per_step"]); k = d["roofline"]["kernel"]
            except Exception as e:
                v.append(float("nan"))
        print(f"  frames in flight {fl}  waves {w}  {k:34s}  " + "  ".join(f"{x:.4f}" for x in v) + f"   median {sorted(v)[len(v)//2]:.4f}")
PY
