#!/bin/bash
# Round 6, closing runs on the final tree (kernel device code unchanged since round 5: profiles/r06/code_objects.txt): the whole
# GPU suite and smoke(); the bench lines of every configuration with the re-stamped PMC records (profiles/traffic.json); the
# timed symbol one launch at a time; the verdict's done criterion for the exchange forms -- `VRC_DIST_BACKEND=gloo python3
# bench.py --gpus 4 --steps 5` (and worlds 2 and 8) with all three forms and their parity bits in one line; RCCL with a world of
# one rank; ten minutes of differential fuzz (the round-5 verdict's cap for a round without a kernel change).
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 > $O/gputests_closing.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> $O/gputests_closing.log
cat $O/gputests_closing.log
python bench.py > $O/final_bench.json.log 2> $O/final_bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench_steps20.json.log 2>/dev/null; echo "steps20 rc=$?"
for c in c2 c4 c5; do python bench.py --config $c --blocks-min-seconds 2 > $O/final_bench_$c.json.log 2>/dev/null; echo "$c rc=$?"; done
python bench.py --inflight 1 --blocks-min-seconds 2 --no-cpu-baseline > $O/final_bench_inflight1.json.log 2>/dev/null; echo "inflight1 rc=$?"
python bench.py --inflight 1 --whole-spp 1 --fused 1 --lane-samples 1 --blocks-min-seconds 2 --no-cpu-baseline > $O/final_bench_timed_symbol_alone.json.log 2>/dev/null; echo "alone rc=$?"
VRC_FORCE_DIST=1 python bench.py --steps 20 --blocks-min-seconds 1 --no-cpu-baseline > $O/force_dist_rccl_world1.json.log 2>/dev/null; echo "rccl world1 rc=$?"
for n in 2 4 8; do
  VRC_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus $n --steps 5 > $O/exchange_forms_gloo$n.json.log 2> $O/exchange_forms_gloo$n.err; echo "gloo $n rc=$?"
done
python3 tools/scale_report.py $O/final_bench.json.log $O/exchange_forms_gloo2.json.log $O/exchange_forms_gloo4.json.log $O/exchange_forms_gloo8.json.log > $O/scale_report_rehearsal.txt 2>&1; cut -c1-200 $O/scale_report_rehearsal.txt | head -30
timeout 700 python tests/tools/fuzz_gpu.py --seconds ${FUZZ_SECONDS:-600} --seed 6006 --quad-bias 0.4 > $O/fuzz_final_10min.log 2>&1; tail -2 $O/fuzz_final_10min.log
