#!/bin/bash
# Round 6: which build of the rollout kernel should CO-RUNNING env groups get?  (tools/variant_sweep.py chose per launch size for
# launches running ALONE; two groups put two waves on every SIMD.)  devknobs build: the size rule's limits from the environment.
O=gpurun_out/r6n; mkdir -p $O
export CPMPPI_LIB=$PWD/build_variants/devknobs.so
for CFG in C4 C3; do
  ST=300; [ $CFG = C3 ] && ST=150
  echo "== $CFG, 2 groups, library default" | tee -a $O/variants_corun.txt
  timeout 200 python tools/dev/groups_gather_cost.py --only none --config $CFG --groups 2 --reps 3 --steps $ST --overlap 2>/dev/null | grep -E "us/step|stream_overlap" | tee -a $O/variants_corun.txt
  echo "== $CFG, 2 groups, R1 throughput build (CPMPPI_LATENCY_MAX_ROLLOUTS=0)" | tee -a $O/variants_corun.txt
  CPMPPI_LATENCY_MAX_ROLLOUTS=0 timeout 200 python tools/dev/groups_gather_cost.py --only none --config $CFG --groups 2 --reps 3 --steps $ST --overlap 2>/dev/null | grep -E "us/step|stream_overlap" | tee -a $O/variants_corun.txt
  echo "== $CFG, 2 groups, R2 lone form" | tee -a $O/variants_corun.txt
  CPMPPI_LONE_FORM_MAX_WAVES=1099511627776 timeout 200 python tools/dev/groups_gather_cost.py --only none --config $CFG --groups 2 --reps 3 --steps $ST --rpl 2 --overlap 2>/dev/null | grep -E "us/step|stream_overlap" | tee -a $O/variants_corun.txt
  echo "== $CFG, 2 groups, R2 phased / throughput" | tee -a $O/variants_corun.txt
  CPMPPI_LONE_FORM_MAX_WAVES=0 timeout 200 python tools/dev/groups_gather_cost.py --only none --config $CFG --groups 2 --reps 3 --steps $ST --rpl 2 --overlap 2>/dev/null | grep -E "us/step|stream_overlap" | tee -a $O/variants_corun.txt
  echo "== $CFG, 4 groups, R1 throughput build" | tee -a $O/variants_corun.txt
  CPMPPI_LATENCY_MAX_ROLLOUTS=0 timeout 200 python tools/dev/groups_gather_cost.py --only none --config $CFG --groups 4 --reps 3 --steps $ST --overlap 2>/dev/null | grep -E "us/step|stream_overlap" | tee -a $O/variants_corun.txt
  echo "== $CFG, 3 groups, library default" | tee -a $O/variants_corun.txt
  timeout 200 python tools/dev/groups_gather_cost.py --only none --config $CFG --groups 3 --reps 3 --steps $ST --overlap 2>/dev/null | grep -E "us/step|stream_overlap" | tee -a $O/variants_corun.txt
done
