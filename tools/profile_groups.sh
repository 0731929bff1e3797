#!/bin/bash
# Runs on the GPU box (via gpurun): the small configurations (BASELINE configs[2] / configs[3]) as ONE launch per step and as TWO env
# groups - rocprofv3 kernel trace + stats (kernel durations, which kernels are in flight together) and a separate SQ / GRBM PMC pass.
# The profiled program is tools/dev/groups_gather_cost.py --only none (cpmppi_groups_run, exactly what bench.py's *_pipelined side
# configurations and their one-group yardstick call), started directly behind `--`.
# Usage: tools/profile_groups.sh <tag>    -> gpurun_out/prof_<tag>/groups_<time>/...   (summarised by tools/summarize_groups.py <tag>)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r6}
OUT=$R/gpurun_out/prof_$TAG/groups_$(date +%Y%m%d_%H%M%S)
mkdir -p $OUT
for CFG in C4 C3; do
  for G in 1 2; do
    P="tools/dev/groups_gather_cost.py --only none --config $CFG --groups $G --reps 1"
    # un-profiled wall time + the bench's own overlap figure
    timeout 300 python3 $P --steps 300 --overlap > $OUT/plain_${CFG}_g$G.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${CFG}_g$G -- python3 $P --steps 300 > $OUT/trace_${CFG}_g$G.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${CFG}_g$G -- python3 $P --steps 40 > $OUT/pmc_${CFG}_g$G.log 2>&1
  done
done
find $OUT -name "*.csv" | wc -l
