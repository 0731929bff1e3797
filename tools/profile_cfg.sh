#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + a separate SQ/GRBM PMC pass of bench.py at BASELINE's own
# single-GPU configs (C3, C4) for each lane mapping.  Usage: tools/profile_cfg.sh <tag> [rpl list]
#   -> gpurun_out/prof_<tag>/cfg_<time>/{stats,pmc}_<config>_rpl<r>/...   (summarised by tools/summarize_cfg.py)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r2}
RPLS=${2:-"0 1 2"}
OUT=$R/gpurun_out/prof_$TAG/cfg_$(date +%Y%m%d_%H%M%S)
mkdir -p $OUT
for CFG in C3 C4; do
  for RPL in $RPLS; do
    B="bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-single-env --config $CFG --rpl $RPL"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${CFG}_rpl$RPL -- python3 $B > $OUT/stats_${CFG}_rpl$RPL.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${CFG}_rpl$RPL -- python3 $B > $OUT/pmc_${CFG}_rpl$RPL.log 2>&1
  done
done
find $OUT -name "*.csv" | wc -l
