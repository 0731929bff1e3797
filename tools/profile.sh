#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r1}
OUT=$R/gpurun_out/prof_$TAG/run_$(date +%Y%m%d_%H%M%S)     # one sub-directory per run: gpurun merges, it never deletes
mkdir -p $OUT
for NOISE in philox buffer buffer-ref; do
  B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-env --no-extra-configs --noise $NOISE"   # (25 launches: the first ones run cold and would dominate a 12-launch average)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$NOISE -- python3 $B > $OUT/stats_$NOISE.log 2>&1
  B2="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-env --no-extra-configs --no-verify --noise $NOISE"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$NOISE -- python3 $B2 > $OUT/pmc_fetch_$NOISE.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$NOISE -- python3 $B2 > $OUT/pmc_write_$NOISE.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq_$NOISE -- python3 $B2 > $OUT/pmc_sq_$NOISE.log 2>&1
  timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq2_$NOISE -- python3 $B2 > $OUT/pmc_sq2_$NOISE.log 2>&1
done
# GRU predictor (MFMA) kernel: kernel-trace stats only
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_gru -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-env --no-extra-configs --predictor gru --envs 256 > $OUT/stats_gru.log 2>&1
find $OUT -name "*.csv" | wc -l
