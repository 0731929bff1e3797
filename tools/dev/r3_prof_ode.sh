#!/bin/bash
# Round 3: kernel-trace stats + SQ counters of the predictor_ODE bench line (separate passes, as tools/profile.sh does for ODE_v0)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r3_ode; mkdir -p $OUT
B="$R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-env --no-extra-configs --predictor-type ODE"
B2="$R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single-env --no-extra-configs --predictor-type ODE"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -- python3 $B2 > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq2 -- python3 $B2 > $OUT/pmc_sq2.log 2>&1
echo done
