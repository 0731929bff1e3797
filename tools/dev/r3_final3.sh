#!/bin/bash
# refresh of the round-3 evidence touched by the last kernel change (mid-size build: one edge test per triple)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3final3
mkdir -p $O
cd $R
V=build_variants
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for A in "--config C3" "--config C4"; do
  T=$(echo $A | tr -d ' -'); timeout 400 python bench.py --no-cpu-baseline --no-single-env --no-extra-configs $A > $O/bench_$T.json 2> $O/bench_$T.err
done
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4_collective -- python3 bench.py --gpus 1 --config C4 --steps 60 --warmup 10 --no-cpu-baseline --no-single-env --no-extra-configs > $O/trace_c4_collective.json 2> $O/trace_c4_collective.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default_b.json 2> $O/bench_default_b.err
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 64 --rollouts 2048 --horizon 50 --rounds 8 --steps 20 --noise philox buffer > $O/kbench_c4.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 64 --rollouts 4096 --horizon 100 --rounds 8 --steps 10 --noise philox > $O/kbench_c3.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 256 --rounds 8 --steps 10 --noise philox > $O/kbench_256.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 1024 --rounds 6 --steps 6 --noise philox > $O/kbench_1024.txt 2>&1
bash tools/profile_cfg.sh r3 "0" > $O/profile_cfg.log 2>&1
tail -3 $O/pytest.log
