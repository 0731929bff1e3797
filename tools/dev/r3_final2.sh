#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3final
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_c_consumer.py tests/test_gpu_bench_contract.py -m gpu -q > $O/pytest2.log 2>&1; echo "pytest rc $?" >> $O/pytest2.log
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default_b.json 2> $O/bench_default_b.err
tail -4 $O/pytest2.log
