#!/bin/bash
# round-3 GPU batch 1: tests, default bench, the 1-rank RCCL collective (native vs torch.distributed), seam latency
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_native.json 2> $O/bench_rccl_native.err
CPMPPI_BENCH_COLLECTIVE=torch timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_torch.json 2> $O/bench_rccl_torch.err
CPMPPI_COMM_READY_FENCE=1 timeout 400 python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_native_fence.json 2> $O/bench_rccl_native_fence.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4_native -- python3 bench.py --gpus 1 --config C4 --steps 50 --warmup 10 --no-cpu-baseline --no-single-env --no-extra-configs > $O/trace_c4_native.json 2> $O/trace_c4_native.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
timeout 300 python tools/dev/seam_latency.py > $O/seam.txt 2> $O/seam.err
CPMPPI_HOST_ZERO_COPY_MAX=0 timeout 300 python tools/dev/seam_latency.py > $O/seam_copy.txt 2> $O/seam_copy.err
ls -la $O
tail -5 $O/pytest.log
