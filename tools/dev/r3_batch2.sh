#!/bin/bash
# round-3 GPU batch 2: step_gather (device-flag ordering) vs events, C3 parity diagnostic, seam breakdown
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_harness.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_flags.json 2> $O/bench_rccl_flags.err
CPMPPI_BENCH_COLLECTIVE=native-events timeout 400 python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_events.json 2> $O/bench_rccl_events.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4_flags -- python3 bench.py --gpus 1 --config C4 --steps 50 --warmup 10 --no-cpu-baseline --no-single-env --no-extra-configs > $O/trace_c4_flags.json 2> $O/trace_c4_flags.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
timeout 400 python bench.py --no-cpu-baseline --no-single-env > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python tools/dev/seam_latency.py > $O/seam.txt 2> $O/seam.err
timeout 600 python tools/dev/cfg_parity_diag.py C3 > $O/diag_c3.jsonl 2> $O/diag_c3.err
timeout 600 python tools/dev/cfg_parity_diag.py C4 > $O/diag_c4.jsonl 2> $O/diag_c4.err
tail -5 $O/pytest.log
