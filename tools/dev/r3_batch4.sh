#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 600 python tools/dev/cfg_parity_diag.py C3 > $O/diag_c3.jsonl 2> $O/diag_c3.err
timeout 600 python tools/dev/cfg_parity_diag.py C4 > $O/diag_c4.jsonl 2> $O/diag_c4.err
DIAG_SENS=1e-4 timeout 600 python tools/dev/cfg_parity_diag.py C3 > $O/diag_c3_s1.jsonl 2> $O/diag_c3_s1.err
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --no-cpu-baseline --no-single-env --no-extra-configs --noise buffer > $O/bench_buffer.json 2> $O/bench_buffer.err
timeout 400 python bench.py --no-cpu-baseline --no-single-env --no-extra-configs --noise buffer-ref > $O/bench_bufferref.json 2> $O/bench_bufferref.err
timeout 300 python tools/dev/seam_latency.py > $O/seam.txt 2> $O/seam.err
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_flags.json 2> $O/bench_rccl_flags.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_configs.py > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
