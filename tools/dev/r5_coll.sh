#!/bin/bash
# Round 5: the bench line with ONE rank on real RCCL (cpmppi_step_gather): what the line now says about the communicator
# (config.collective: rccl_ranks from ncclCommCount, the gathered blocks against every rank's own checksum).
O=gpurun_out/r5h; mkdir -p $O
python -m pytest tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -3 > $O/boundary_tests.txt
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 CPMPPI_BENCH_FORCE_COLLECTIVE=1
MASTER_PORT=29631 python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
echo "rc $?" >> $O/boundary_tests.txt
unset RANK WORLD_SIZE LOCAL_RANK CPMPPI_BENCH_FORCE_COLLECTIVE
python - <<'PY'
import json
L=[l for l in open('gpurun_out/r5h/bench_rccl_1rank.json') if l.startswith('{"metric"')]
d=json.loads(L[-1]); print(json.dumps(d['config']['collective'], indent=1)); print({k:(v['ms_per_step'], v.get('collective')) for k,v in d.get('configs',{}).items()})
PY
cat $O/boundary_tests.txt; tail -3 $O/bench_rccl_1rank.err
