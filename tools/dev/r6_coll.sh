#!/bin/bash
# Round 6: the new collective tests + the bench line with ONE rank on real RCCL, now with the grouped configurations under the
# collective (configs.C4_pipelined.collective, collective_cost = with / without the per-step all-gather, same process).
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_two_rank_gather.py tests/test_gpu_pipeline.py tests/test_gpu_schedule.py tests/test_gpu_boundary.py -m gpu -x -q > $O/tests.txt 2>&1; echo "rc $?" >> $O/tests.txt
tail -25 $O/tests.txt
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 CPMPPI_BENCH_FORCE_COLLECTIVE=1
MASTER_PORT=29631 timeout 900 python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
echo "bench rc $?"
unset RANK WORLD_SIZE LOCAL_RANK CPMPPI_BENCH_FORCE_COLLECTIVE
python - <<'PY'
import json
L=[l for l in open('gpurun_out/r6c/bench_rccl_1rank.json') if l.startswith('{"metric"')]
d=json.loads(L[-1]); print(json.dumps(d['config'].get('collective'), indent=1))
for k,v in d.get('configs',{}).items():
    print(k, {x: v.get(x) for x in ('ms_per_step','one_group_ms_per_step','without_collective_ms_per_step','collective_cost','stream_overlap','error')}, json.dumps(v.get('collective'))[:400])
print({k:(v.get('ok'), v.get('error')) for k,v in d.get('verified_configs', d.get('verified', {})).items()} if isinstance(d.get('verified'), dict) else None)
PY
tail -5 $O/bench_rccl_1rank.err
