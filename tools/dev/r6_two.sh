#!/bin/bash
O=gpurun_out/r6k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_pipeline.py tests/test_gpu_two_rank_gather.py -m gpu -x -q -s > $O/tests.txt 2>&1; echo "tests rc $?"; grep -E "passed|failed|without the guard" $O/tests.txt | tail -4
CPMPPI_BENCH_BACKEND=gloo CPMPPI_BENCH_ONE_DEVICE=1 CPMPPI_BENCH_COLLECTIVE=native CPMPPI_BENCH_RCCL_PATH=$PWD/tests/fake_rccl/libfake_rccl.so \
  timeout 600 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_2ranks_one_device_fake_rccl.json 2> $O/bench_2ranks.err; echo "2-rank rc $?"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 CPMPPI_BENCH_FORCE_COLLECTIVE=1 \
  python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err; echo "1-rank rc $?"
python - <<'PY'
import json
for f in ('bench_rccl_1rank.json', 'bench_2ranks_one_device_fake_rccl.json'):
    try:
        L=[l for l in open('gpurun_out/r6k/'+f) if l.startswith('{"metric"')]
        d=json.loads(L[-1])
        print(f, "n_gpus", d['n_gpus'], "value %.4g" % d['value'], "ms/step", d['ms_per_step'], json.dumps(d['config'].get('collective'))[:700], d['verified']['ok'])
        for k,v in d.get('configs',{}).items():
            print("  ", k, {x: v.get(x) for x in ('ms_per_step','without_collective_ms_per_step','collective_cost','error')}, json.dumps(v.get('collective'))[:200], (v.get('verified') or {}).get('ok'))
    except Exception as e:
        print(f, "FAILED", e)
PY
grep -v "WARN\|^$\|iommu" $O/bench_2ranks.err | grep -i "error\|Traceback\|rank0\]" | head -20
