#!/bin/bash
# Round 4: the per-step all-gather with ONE rank on real RCCL (cpmppi_step_gather) against the plain run on the same box:
# default line (8192 envs, C3, C4 as side configurations) and C4 on its own; both forms of the side-stream waiter.
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -3 > $O/coll_tests.txt
COMMON="--no-cpu-baseline --no-single-env"
python bench.py $COMMON > $O/coll_plain.json 2>/dev/null
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 CPMPPI_BENCH_FORCE_COLLECTIVE=1
MASTER_PORT=29531 python bench.py --gpus 1 $COMMON > $O/coll_rccl_streamops.json 2>/dev/null
MASTER_PORT=29532 CPMPPI_COMM_WAITER=kernel python bench.py --gpus 1 $COMMON > $O/coll_rccl_kernelwaiter.json 2>/dev/null
unset RANK WORLD_SIZE LOCAL_RANK CPMPPI_BENCH_FORCE_COLLECTIVE
for rep in 1 2; do
  python bench.py --config C4 --steps 400 --warmup 50 $COMMON --no-extra-configs --no-verify > $O/coll_c4_plain_$rep.json 2>/dev/null
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_PORT=2954$rep CPMPPI_BENCH_FORCE_COLLECTIVE=1 python bench.py --gpus 1 --config C4 --steps 400 --warmup 50 $COMMON --no-extra-configs --no-verify > $O/coll_c4_rccl_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4/coll_*.json')):
    L=[l for l in open(f) if l.startswith('{"metric"')]
    if not L: print(f, "NO LINE"); continue
    d=json.loads(L[-1]); c=d.get("configs",{})
    print(f.split('/')[-1], "main %.4f ms" % d['ms_per_step'], " ".join("%s %.4f" % (k, v['ms_per_step']) for k,v in c.items()), d['config'].get('collective','')[:60])
PY
cat $O/coll_tests.txt
