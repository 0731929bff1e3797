#!/bin/bash
# the C4 step with and without the per-step all-gather (one rank on RCCL), for two builds of the library (development batch)
O=gpurun_out/prio; mkdir -p $O
# builds: python __graft_entry__.py --variant prio0 -DCPMPPI_WAVE_PRIORITY=0 ; python __graft_entry__.py --variant prio1
for rep in 1 2; do for v in prio0 prio1; do
  export CPMPPI_LIB=build_variants/$v.so
  timeout 300 python bench.py --config C4 --steps 400 --warmup 50 --no-cpu-baseline --no-single-env --no-extra-configs > $O/plain_${v}_$rep.json 2>/dev/null
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$rep CPMPPI_BENCH_FORCE_COLLECTIVE=1 timeout 300 python bench.py --gpus 1 --config C4 --steps 400 --warmup 50 --no-cpu-baseline --no-single-env --no-extra-configs > $O/coll_${v}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/prio/*.json')):
    L=[l for l in open(f) if l.startswith('{"metric"')]
    d=json.loads(L[-1]); print(f, round(d['ms_per_step']*1e3,2), 'us/step  kernel', d.get('roofline',{}).get('kernel_ms'), d.get('kernel_ms'))
PY
