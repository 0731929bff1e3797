#!/bin/bash
# the N > 1 code path of bench.py on a 1-GPU box: two ranks over gloo sharing the device (plumbing check, not a scaling point)
O=gpurun_out/r4; mkdir -p $O
CPMPPI_BENCH_BACKEND=gloo CPMPPI_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err
echo rc=$?; tail -3 $O/bench_gloo2.err
python - <<'PY'
import json
L=[l for l in open("gpurun_out/r4/bench_gloo2.json") if l.startswith('{"metric"')]
d=json.loads(L[-1]); print(d["n_gpus"], "%.4g"%d["value"], d["config"].get("collective","")[:70], d["verified"]["ok"], {k:(v["ms_per_step"], v.get("verified",{}).get("ok")) for k,v in d.get("configs",{}).items()})
PY
