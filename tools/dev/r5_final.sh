#!/bin/bash
# Round 5: the evidence run of the final tree -> gpurun_out/r5_final/ (copied into profiles/r5/ afterwards).
O=gpurun_out/r5_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python tools/dev/gen_time.py 256 10 > $O/gen_time.txt 2>&1
python tools/dev/gen_time.py 64 10 >> $O/gen_time.txt 2>&1
python tools/dev/gen_time.py 1024 10 >> $O/gen_time.txt 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_final/bench_default.json'))
print("main %.4g rollouts/s, %.4f ms/step, valu frac %.3f, verified %s" % (d['value'], d['ms_per_step'], d['roofline_valu']['frac'], d['verified']['ok']))
for k,v in d['configs'].items():
    print(k, "ms/step %.4f" % v['ms_per_step'], "verified", (v.get('verified') or {}).get('ok'), v.get('vs_one_launch_per_step'), v.get('stream_overlap'))
se=d['single_env']; print("single env %.1f us, verified %s clear %d" % (se['us_per_step'], se['verified']['ok'], se['verified']['clear']))
PY
