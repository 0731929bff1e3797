#!/bin/bash
# Round 6 evidence run: GPU suite, default bench line (wall clock), C3 spread with clear fractions, groups profile
O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
/usr/bin/time -v -o $O/bench_time.txt timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
grep -E "Elapsed|Maximum resident" $O/bench_time.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6h/bench_default.json'))
print("main %.4g rollouts/s, %.4f ms/step, valu frac %.3f, verified %s second_stage %s" % (d['value'], d['ms_per_step'], d['roofline_valu']['frac'], d['verified']['ok'], d['verified'].get('second_stage_envs')))
for k,v in d['configs'].items():
    ver = v.get('verified') or {}
    print(k, "ms/step %.4f" % v['ms_per_step'], "verified", ver.get('ok'), "2nd", ver.get('second_stage_envs'), v.get('vs_one_launch_per_step'), v.get('stream_overlap'))
print(d.get('wall_s')); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
PY
timeout 900 python tools/c3_spread.py --json $O/c3_spread.json > $O/c3_spread.txt 2>/dev/null; cat $O/c3_spread.txt
bash tools/profile_groups.sh r6 > $O/profile_groups.log 2>&1; tail -2 $O/profile_groups.log
