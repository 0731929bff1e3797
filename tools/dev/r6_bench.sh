#!/bin/bash
# Round 6: the default bench line with its wall clock, then the round's rocprofv3 profile of the bench command (tools/profile.sh)
O=gpurun_out/r6i; mkdir -p $O
T0=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
T1=$(date +%s.%N)
echo "bench wall $(echo "$T1 - $T0" | bc) s" | tee $O/bench_wall.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6i/bench_default.json'))
print("main %.4g rollouts/s, %.4f ms/step, valu frac %.3f, verified %s second_stage %s" % (d['value'], d['ms_per_step'], d['roofline_valu']['frac'], d['verified']['ok'], d['verified'].get('second_stage_envs')))
for k,v in d['configs'].items():
    ver = v.get('verified') or {}
    print(k, "ms/step %.4f" % v['ms_per_step'], "verified", ver.get('ok'), "2nd", ver.get('second_stage_envs'), v.get('vs_one_launch_per_step'), v.get('stream_overlap'))
print(d.get('wall_s')); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
se=d['single_env']; print("single env %.1f us, verified %s" % (se['us_per_step'], se['verified']['ok']))
PY
T0=$(date +%s.%N)
timeout 900 python bench.py --cpu-baseline full > $O/bench_full_baseline.json 2>/dev/null
T1=$(date +%s.%N)
echo "bench --cpu-baseline full wall $(echo "$T1 - $T0" | bc) s" | tee -a $O/bench_wall.txt
bash tools/profile.sh r6 > $O/profile.log 2>&1; tail -1 $O/profile.log
