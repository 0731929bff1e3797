#!/bin/bash
O=gpurun_out/r6j; mkdir -p $O
for i in 1 2; do
S=$(date +%s%N)
timeout 900 python bench.py > $O/bench_default_$i.json 2> $O/bench_default_$i.err; echo "bench rc $?"
E=$(date +%s%N)
echo "run $i: bench wall $(( (E - S) / 1000000 )) ms" | tee -a $O/bench_wall.txt
done
python - <<'PY'
import json
for i in (1,2):
    d=json.load(open(f'gpurun_out/r6j/bench_default_{i}.json'))
    print("main %.4g rollouts/s, %.4f ms/step, valu frac %.3f, verified %s" % (d['value'], d['ms_per_step'], d['roofline_valu']['frac'], d['verified']['ok']))
    print(d.get('wall_s'))
    for k,v in d['cpu_baseline']['builds'].items(): print(' ', k, v['all_cores']['sample'], '%.3g'%v['all_cores']['value'])
PY
