#!/bin/bash
# Round 6: the evidence run of the final tree -> gpurun_out/r6_final/ (copied into profiles/r6/ afterwards).
O=gpurun_out/r6_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"
python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
S=$(date +%s%N)
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
E=$(date +%s%N)
echo "python bench.py: wall $(( (E - S) / 1000000 )) ms" | tee $O/bench_wall.txt
# one rank on real RCCL: the per-step all-gather in the line, the grouped configurations under the collective
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 CPMPPI_BENCH_FORCE_COLLECTIVE=1 \
  python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err; echo "1-rank rc $?"
# TWO ranks on one device through the library's own communicator bound to the stand-in collective library (torch.distributed over gloo
# for the rendezvous and the reductions of the report): the whole N > 1 path of bench.py with a real peer
CPMPPI_BENCH_BACKEND=gloo CPMPPI_BENCH_ONE_DEVICE=1 CPMPPI_BENCH_COLLECTIVE=native CPMPPI_BENCH_RCCL_PATH=$PWD/tests/fake_rccl/libfake_rccl.so \
  timeout 900 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_2ranks_one_device_fake_rccl.json 2> $O/bench_2ranks.err; echo "2-rank rc $?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_final/bench_default.json'))
print("main %.4g rollouts/s, %.4f ms/step, valu frac %.3f, verified %s" % (d['value'], d['ms_per_step'], d['roofline_valu']['frac'], d['verified']['ok']))
for k,v in d['configs'].items():
    ver = v.get('verified') or {}
    print(k, "ms/step %.4f" % v['ms_per_step'], "verified", ver.get('ok'), "2nd", ver.get('second_stage_envs'), v.get('vs_one_launch_per_step'), v.get('stream_overlap'))
print(d.get('wall_s')); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'])
se=d['single_env']; print("single env %.1f us, verified %s" % (se['us_per_step'], se['verified']['ok']))
for f in ('bench_rccl_1rank.json', 'bench_2ranks_one_device_fake_rccl.json'):
    try:
        L=[l for l in open('gpurun_out/r6_final/'+f) if l.startswith('{"metric"')]
        d=json.loads(L[-1])
        print(f, "n_gpus", d['n_gpus'], "value %.4g" % d['value'], json.dumps(d['config'].get('collective'))[:600])
        for k,v in d.get('configs',{}).items():
            print("  ", k, {x: v.get(x) for x in ('ms_per_step','without_collective_ms_per_step','collective_cost','error')}, json.dumps(v.get('collective'))[:300], (v.get('verified') or {}).get('ok'))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 $O/bench_2ranks.err
