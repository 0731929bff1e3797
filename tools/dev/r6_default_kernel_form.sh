#!/bin/bash
# Round 6, last change: the side stream's default form is the folded waiter kernel.  Tests, then the bench lines with one rank on RCCL
# in both forms on one box.
O=gpurun_out/r6q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_pipeline.py tests/test_gpu_two_rank_gather.py tests/test_gpu_c_consumer.py tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for FORM in kernel stream-ops; do
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 CPMPPI_BENCH_FORCE_COLLECTIVE=1 CPMPPI_COMM_WAITER=$FORM \
    python bench.py --gpus 1 --no-cpu-baseline --no-single-env --no-verify > $O/bench_rccl_1rank_$FORM.json 2> $O/bench_rccl_1rank_$FORM.err; echo "$FORM rc $?"
done
python bench.py --no-cpu-baseline --no-single-env --no-verify > $O/bench_plain.json 2>/dev/null
python - <<'PY'
import json
def load(f):
    L=[l for l in open(f) if l.startswith('{"metric"')]; return json.loads(L[-1])
p=load('gpurun_out/r6q/bench_plain.json')
print("plain: main %.4f ms  C3 %.4f C4 %.4f" % (p['ms_per_step'], p['configs']['C3']['ms_per_step'], p['configs']['C4']['ms_per_step']))
for form in ('kernel','stream-ops'):
    d=load(f'gpurun_out/r6q/bench_rccl_1rank_{form}.json')
    c=d['configs']
    print(form, "main %.4f ms (%.3f x)  C3 %.4f (%.3f x)  C4 %.4f (%.3f x)  C4_pipelined cost %.3f  C3_pipelined cost %.3f  stream_memory_ops %s" % (
        d['ms_per_step'], d['ms_per_step']/p['ms_per_step'], c['C3']['ms_per_step'], c['C3']['ms_per_step']/p['configs']['C3']['ms_per_step'],
        c['C4']['ms_per_step'], c['C4']['ms_per_step']/p['configs']['C4']['ms_per_step'], c['C4_pipelined']['collective_cost'], c['C3_pipelined']['collective_cost'],
        d['config']['collective']['stream_memory_ops']))
PY
