#!/bin/bash
# Round 4: C4 alone with the forced one-rank collective: does the timed window's position matter (RCCL's first calls)?
COMMON="--no-cpu-baseline --no-single-env --no-extra-configs --no-verify --config C4"
run() { python bench.py $@ 2>/dev/null | python -c "
import sys,json
L=[l for l in sys.stdin if l.startswith('{\"metric\"')]
d=json.loads(L[-1]); print('%.4f ms/step  kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 CPMPPI_BENCH_FORCE_COLLECTIVE=1
for sw in "200 20" "200 100" "200 400" "400 50" "2000 50"; do set -- $sw
echo "gather steps $1 warmup $2: $(MASTER_PORT=298$((RANDOM%90+10)) run --gpus 1 $COMMON --steps $1 --warmup $2)"
done
unset RANK WORLD_SIZE LOCAL_RANK CPMPPI_BENCH_FORCE_COLLECTIVE
for sw in "200 20" "400 50" "2000 50"; do set -- $sw
echo "plain  steps $1 warmup $2: $(run $COMMON --steps $1 --warmup $2)"
done
