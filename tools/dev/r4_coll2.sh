#!/bin/bash
# Round 4: does the one-rank RCCL gather slow a workload by itself, or only as a LATER configuration of a process that already
# ran others?  predictor_ODE and GRU lines alone, plain vs forced collective, alternating.
O=gpurun_out/r4; mkdir -p $O
COMMON="--no-cpu-baseline --no-single-env --no-extra-configs --no-verify"
run() { python bench.py $@ 2>/dev/null | python -c "
import sys,json
L=[l for l in sys.stdin if l.startswith('{\"metric\"')]
d=json.loads(L[-1]); print('%.4f ms/step  kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))"; }
for rep in 1 2; do
echo "ODE plain   $(run $COMMON --predictor-type ODE)"
echo "ODE gather  $(RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2960$rep CPMPPI_BENCH_FORCE_COLLECTIVE=1 run --gpus 1 $COMMON --predictor-type ODE)"
echo "GRU plain   $(run $COMMON --predictor gru --envs 256)"
echo "GRU gather  $(RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2961$rep CPMPPI_BENCH_FORCE_COLLECTIVE=1 run --gpus 1 $COMMON --predictor gru --envs 256)"
echo "main plain  $(run $COMMON)"
echo "main gather $(RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2962$rep CPMPPI_BENCH_FORCE_COLLECTIVE=1 run --gpus 1 $COMMON)"
done | tee $O/coll_alone.txt
