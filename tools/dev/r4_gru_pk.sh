#!/bin/bash
# GRU (C5): packed gate math (two register pairs per stage) vs -DCPMPPI_GRU_PACKED_GATES=0, alternating runs on one box
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_gru.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" > $O/gru_pk_tests.txt
cat $O/gru_pk_tests.txt
for rep in 1 2 3; do for v in gru_pk0 cur; do
  if [ $v = cur ]; then unset CPMPPI_LIB; else export CPMPPI_LIB=build_variants/$v.so; fi
  python bench.py --predictor gru --envs 256 --steps 30 --warmup 5 --no-cpu-baseline --no-single-env --no-extra-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep $rep', '%.4g rollouts/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'], 'verified', d['verified']['ok'])"
done; done | tee $O/gru_pk_ab.txt
unset CPMPPI_LIB
