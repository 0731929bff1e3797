#!/bin/bash
# GRU (C5) A/B on one box: -DCPMPPI_GRU_INTERLEAVE=0 variant vs current, alternating runs of bench.py --predictor gru --envs 256; phase stamps
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_gru.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" > $O/gru_tests.txt
cat $O/gru_tests.txt
for rep in 1 2 3; do for v in gru_il0 cur; do
  if [ $v = cur ]; then unset CPMPPI_LIB; else export CPMPPI_LIB=build_variants/$v.so; fi
  python bench.py --predictor gru --envs 256 --steps 30 --warmup 5 --no-cpu-baseline --no-single-env --no-extra-configs --no-verify 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v rep $rep', '%.4g rollouts/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])"
done; done | tee $O/gru_ab.txt
unset CPMPPI_LIB
for v in gru_st0 gru_st1; do echo "== $v"; python tools/gru_stamps.py build_variants/$v.so 256 2>/dev/null | tail -8; done | tee $O/gru_stamps.txt
