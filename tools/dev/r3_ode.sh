#!/bin/bash
# Round 3: predictor_type "ODE" on the GPU - its parity tests, the whole GPU suite, bench lines for both ODE predictors
set -u
O=gpurun_out/r3ode; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ode_predictor.py -q -x -m gpu > $O/pytest_ode.log 2>&1; echo "ode rc $?" >> $O/pytest_ode.log
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "all rc $?" >> $O/pytest.log
for pt in ODE ODE_v0; do
  timeout 600 python bench.py --predictor-type $pt --no-extra-configs > $O/bench_$pt.json 2> $O/bench_$pt.err
done
timeout 300 python bench.py --predictor-type ODE --math precise --no-cpu-baseline --no-single-env > $O/bench_ODE_precise.json 2> $O/bench_ODE_precise.err
timeout 300 python bench.py --predictor-type ODE --config C4 --no-cpu-baseline --no-single-env > $O/bench_ODE_C4.json 2> $O/bench_ODE_C4.err
timeout 300 python bench.py --predictor-type ODE --config C3 --no-cpu-baseline --no-single-env > $O/bench_ODE_C3.json 2> $O/bench_ODE_C3.err
timeout 300 python bench.py --predictor-type ODE --noise buffer --no-cpu-baseline --no-single-env > $O/bench_ODE_buffer.json 2> $O/bench_ODE_buffer.err
timeout 300 python bench.py --predictor-type ODE --noise buffer-ref --no-cpu-baseline --no-single-env > $O/bench_ODE_bufferref.json 2> $O/bench_ODE_bufferref.err
echo done
