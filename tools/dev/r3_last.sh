#!/bin/bash
# Round 3, last check of the final tree: smoke, GPU suite, the default bench line (as the driver runs it), a kernel-trace profile of
# the predictor_ODE bench line
set -u
O=gpurun_out/r3last; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_ode -o run -- python3 $GRAFT_REPO_ROOT/bench.py --predictor-type ODE --no-extra-configs --no-cpu-baseline --no-single-env > $GRAFT_REPO_ROOT/$O/bench_ODE_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_ode.err
echo done
