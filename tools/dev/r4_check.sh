#!/bin/bash
# Round-4 GPU check: the -m gpu suite, smoke, the default bench line (with its `verified` objects).  Outputs under gpurun_out/r4/.
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r4/gpu_tests.txt
echo "pytest rc=$?" >> gpurun_out/r4/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke.txt 2>&1
python bench.py > gpurun_out/r4/bench_default.json 2> gpurun_out/r4/bench_default.err
echo "bench rc=$?" >> gpurun_out/r4/bench_default.err
tail -3 gpurun_out/r4/gpu_tests.txt; tail -2 gpurun_out/r4/bench_default.err; head -c 600 gpurun_out/r4/bench_default.json
