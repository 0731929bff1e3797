#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
V=build_variants
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_near2.so $V/r3_ldr.so --envs 8192 --rounds 5 --steps 4 --noise buffer tiled philox > $O/kb_8192.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_near2.so $V/r3_ldr.so --envs 1 --rounds 8 --steps 30 --noise philox buffer > $O/kb_single.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_near2.so $V/r3_ldr.so --envs 64 --rollouts 2048 --horizon 50 --rounds 6 --steps 20 --noise philox buffer > $O/kb_c4.txt 2>&1
timeout 400 python bench.py --no-cpu-baseline --no-extra-configs > $O/bench_default.json 2> $O/bench_default.err
timeout 400 python bench.py --no-cpu-baseline --no-single-env --no-extra-configs --noise buffer-ref > $O/bench_bufferref.json 2> $O/bench_bufferref.err
timeout 300 python tools/dev/seam_latency.py > $O/seam.txt 2> $O/seam.err
tail -4 $O/pytest.log
