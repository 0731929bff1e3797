#!/bin/bash
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
V=build_variants
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_ss0.so $V/r3_near.so --envs 8192 --rounds 5 --steps 4 --noise philox tiled > $O/kb_8192.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_ss0.so $V/r3_near.so --envs 64 --rollouts 2048 --horizon 50 --rounds 8 --steps 20 --noise philox > $O/kb_c4.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_ss0.so $V/r3_near.so --envs 64 --rollouts 4096 --horizon 100 --rounds 8 --steps 10 --noise philox > $O/kb_c3.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_ss0.so $V/r3_near.so --envs 1 --rounds 8 --steps 30 --noise philox > $O/kb_single.txt 2>&1
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
bash tools/profile.sh r3 > $O/profile.log 2>&1
bash tools/profile_cfg.sh r3 "0" > $O/profile_cfg.log 2>&1
tail -4 $O/pytest.log
