#!/bin/bash
# round-3 GPU batch 3: A/B of kernel variants (kbench), tests, parity diagnostic with the envelope rule, collective
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
V=build_variants
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_ss0.so $V/r3_cur.so --envs 8192 --rounds 5 --steps 4 --noise philox > $O/kb_8192.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_ss0.so $V/r3_cur.so --envs 8192 --rounds 4 --steps 4 --noise tiled buffer > $O/kb_8192_buf.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_cur.so $V/r3_unp_s.so $V/r3_pk_s.so --envs 64 --rollouts 2048 --horizon 50 --rounds 8 --steps 20 --noise philox > $O/kb_c4.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_cur.so $V/r3_unp_s.so $V/r3_pk_s.so --envs 64 --rollouts 4096 --horizon 100 --rounds 8 --steps 10 --noise philox > $O/kb_c3.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_cur.so --envs 1 --rounds 8 --steps 30 --noise philox > $O/kb_single.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so $V/r3_cur.so $V/r3_unp_s.so --envs 256 --rounds 6 --steps 10 --noise philox > $O/kb_256.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_flags.json 2> $O/bench_rccl_flags.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python tools/dev/cfg_parity_diag.py C3 > $O/diag_c3.jsonl 2> $O/diag_c3.err
timeout 600 python tools/dev/cfg_parity_diag.py C4 > $O/diag_c4.jsonl 2> $O/diag_c4.err
tail -15 $O/pytest.log
