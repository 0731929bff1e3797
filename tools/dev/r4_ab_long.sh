#!/bin/bash
# Round 4: longer same-process A/B (kernel times drift by +-3 % within a box): round-start library, the tree before the per-env
# fold + triple edge test (build_variants/pre_fold.so), the current library
O=gpurun_out/r4; mkdir -p $O
L="build_variants/r4_start.so build_variants/pre_fold.so cartpolesimulation_amd/libcpmppi.so"
{
python tools/kbench.py $L --envs 8192 --rounds 30 --steps 5 --noise philox
python tools/kbench.py $L --envs 1 --rounds 40 --steps 20 --noise philox knots
python tools/kbench.py $L --envs 64 --rollouts 2048 --horizon 50 --rounds 40 --steps 20 --noise philox
python tools/kbench.py $L --envs 64 --rollouts 4096 --horizon 100 --rounds 30 --steps 10 --noise philox
} 2>/dev/null > $O/kbench_ab_long.txt
cat $O/kbench_ab_long.txt
