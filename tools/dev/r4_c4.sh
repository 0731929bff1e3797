#!/bin/bash
# Round 4: C4 / C3 / 256-env kernel times, tree before the fold + triple edge test vs the current library, many rounds
O=gpurun_out/r4; mkdir -p $O
L="build_variants/pre_fold.so cartpolesimulation_amd/libcpmppi.so"
{
for i in 1 2; do python tools/kbench.py $L --envs 64 --rollouts 2048 --horizon 50 --rounds 100 --steps 20 --noise philox; done
python tools/kbench.py $L --envs 64 --rollouts 4096 --horizon 100 --rounds 60 --steps 10 --noise philox
python tools/kbench.py $L --envs 256 --rounds 60 --steps 10 --noise philox
} 2>/dev/null > $O/kbench_c4.txt
cat $O/kbench_c4.txt
