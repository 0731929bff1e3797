#!/bin/bash
# Round 4: the working tree's library against the committed HEAD (build_variants/head.so, built from a git worktree), headline + sizes
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -2
L="build_variants/head.so cartpolesimulation_amd/libcpmppi.so"
{
for i in 1 2; do python tools/kbench.py $L --envs 8192 --rounds 40 --steps 5 --noise philox; done
python tools/kbench.py $L --envs 8192 --rounds 15 --steps 5 --noise tiled buffer
python tools/kbench.py $L --envs 1024 --rounds 30 --steps 10 --noise philox
python tools/kbench.py $L --envs 64 --rollouts 4096 --horizon 100 --rounds 40 --steps 10 --noise philox
python tools/kbench.py $L --envs 64 --rollouts 2048 --horizon 50 --rounds 60 --steps 20 --noise philox
python tools/kbench.py $L --envs 1 --rounds 40 --steps 20 --noise philox
} 2>/dev/null > $O/kbench_head_ab.txt
grep -E "^E=|median" $O/kbench_head_ab.txt | cut -c1-150
bash tools/dev/r4_pmc.sh 2>&1 | tail -2
