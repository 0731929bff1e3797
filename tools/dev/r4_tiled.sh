#!/bin/bash
# Round 4: tiled layout, weighted column sums as a reduce-scatter vs -DCPMPPI_TILED_SCATTER=0 (build_variants/ts0.so)
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "tiled or noise or layout or sampler or buffer or builds_of_the_kernel" 2>&1 | grep -E "passed|failed|Error" | tail -2
L="build_variants/ts0.so cartpolesimulation_amd/libcpmppi.so"
{
for i in 1 2; do python tools/kbench.py $L --envs 8192 --rounds 30 --steps 5 --noise tiled; done
python tools/kbench.py $L --envs 1024 --rounds 30 --steps 10 --noise tiled
python tools/kbench.py $L --envs 64 --rollouts 2048 --horizon 50 --rounds 40 --steps 20 --noise tiled
python tools/kbench.py $L --envs 1 --rounds 40 --steps 20 --noise tiled
} 2>/dev/null > $O/kbench_tiled.txt
grep -E "^E=|median" $O/kbench_tiled.txt | cut -c1-150
