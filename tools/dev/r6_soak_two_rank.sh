#!/bin/bash
# Round 6: soak of the two-process protocol - thousands of steps, a third of the all-gathers late by up to 1.5 ms on either rank,
# host naps, several seeds; one handle per rank and env groups under one communicator.  Every block must equal the single-process run.
O=gpurun_out/r6p; mkdir -p $O
for SEED in 7 8 9 10; do
  CPMPPI_TWO_RANK_STRESS="6000:300:1500:$SEED" timeout 800 python -m pytest tests/test_gpu_two_rank_gather.py -m gpu -x -q -k random_stalls 2>&1 | grep -E "passed|failed|Error|assert" | tail -3 | sed "s/^/seed $SEED: /" | tee -a $O/soak_two_rank.txt
done
