#!/bin/bash
# Round 4 A/B in one process per shape: the round-start library (build_variants/r4_start.so) vs the current one
O=gpurun_out/r4; mkdir -p $O
NEW=cartpolesimulation_amd/libcpmppi.so; OLD=build_variants/r4_start.so
{
python tools/kbench.py $OLD $NEW --envs 8192 --rounds 5 --steps 5 --noise philox tiled buffer
python tools/kbench.py $OLD $NEW --envs 1024 --rounds 8 --steps 10 --noise philox
python tools/kbench.py $OLD $NEW --envs 256 --rounds 8 --steps 10 --noise philox
python tools/kbench.py $OLD $NEW --envs 64 --rollouts 2048 --horizon 50 --rounds 15 --steps 20 --noise philox buffer
python tools/kbench.py $OLD $NEW --envs 64 --rollouts 4096 --horizon 100 --rounds 10 --steps 10 --noise philox
python tools/kbench.py $OLD $NEW --envs 1 --rounds 15 --steps 20 --noise philox knots
} 2>/dev/null > $O/kbench_ab.txt
cat $O/kbench_ab.txt
