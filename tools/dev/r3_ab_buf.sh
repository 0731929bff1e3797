#!/bin/bash
# A/B of variant libraries on the mid-size shapes with buffer-fed noise (development batch)
out=gpurun_out/$1; shift
mkdir -p $out
K="timeout 300 python tools/kbench.py $@"
$K --envs 64 --rollouts 2048 --horizon 50 --noise philox buffer tiled knots --rounds 8 --steps 40 > $out/kb_c4.txt 2>&1
$K --envs 64 --rollouts 4096 --horizon 100 --noise buffer tiled --rounds 6 --steps 30 > $out/kb_c3.txt 2>&1
$K --envs 256 --noise buffer tiled --rounds 6 --steps 30 > $out/kb_256.txt 2>&1
$K --envs 1024 --noise buffer tiled --rounds 6 --steps 6 > $out/kb_1024.txt 2>&1
grep -h "E=\|\.so" $out/kb_*.txt
