#!/bin/bash
# A/B of variant libraries on launches of one wave per SIMD (development batch)
out=gpurun_out/$1; shift
mkdir -p $out
K="timeout 300 python tools/kbench.py $@"
$K --envs 64 --rollouts 2048 --horizon 50 --noise philox knots buffer tiled --rounds 8 --steps 40 > $out/kb_c4.txt 2>&1
$K --envs 128 --rollouts 1024 --horizon 50 --noise philox --rpl 2 --rounds 6 --steps 30 > $out/kb_128.txt 2>&1
$K --envs 32 --rollouts 4096 --horizon 100 --noise philox --rounds 6 --steps 30 > $out/kb_32x4096.txt 2>&1
grep -h "E=\|\.so" $out/kb_*.txt
