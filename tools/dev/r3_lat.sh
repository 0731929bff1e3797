#!/bin/bash
# A/B of variant libraries on the latency build's shapes (development batch): usage  tools/dev/r3_lat.sh OUTDIR lib_a.so lib_b.so ...
out=gpurun_out/$1; shift
mkdir -p $out
K="timeout 300 python tools/kbench.py $@"
$K --envs 1 --rollouts 1024 --horizon 50 --noise philox knots buffer tiled --rounds 6 --steps 30 > $out/kb_single.txt 2>&1
$K --envs 1 --rollouts 256 --horizon 20 --noise philox knots --rounds 6 --steps 30 > $out/kb_c1.txt 2>&1
$K --envs 1 --rollouts 3500 --horizon 35 --noise philox buffer --rounds 6 --steps 30 > $out/kb_3500.txt 2>&1
$K --envs 16 --rollouts 1024 --horizon 50 --noise philox --rounds 6 --steps 30 > $out/kb_16.txt 2>&1
$K --envs 64 --rollouts 1024 --horizon 50 --noise philox --rounds 6 --steps 30 > $out/kb_64.txt 2>&1
grep -h "E=\|\.so" $out/kb_*.txt
