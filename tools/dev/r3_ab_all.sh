#!/bin/bash
# A/B of variant libraries, Philox noise, every launch shape once (development batch)
out=gpurun_out/$1; shift
mkdir -p $out
K="timeout 300 python tools/kbench.py $@"
$K --envs 64 --rollouts 2048 --horizon 50 --noise philox --rounds 8 --steps 40 > $out/kb_c4.txt 2>&1
$K --envs 64 --rollouts 4096 --horizon 100 --noise philox --rounds 6 --steps 30 > $out/kb_c3.txt 2>&1
$K --envs 256 --noise philox --rounds 6 --steps 30 > $out/kb_256.txt 2>&1
$K --envs 1024 --noise philox --rounds 6 --steps 6 > $out/kb_1024.txt 2>&1
$K --envs 1 --rollouts 1024 --horizon 50 --noise philox --rounds 6 --steps 30 > $out/kb_single.txt 2>&1
$K --envs 8192 --noise philox --rounds 5 --steps 4 > $out/kb_8192.txt 2>&1
grep -h "E=\|\.so" $out/kb_*.txt
