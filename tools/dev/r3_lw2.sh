#!/bin/bash
mkdir -p gpurun_out/lw
for n in a b c d; do echo "== iterative-ilp variant $n (a: -disable-vector-combine, b: plain, c: a + -fno-slp-vectorize, d: -fno-slp-vectorize)"; timeout 200 ./build_variants/lone_wave_ii_$n | sed -n "1,17p"; done > gpurun_out/lw/lone_wave_flags.txt 2>&1
K="timeout 300 python tools/kbench.py cartpolesimulation_amd/libcpmppi.so"
$K --envs 1 --rollouts 1024 --horizon 50 --noise philox knots buffer --rounds 6 --steps 30 > gpurun_out/lw/kb_single_rpl0.txt 2>&1
$K --envs 1 --rollouts 1024 --horizon 50 --noise philox knots buffer --rounds 6 --steps 30 --rpl 2 > gpurun_out/lw/kb_single_rpl2.txt 2>&1
$K --envs 1 --rollouts 256 --horizon 20 --noise philox --rounds 6 --steps 30 > gpurun_out/lw/kb_c1_rpl0.txt 2>&1
$K --envs 1 --rollouts 256 --horizon 20 --noise philox --rounds 6 --steps 30 --rpl 2 > gpurun_out/lw/kb_c1_rpl2.txt 2>&1
$K --envs 16 --rollouts 1024 --horizon 50 --noise philox --rounds 6 --steps 30 > gpurun_out/lw/kb_16_rpl0.txt 2>&1
$K --envs 16 --rollouts 1024 --horizon 50 --noise philox --rounds 6 --steps 30 --rpl 2 > gpurun_out/lw/kb_16_rpl2.txt 2>&1
$K --envs 64 --rollouts 1024 --horizon 50 --noise philox --rounds 6 --steps 30 > gpurun_out/lw/kb_64_rpl0.txt 2>&1
$K --envs 64 --rollouts 1024 --horizon 50 --noise philox --rounds 6 --steps 30 --rpl 2 > gpurun_out/lw/kb_64_rpl2.txt 2>&1
cat gpurun_out/lw/lone_wave_flags.txt; for f in gpurun_out/lw/kb_*rpl*.txt; do echo $f; grep -h "E=\|\.so" $f; done
