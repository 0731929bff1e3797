#!/bin/bash
# mid-size build vs throughput build by launch size (development batch): usage tools/dev/r3_cross.sh OUT lib...
out=gpurun_out/$1; shift
mkdir -p $out
for E in 64 128 256 512 768 1024 1536 2048 3072 4096 8192; do
  timeout 300 python tools/kbench.py $@ --envs $E --noise philox --rounds 5 --steps 10 > $out/kb_$E.txt 2>&1
  grep -h "E=\|\.so" $out/kb_$E.txt | cut -c1-120
done
