#!/bin/bash
# Round 4: longer same-process A/B of the edge-test variants at the headline shape (kernel times drift by +-3 % within a box)
O=gpurun_out/r4; mkdir -p $O
for i in 1 2 3; do
python tools/kbench.py build_variants/rb0.so build_variants/rb1.so cartpolesimulation_amd/libcpmppi.so --envs 8192 --rounds 40 --steps 5 --noise philox 2>/dev/null | tail -3
done > $O/kbench_rb_long.txt
python tools/kbench.py build_variants/rb0.so build_variants/rb1.so cartpolesimulation_amd/libcpmppi.so --envs 2048 --rounds 40 --steps 5 --noise philox 2>/dev/null | tail -3 >> $O/kbench_rb_long.txt
python tools/kbench.py build_variants/rb0.so build_variants/rb1.so cartpolesimulation_amd/libcpmppi.so --envs 8192 --rounds 20 --steps 5 --noise tiled buffer 2>/dev/null | tail -6 >> $O/kbench_rb_long.txt
cat $O/kbench_rb_long.txt
