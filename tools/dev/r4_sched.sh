#!/bin/bash
# Round 4: scheduling strategy of the throughput and mid-size units re-measured after the triple edge test (variants built with
# CPMPPI_SCHED_THROUGHPUT / CPMPPI_SCHED_MID=<strategy> python __graft_entry__.py --variant sched_<strategy>)
O=gpurun_out/r4; mkdir -p $O
L="cartpolesimulation_amd/libcpmppi.so build_variants/sched_iterative-ilp.so build_variants/sched_max-memory-clause.so build_variants/sched_max-occupancy.so"
{
python tools/kbench.py $L --envs 8192 --rounds 30 --steps 5 --noise philox
python tools/kbench.py $L --envs 1024 --rounds 30 --steps 10 --noise philox
python tools/kbench.py $L --envs 64 --rollouts 4096 --horizon 100 --rounds 40 --steps 10 --noise philox
python tools/kbench.py $L --envs 64 --rollouts 2048 --horizon 50 --rounds 60 --steps 20 --noise philox
} 2>/dev/null > $O/kbench_sched.txt
grep -E "^E=|median" $O/kbench_sched.txt | cut -c1-150
