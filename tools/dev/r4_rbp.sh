#!/bin/bash
# Round 4: the triple edge test in the phased mid-size build - parity tests, then same-process A/B against -DCPMPPI_ROLLBACK_PHASED=0
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 > $O/rbp_tests.txt
tail -3 $O/rbp_tests.txt
A=build_variants/rbp0.so; B=cartpolesimulation_amd/libcpmppi.so
{
python tools/kbench.py $A $B --envs 64 --rollouts 2048 --horizon 50 --rounds 30 --steps 20 --noise philox buffer
python tools/kbench.py $A $B --envs 64 --rollouts 4096 --horizon 100 --rounds 20 --steps 10 --noise philox
python tools/kbench.py $A $B --envs 256 --rounds 20 --steps 10 --noise philox
python tools/kbench.py $A $B --envs 1024 --rounds 20 --steps 10 --noise philox
} 2>/dev/null > $O/kbench_rbp.txt
cat $O/kbench_rbp.txt
