#!/bin/bash
# Round 4: per-env constants from fold_env_kernel (CPMPPI_ENV_FOLD) - throughput-shape parity tests, A/B against the in-kernel fold
# (build_variants/fold0.so = this tree with -DCPMPPI_ENV_FOLD=0), SQ counters.
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 > $O/fold_tests.txt
tail -3 $O/fold_tests.txt
python tools/kbench.py build_variants/fold0.so cartpolesimulation_amd/libcpmppi.so --envs 8192 --rounds 6 --steps 5 --noise philox tiled buffer 2>/dev/null > $O/kbench_fold.txt
python tools/kbench.py build_variants/fold0.so cartpolesimulation_amd/libcpmppi.so --envs 3072 --rounds 6 --steps 8 --noise philox 2>/dev/null >> $O/kbench_fold.txt
cat $O/kbench_fold.txt
bash tools/dev/r4_pmc.sh 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-single-env --no-extra-configs > $O/bench_fold.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_fold.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['verified']['ok'])"
