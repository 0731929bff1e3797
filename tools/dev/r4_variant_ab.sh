#!/bin/bash
# Round 4 A/B of the throughput build edge test: build_variants/rb0.so (-DCPMPPI_ROLLBACK=0: per substep), rb1.so (once per control
# step), the library (once per three substeps) - parity tests of the throughput shapes, same-process kernel times, SQ counters.
# (variants: python __graft_entry__.py --variant rb0 -DCPMPPI_ROLLBACK=0; rb1 = the one-test-per-control-step form, removed from the
# source after this measurement - git show 65f3aaf; for the per-env fold: -DCPMPPI_ENV_FOLD=0)
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 > $O/rb_tests.txt
tail -3 $O/rb_tests.txt
python tools/kbench.py build_variants/rb0.so build_variants/rb1.so cartpolesimulation_amd/libcpmppi.so --envs 8192 --rounds 8 --steps 5 --noise philox tiled buffer 2>/dev/null > $O/kbench_rb.txt
python tools/kbench.py build_variants/rb0.so build_variants/rb1.so cartpolesimulation_amd/libcpmppi.so --envs 3072 --rounds 6 --steps 8 --noise philox 2>/dev/null >> $O/kbench_rb.txt
cat $O/kbench_rb.txt
bash tools/dev/r4_pmc.sh 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-single-env --no-extra-configs > $O/bench_rb.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_rb.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['verified']['ok'])"
