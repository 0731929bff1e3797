#!/bin/bash
# round-3 evidence run after the last kernel changes (GPU box): tests, bench lines, collective, profiles, parity statistics -> gpurun_out/r3final8/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3final8
mkdir -p $O
cd $R
V=build_variants
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for A in "--noise buffer" "--noise buffer-ref" "--math precise" "--config C3" "--config C4" "--predictor gru --envs 256" "--predictor gru --envs 256 --math precise"; do
  T=$(echo $A | tr -d ' -'); timeout 400 python bench.py --no-cpu-baseline --no-single-env --no-extra-configs $A > $O/bench_$T.json 2> $O/bench_$T.err
done
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_1rank.json 2> $O/bench_rccl_1rank.err
CPMPPI_BENCH_COLLECTIVE=native-events timeout 400 python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_1rank_events.json 2> $O/bench_rccl_1rank_events.err
CPMPPI_BENCH_COLLECTIVE=torch timeout 400 python bench.py --gpus 1 --no-cpu-baseline --no-single-env > $O/bench_rccl_1rank_torch.json 2> $O/bench_rccl_1rank_torch.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4_collective -- python3 bench.py --gpus 1 --config C4 --steps 60 --warmup 10 --no-cpu-baseline --no-single-env --no-extra-configs > $O/trace_c4_collective.json 2> $O/trace_c4_collective.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
CPMPPI_BENCH_BACKEND=gloo CPMPPI_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err
timeout 300 python tools/dev/seam_latency.py > $O/seam_latency.txt 2> $O/seam_latency.err
timeout 900 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 8192 --rounds 5 --steps 4 --noise philox tiled buffer > $O/kbench_8192.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 64 --rollouts 2048 --horizon 50 --rounds 8 --steps 40 --noise philox buffer > $O/kbench_c4.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 64 --rollouts 4096 --horizon 100 --rounds 6 --steps 30 --noise philox > $O/kbench_c3.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 1 --rounds 8 --steps 30 --noise philox knots buffer tiled > $O/kbench_single.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 1 --rollouts 256 --horizon 20 --rounds 8 --steps 30 --noise philox knots > $O/kbench_c1.txt 2>&1
timeout 900 python tools/fuzz_parity.py --costs qbgm default legacy > $O/fuzz_parity.json 2> $O/fuzz_parity.err
timeout 600 python tools/dev/parity_buckets.py > $O/parity_buckets.jsonl 2> $O/parity_buckets.err
timeout 600 python tools/dev/cfg_parity_diag.py C3 > $O/cfg_parity_diag_C3.jsonl 2> $O/cfg_parity_diag_C3.err
timeout 600 python tools/dev/cfg_parity_diag.py C4 > $O/cfg_parity_diag_C4.jsonl 2> $O/cfg_parity_diag_C4.err
timeout 600 python tools/baseline_table.py > $O/baseline_table.json 2> $O/baseline_table.err
bash tools/profile.sh r3 > $O/profile.log 2>&1
bash tools/profile_cfg.sh r3 "0" > $O/profile_cfg.log 2>&1
tail -3 $O/pytest.log
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 256 --rounds 6 --steps 30 --noise philox > $O/kbench_256.txt 2>&1
timeout 600 python tools/kbench.py $V/r3_base.so cartpolesimulation_amd/libcpmppi.so --envs 1024 --rounds 6 --steps 6 --noise philox > $O/kbench_1024.txt 2>&1
CPMPPI_LIB=$V/sec.so timeout 250 python tools/dev/sections.py --config C4 > $O/sections_C4.txt 2>&1
CPMPPI_LIB=$V/sec.so timeout 250 python tools/dev/sections.py --config C3 > $O/sections_C3.txt 2>&1
CPMPPI_LIB=$V/sec.so timeout 250 python tools/dev/sections.py --config C2 > $O/sections_C2.txt 2>&1
timeout 250 $V/lone_wave > $O/lone_wave.txt 2>&1
unset RANK; export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 CPMPPI_BENCH_FORCE_COLLECTIVE=1
timeout 400 python bench.py --gpus 1 --no-cpu-baseline > $O/bench_rccl_1rank_b.json 2> $O/bench_rccl_1rank_b.err
unset RANK WORLD_SIZE LOCAL_RANK MASTER_ADDR MASTER_PORT CPMPPI_BENCH_FORCE_COLLECTIVE
timeout 400 python bench.py --no-cpu-baseline > $O/bench_default_b.json 2> $O/bench_default_b.err
tail -3 $O/pytest.log
