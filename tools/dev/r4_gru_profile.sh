#!/bin/bash
# Round 4: the GRU line of tools/profile.sh alone (the library's GRU kernel changed after the full profile run)
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/r4/gru_profile; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_gru -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single-env --no-extra-configs --predictor gru --envs 256 > $O/stats_gru.log 2>&1
find $O -name "*_kernel_stats.csv" | head -2; grep "^{" $O/stats_gru.log | tail -1 | cut -c1-300
