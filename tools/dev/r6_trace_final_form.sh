#!/bin/bash
# Round 6: kernel trace of the grouped C4 / C3 with the per-step all-gather in its FINAL form (one folded waiter kernel per step)
export TMPDIR=/tmp
O=gpurun_out/r6r; mkdir -p $O
for CFG in C4 C3; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$CFG -- python3 tools/dev/groups_gather_cost.py --only gather-stamped --config $CFG --reps 1 --steps 300 > $O/trace_$CFG.log 2>&1
  f=$(find $O/trace_$CFG -name "*kernel_trace.csv" | head -1)
  python3 tools/dev/groups_trace.py $f --last 200 --json $O/groups_trace_${CFG}_gather_final_form.json > /dev/null
  grep "us/step" $O/trace_$CFG.log
  python3 - <<PY
import json
d=json.load(open("$O/groups_trace_${CFG}_gather_final_form.json"))
print("$CFG", {k: round(d[k],3) for k in ("wall_us_per_step_of_all_groups","in_flight_together","concurrency")})
for s,p in d["per_stream"].items(): print("  stream", s, {k: p[k] for k in ("kernels","duration_us_median","gap_us_median")})
for o in d["other_kernels_in_window"][:6]: print("  other", o)
PY
done
