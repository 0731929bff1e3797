#!/bin/bash
# Round 6: kernel traces of the grouped C4 with and without the per-step all-gather (one rank on real RCCL)
export TMPDIR=/tmp
O=gpurun_out/r6e; mkdir -p $O
for c in none gather-stamped; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$c -- python3 tools/dev/groups_gather_cost.py --only $c --reps 1 --steps 300 > $O/trace_$c.log 2>&1
  f=$(find $O/trace_$c -name "*kernel_trace.csv" | head -1)
  python3 tools/dev/groups_trace.py $f --last 200 --json $O/groups_trace_$c.json > /dev/null
  grep "us/step" $O/trace_$c.log
  python3 - <<PY
import json
d=json.load(open("$O/groups_trace_$c.json"))
print("$c", {k: d[k] for k in ("wall_us_per_step_of_all_groups","in_flight_together","concurrency")})
for s,p in d["per_stream"].items(): print("  stream", s, p)
for o in d["other_kernels_in_window"][:8]: print("  other", o)
PY
done
