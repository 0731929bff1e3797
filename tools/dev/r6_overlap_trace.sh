#!/bin/bash
# Round 6: bench.py's `stream_overlap` (time of the groups alone / together, from the host's clock) against the SAME ratio derived from
# the kernel trace's time stamps of the same process (EnvGroups.overlap() traced).
export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
for CFG in C4 C3; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$CFG -- python3 tools/dev/groups_gather_cost.py --only none --config $CFG --groups 2 --reps 1 --steps 200 --overlap > $O/trace_$CFG.log 2>&1
  f=$(find $O/trace_$CFG -name "*kernel_trace.csv" | head -1)
  python3 tools/dev/groups_trace.py $f --last 150 --alone-phases --json $O/overlap_trace_$CFG.json > /dev/null
  grep -E "stream_overlap|us/step" $O/trace_$CFG.log
  python3 -c "
import json; d=json.load(open('$O/overlap_trace_$CFG.json')); print('$CFG from the trace:', d.get('alone_phases'))"
done
