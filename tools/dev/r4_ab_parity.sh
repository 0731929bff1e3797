#!/bin/bash
# C3 full-width parity diagnosis per library variant (tools/dev/cfg_parity_diag.py prints launch-wide totals)
O=gpurun_out/r4; mkdir -p $O
for v in "" wt0 acc0; do
  if [ -n "$v" ]; then export CPMPPI_LIB=build_variants/$v.so; else unset CPMPPI_LIB; fi
  python tools/dev/cfg_parity_diag.py C3 2>/dev/null | tail -1 > $O/diag_C3_${v:-lib}.json
  python - <<PY
import json
d=json.load(open("$O/diag_C3_${v:-lib}.json"))["totals"]
for k,t in d.items(): print("${v:-lib}", k, {x:t[x] for x in ("env_clear_off","env_sens_off","env_worst_clear","u_worst","clear_off","flagged_off")})
PY
done
