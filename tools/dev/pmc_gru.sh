#!/bin/bash
# Runs on the GPU box: PMC passes (no tracing alongside) + a kernel-trace pass of the GRU bench, both arithmetic variants.
# -> gpurun_out/prof_gru/<math>/..., summary printed as JSON lines (copy into profiles/<tag>/gru_pmc.json)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for MATH in fast precise; do
  OUT=$R/gpurun_out/prof_gru/run_$(date +%H%M%S)_$MATH
  mkdir -p $OUT
  B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-env --predictor gru --envs 256 --math $MATH"
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $B > $OUT/pmc1.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -- python3 $B > $OUT/pmc2.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B > $OUT/stats.log 2>&1
  python3 - "$OUT" "$MATH" <<'PY'
import csv, glob, collections, json, sys
out, math = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/pmc*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gru_rollout' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
rec = {"math": math, "counters_mean_per_launch": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}
for f in glob.glob(out + '/stats/**/*_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gru_rollout' in r['Name']:
            rec["kernel"] = r['Name'].split('(')[0]
            rec["average_ns_under_kernel_trace"] = float(r['AverageNs'])
c = rec["counters_mean_per_launch"]
if "GRBM_GUI_ACTIVE" in c and "average_ns_under_kernel_trace" in rec:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    rec["cycles_per_xcd"] = cyc
    rec["effective_clock_ghz"] = cyc / rec["average_ns_under_kernel_trace"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        rec["mfma_busy_fraction_of_simd_time"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
    if "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
        rec["mfma_busy_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"]
print(json.dumps(rec))
PY
done
