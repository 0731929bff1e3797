export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_gru
mkdir -p $OUT
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-env --predictor gru --envs 256"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $B > $OUT/pmc1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d $OUT/pmc2 -- python3 $B > $OUT/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B > $OUT/stats.log 2>&1
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_gru'
agg=collections.defaultdict(list)
for f in glob.glob(out+'/pmc*/**/*_counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gru_rollout' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(k, sum(v)/len(v), len(v))
for f in glob.glob(out+'/stats/**/*_kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gru_rollout' in r['Name']: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
