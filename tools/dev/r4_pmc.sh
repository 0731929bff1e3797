#!/bin/bash
# quick SQ counters of the headline kernel (in-kernel Philox): instructions per rollout, VALU-busy
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/r4/pmc_quick_$(date +%H%M%S); mkdir -p $O
B2="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-env --no-extra-configs --no-verify"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d $O/sq -- python3 $B2 > $O/sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS --output-format csv -d $O/sq2 -- python3 $B2 > $O/sq2.log 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$O/sq*/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "rollout_cost_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in agg.items()}
print(m)
w=m.get("SQ_WAVES",65536)
print("VALU per wave %.0f  per rollout %.0f   SALU per wave %.0f  trans per wave %.0f" % (m["SQ_INSTS_VALU"]/w, m["SQ_INSTS_VALU"]/w/128*... if False else m["SQ_INSTS_VALU"]/w/2, m["SQ_INSTS_SALU"]/w, m.get("SQ_INSTS_VALU_TRANS",0)/w))
print("VALU busy %.3f" % (m["SQ_ACTIVE_INST_VALU"]*4/1024/(m["GRBM_GUI_ACTIVE"]/8)))
PY
