#!/bin/bash
# SQ counters of the single-env (latency build) and C4 (lone-wave packed build) launches: instructions per wave by class, cycles per wave
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/r4/single_pmc; mkdir -p $O
for E in 1 128; do
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $O/e$E -- python3 tools/dev/valu_split.py bench $E > $O/e$E.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_VALU_TRANS SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $O/b$E -- python3 tools/dev/valu_split.py bench $E > $O/b$E.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for E in (1,128):
    agg=collections.defaultdict(list)
    for f in glob.glob("$O/[eb]%d/**/*_counter_collection.csv"%E,recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_cost_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"])); k=r["Kernel_Name"][:70]
    m={c:sum(v)/len(v) for c,v in agg.items()}
    w=m.get("SQ_WAVES",1)
    print("E=%d"%E, k, {c:round(v/w,1) for c,v in m.items()})
PY
