#!/bin/bash
export TMPDIR=/tmp
O=$(pwd)/gpurun_out/r4/valu_split; mkdir -p $O
for m in quiet bench; do
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/$m -- python3 tools/dev/valu_split.py $m > $O/$m.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for m in ("quiet","bench"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*_counter_collection.csv"%m,recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_cost_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    w=agg["SQ_WAVES"][0]
    print(m, "VALU per wave per launch:", [round(v/w) for v in agg["SQ_INSTS_VALU"]], "SALU", [round(v/w) for v in agg["SQ_INSTS_SALU"]])
PY
