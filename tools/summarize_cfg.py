#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/cfg_*/ (tools/profile_cfg.sh: kernel-trace stats + SQ/GRBM PMC of bench.py at
BASELINE's C3 / C4 configs) into profiles/<tag>/cfg_pmc.json."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
runs = sorted(d for d in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "cfg_*")) if os.path.isdir(d))
src = runs[-1]
out = {"source": os.path.relpath(src, ROOT)}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[4:]
    agg, meta = collections.defaultdict(list), {}
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_cost_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"kernel": r["Kernel_Name"].split("(")[0].split("::")[-1], "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                        "lds_block_bytes": int(r["LDS_Block_Size"]), "scratch_bytes": int(r["Scratch_Size"]), "grid_threads": int(r["Grid_Size"])}
    c = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
    rec = dict(meta, counters_mean_per_launch=c)
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        rec["cycles_per_xcd"] = cyc
        rec["valu_issue_cycles_per_simd"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0
        rec["valu_busy_fraction"] = rec["valu_issue_cycles_per_simd"] / cyc
        rec["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        rec["waves_per_simd"] = c["SQ_WAVES"] / 1024.0
    for f in glob.glob(os.path.join(src, "stats_" + name, "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "rollout_cost" in r["Name"]:
                rec["kernel_trace"] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                       "max_ns": float(r["MaxNs"])}
    log = os.path.join(src, "stats_" + name + ".log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            b = json.loads(lines[-1])
            rec["bench_line_under_rocprof"] = {k: b[k] for k in ("value", "ms_per_step", "config", "roofline_valu") if k in b}
    out[name] = rec
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "cfg_pmc.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, v.get("kernel"), "VALU busy", round(v.get("valu_busy_fraction", 0), 3), "waves/SIMD", v.get("waves_per_simd"),
              v.get("kernel_trace"))
