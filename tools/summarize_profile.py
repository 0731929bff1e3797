#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into profiles/<tag>/ (kernel stats, PMC summary, pmc_traffic.json)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
runs = sorted(d for d in glob.glob(os.path.join(src, "run_*")) if os.path.isdir(d))
if runs:                      # tools/profile.sh writes one sub-directory per run; take the latest
    src = runs[-1]
print("summarising", src)
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def short(name):
    """'void (anonymous namespace)::kernel<...>(args)' / 'void cpmppi_k::kernel<...>(args)' -> 'kernel<...>'"""
    for ns in ("(anonymous namespace)::", "cpmppi_k::"):
        if ns in name:
            return name.split(ns, 1)[1].split("(")[0]
    return None


summary = {}
for noise in ("philox", "buffer", "buffer-ref"):
    ks = glob.glob(os.path.join(src, f"stats_{noise}", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(dst, f"kernel_stats_{noise}.csv"))
    log = os.path.join(src, f"stats_{noise}.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            open(os.path.join(dst, f"bench_line_under_rocprof_{noise}.json"), "w").write(lines[-1])
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, f"pmc_*_{noise}", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
                agg[(k, "_VGPR_Count")].append(float(r["VGPR_Count"]))
                agg[(k, "_LDS_Block_Size")].append(float(r["LDS_Block_Size"]))
    out = {}
    for (k, c), v in sorted(agg.items()):
        out.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    summary[noise] = out
ks = glob.glob(os.path.join(src, "stats_gru", "**", "*_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(dst, "kernel_stats_gru.csv"))
    lines = [l for l in open(os.path.join(src, "stats_gru.log")) if l.startswith("{")]
    if lines:
        open(os.path.join(dst, "bench_line_under_rocprof_gru.json"), "w").write(lines[-1])
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)

# HBM-side traffic of the dominant kernel per launch: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
# one half of the bytes of a coalesced read (MI355X_MICROARCH.md, HBM).  tools/dev/fetch_calib.hip confirmed the same
# factor for this library's 4-byte-per-lane loads (1 GiB read once -> FETCH_SIZE 0.5 GiB), so the reported figure is
# 2 x FETCH_SIZE + WRITE_SIZE; the raw sum is kept beside it.
rec = {}
for noise, out in summary.items():
    for k, c in out.items():
        if k.startswith("rollout_cost_kernel") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f, w = c["FETCH_SIZE"]["mean_per_launch"] * 1024, c["WRITE_SIZE"]["mean_per_launch"] * 1024
            rec[noise] = {"kernel": k, "fetch_bytes_raw": f, "write_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
                          "hbm_bytes_per_launch_uncorrected": f + w}
bench = {}
for noise in rec:
    p = os.path.join(dst, f"bench_line_under_rocprof_{noise}.json")
    if os.path.exists(p):
        cfg = json.load(open(p))["config"]
        rec[noise].update(E=cfg["envs_per_gpu"], N=cfg["rollouts"], H=cfg["horizon"], noise=noise)
import datetime
for r in rec.values():
    r["collected"] = datetime.date.today().isoformat() + " (tools/profile.sh " + tag + ", separate --pmc passes)"
json.dump(rec, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
