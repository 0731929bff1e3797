#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/groups_*/ (tools/profile_groups.sh) into profiles/<tag>/cfg_pmc.json: for C4 / C3 as one launch per
step and as two env groups -
  * from the kernel trace (tools/dev/groups_trace.py): kernel duration per stream, wall time per step of all envs, the fraction of
    the window in which rollout kernels of two different groups are in flight TOGETHER, their mean concurrency;
  * from the PMC pass (kernels serialised by the profiler: an instruction count does not depend on that): VALU instructions per
    rollout, VALU issue cycles per SIMD and launch; and the DEVICE's VALU-busy fraction of the un-profiled run = issue cycles of
    all the launches of a step / (wall time per step x the clock measured by GRBM_GUI_ACTIVE over the kernel's traced duration);
  * the un-profiled wall time per step and bench.py's `stream_overlap` (time of the groups alone / together) of the same build."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
runs = sorted(d for d in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "groups_*")) if os.path.isdir(d))
src = runs[-1]
out = {"source": os.path.relpath(src, ROOT)}
SHAPES = {"C4": (64, 2048, 50), "C3": (64, 4096, 100)}
for cfg in ("C4", "C3"):
    for g in (1, 2):
        name = f"{cfg}_g{g}"
        rec = {"envs": SHAPES[cfg][0], "rollouts": SHAPES[cfg][1], "horizon": SHAPES[cfg][2], "groups": g}
        log = os.path.join(src, f"plain_{name}.log")
        if os.path.exists(log):
            txt = open(log).read()
            m = re.search(r"([\d.]+) us/step", txt)
            if m:
                rec["wall_us_per_step_unprofiled"] = float(m.group(1))
            m = re.search(r"stream_overlap ([\d.]+)", txt)
            if m:
                rec["stream_overlap_alone_over_together"] = float(m.group(1))
        tr = glob.glob(os.path.join(src, f"trace_{name}", "**", "*_kernel_trace.csv"), recursive=True)
        if tr:
            j = os.path.join(src, f"groups_trace_{name}.json")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "groups_trace.py"), tr[0], "--last", "250", "--json", j],
                           check=True, stdout=subprocess.DEVNULL)
            t = json.load(open(j))
            rec["trace"] = {k: t[k] for k in ("wall_us_per_step_of_all_groups", "in_flight_together", "at_least_one_in_flight", "concurrency",
                                             "steps_in_window", "per_stream")}
        agg, meta = collections.defaultdict(list), {}
        for f in glob.glob(os.path.join(src, f"pmc_{name}", "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "rollout_cost_kernel" in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {"kernel": r["Kernel_Name"].split("(")[0].split("::")[-1], "vgpr": int(r["VGPR_Count"]),
                            "lds_block_bytes": int(r["LDS_Block_Size"]), "scratch_bytes": int(r["Scratch_Size"]), "grid_threads": int(r["Grid_Size"])}
        c = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
        if c:
            rec["pmc"] = dict(meta, counters_mean_per_launch=c)
            E, N, H = SHAPES[cfg]
            per_launch_rollouts = E * N / g
            p = rec["pmc"]
            p["valu_instructions_per_rollout"] = c["SQ_INSTS_VALU"] * 64.0 / meta["grid_threads"] / (per_launch_rollouts / meta["grid_threads"])
            p["waves_per_simd_per_launch"] = c["SQ_WAVES"] / 1024.0
            p["valu_issue_cycles_per_simd_per_launch"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0
            p["valu_busy_fraction_of_the_serialised_launch"] = p["valu_issue_cycles_per_simd_per_launch"] / (c["GRBM_GUI_ACTIVE"] / 8.0)
            wall = rec.get("wall_us_per_step_unprofiled")
            if wall and "trace" in rec:
                # the clock: cycles per XCD of the serialised launch over its duration in that same PMC pass is not available (no
                # trace there), so the traced duration of the ONE-launch form calibrates it: see `clock_ghz` below
                rec["_issue_cycles_per_step"] = g * p["valu_issue_cycles_per_simd_per_launch"]
        out[name] = rec
# clock from the one-launch forms: GRBM_GUI_ACTIVE / 8 cycles per XCD over the launch's traced duration
for cfg in ("C4", "C3"):
    one = out.get(f"{cfg}_g1", {})
    try:
        dur_us = list(one["trace"]["per_stream"].values())[0]["duration_us_median"]
        ghz = one["pmc"]["counters_mean_per_launch"]["GRBM_GUI_ACTIVE"] / 8.0 / (dur_us * 1e3)
    except (KeyError, IndexError):
        continue
    for g in (1, 2):
        r = out.get(f"{cfg}_g{g}", {})
        if "_issue_cycles_per_step" in r:
            r["clock_ghz_from_the_one_launch_form"] = ghz
            r["device_valu_busy_fraction"] = r.pop("_issue_cycles_per_step") / (r["wall_us_per_step_unprofiled"] * 1e3 * ghz)
            # bench.py's `stream_overlap` is time(groups one after the other) / time(together); the same ratio from independent
            # evidence: a group's launch ALONE (the PMC pass serialises kernels: GRBM_GUI_ACTIVE / 8 cycles per XCD) x groups over
            # the un-profiled wall time per step
            alone_us = r["pmc"]["counters_mean_per_launch"]["GRBM_GUI_ACTIVE"] / 8.0 / (ghz * 1e3)
            r["launch_alone_us_from_counters"] = alone_us
            r["alone_over_together_from_counters"] = g * alone_us / r["wall_us_per_step_unprofiled"]
dst = os.path.join(ROOT, "profiles", tag)
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, "cfg_pmc.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        t = v.get("trace", {})
        print(k, "wall", v.get("wall_us_per_step_unprofiled"), "us; traced", round(t.get("wall_us_per_step_of_all_groups", 0), 1), "us; together",
              round(t.get("in_flight_together", 0), 3), "concurrency", round(t.get("concurrency", 0), 3), "| bench overlap",
              v.get("stream_overlap_alone_over_together"), "from counters", round(v.get("alone_over_together_from_counters", 0), 3), "| device VALU busy", round(v.get("device_valu_busy_fraction", 0), 3),
              "instr/rollout", round(v.get("pmc", {}).get("valu_instructions_per_rollout", 0)))
