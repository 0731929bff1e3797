#!/usr/bin/env python3
"""Development tool (CPU): what a `rocprofv3 --kernel-trace` CSV says about ENV GROUPS (and their per-step all-gather).

From the start / end time stamps of the rollout kernels, per stream (= per env group):
  * kernel duration and the gap between consecutive kernels of a stream,
  * `in_flight_together` - the fraction of the traced window in which rollout kernels of at least two different streams are in
    flight at the same time - and `concurrency` = sum of all rollout kernel durations / the time at least one is in flight
    (1.0 = the groups ran one after the other, G = always all G side by side),
  * wall time per step of all groups = the window / steps,
and what else ran (the side stream: RCCL's kernel, the stamp kernel; with their durations and where they fall).
Only the steady state is looked at: the last `--last` rollout kernels of every stream.

  python tools/dev/groups_trace.py <kernel_trace.csv> [--last 150] [--json out.json]
"""
import argparse
import csv
import json
import statistics as st

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--last", type=int, default=150)
ap.add_argument("--json", default=None)
ap.add_argument("--alone-phases", action="store_true",
                help="the traced program also ran EnvGroups.overlap() (every group's steps alone, then all together): derive "
                     "bench.py's `stream_overlap` = time alone / time together from the trace's own time stamps")
args = ap.parse_args()

rows = list(csv.DictReader(open(args.trace)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
roll = [r for r in rows if "rollout_cost_kernel" in r["Kernel_Name"]]
streams = {}
for r in sorted(roll, key=lambda r: r["s"]):
    streams.setdefault(r["Stream_Id"], []).append(r)
# the steady state: the last N kernels of the streams that carry most of the launches
n_max = max(len(v) for v in streams.values())
streams = {k: v[-args.last:] for k, v in streams.items() if len(v) >= n_max // 2}
t0 = max(v[0]["s"] for v in streams.values())
t1 = min(v[-1]["e"] for v in streams.values())
ks = [r for v in streams.values() for r in v if r["s"] >= t0 and r["e"] <= t1]
# sweep: how many rollout kernels of DIFFERENT streams are in flight
ev = sorted([(r["s"], 1, r["Stream_Id"]) for r in ks] + [(r["e"], -1, r["Stream_Id"]) for r in ks])
live, last, busy1, busy2 = {}, None, 0, 0
for t, d, sid in ev:
    if last is not None:
        act = sum(1 for c in live.values() if c > 0)
        busy1 += (t - last) if act >= 1 else 0
        busy2 += (t - last) if act >= 2 else 0
    live[sid] = live.get(sid, 0) + d
    last = t
window = t1 - t0
per = {}
for sid, v in streams.items():
    vv = [r for r in v if r["s"] >= t0 and r["e"] <= t1]
    per[sid] = {"kernels": len(vv), "duration_us_median": st.median((r["e"] - r["s"]) / 1e3 for r in vv),
                "gap_us_median": st.median((b["s"] - a["e"]) / 1e3 for a, b in zip(vv, vv[1:])) if len(vv) > 1 else None,
                "kernel": vv[0]["Kernel_Name"].split("(")[0], "grid": int(vv[0]["Grid_Size_X"]) // int(vv[0]["Workgroup_Size_X"]),
                "vgpr": int(vv[0]["VGPR_Count"])}
steps = min(p["kernels"] for p in per.values())
others = {}
for r in rows:
    if "rollout_cost_kernel" in r["Kernel_Name"] or not (t0 <= r["s"] <= t1):
        continue
    o = others.setdefault((r["Kernel_Name"].split("(")[0][:90], r["Stream_Id"]), [])
    o.append((r["e"] - r["s"]) / 1e3)
out = {"trace": args.trace, "window_us": window / 1e3, "streams_with_rollout_kernels": len(per), "per_stream": per,
       "steps_in_window": steps, "wall_us_per_step_of_all_groups": window / 1e3 / steps,
       "in_flight_together": busy2 / window, "at_least_one_in_flight": busy1 / window,
       "concurrency": sum(r["e"] - r["s"] for r in ks) / busy1,
       "other_kernels_in_window": [{"kernel": k[0], "stream": k[1], "count": len(v), "duration_us_median": st.median(v)} for k, v in
                                   sorted(others.items(), key=lambda kv: -len(kv[1]))]}
if args.alone_phases:
    # EnvGroups.overlap(): `steps` launches of every group ALONE, one group after the other, then the same launches of all groups
    # together.  A kernel is "solo" if no rollout kernel of another stream is in flight at any time of its life; a run of >= 20
    # consecutive solo kernels of one stream is that group's alone phase, the interleaved kernels right behind the LAST alone phase
    # are the together phase (as many launches per stream as an alone phase had).
    allk = sorted(roll, key=lambda r: r["s"])
    by_stream = {}
    for r in allk:
        by_stream.setdefault(r["Stream_Id"], []).append(r)
    import bisect
    starts = {sid: [r["s"] for r in v] for sid, v in by_stream.items()}

    def solo(r):
        for sid, v in by_stream.items():
            if sid == r["Stream_Id"]:
                continue
            i = bisect.bisect_left(starts[sid], r["e"])          # kernels of the other stream starting before this one ends ...
            j = i - 1
            while j >= 0 and v[j]["s"] > r["s"] - 10_000_000:    # ... that end after it starts (look back 10 ms at most)
                if v[j]["e"] > r["s"]:
                    return False
                j -= 1
        return True
    flags = [(r, solo(r)) for r in allk]
    runs, cur = [], []
    for r, so in flags:
        if so and (not cur or cur[-1]["Stream_Id"] == r["Stream_Id"]):
            cur.append(r)
        else:
            if len(cur) >= 20:
                runs.append(cur)
            cur = [r] if so else []
    if len(cur) >= 20:
        runs.append(cur)
    # the LAST alone phase of every stream (overlap() is the last thing the tool runs per repetition; the final repetition counts)
    last = {}
    for run in runs:
        last[run[0]["Stream_Id"]] = run
    if len(last) >= 2:
        n = min(len(v) for v in last.values())
        alone_us = {sid: (v[-1]["e"] - v[0]["s"]) / 1e3 / len(v) for sid, v in last.items()}
        t_end = max(v[-1]["e"] for v in last.values())
        tog = [r for r in allk if r["s"] >= t_end]
        per = {}
        for r in tog:
            per.setdefault(r["Stream_Id"], []).append(r)
        per = {sid: v[:n] for sid, v in per.items() if sid in last}
        if per and all(len(v) == n for v in per.values()):
            t0_, t1_ = min(v[0]["s"] for v in per.values()), max(v[-1]["e"] for v in per.values())
            together_us = (t1_ - t0_) / 1e3 / n
            out["alone_phases"] = {"launches_per_phase": n, "alone_us_per_step": alone_us, "together_us_per_step": together_us,
                                   "alone_over_together": sum(alone_us.values()) / together_us}
print(json.dumps(out, indent=1))
if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
