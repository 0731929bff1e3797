#!/usr/bin/env python3
"""Development tool (CPU): from a `rocprofv3 --kernel-trace` CSV of a bench.py run with the per-step collective
(CPMPPI_BENCH_FORCE_COLLECTIVE=1, one rank), show where the side stream's work — the one-lane waiter, the all-gather
(with one rank RCCL performs it as a device copy kernel; with more ranks it is RCCL's own kernel) and the one-lane post —
sits relative to the rollout kernels on the launch stream.
  python tools/dev/overlap_from_trace.py gpurun_out/.../*_kernel_trace.csv [out.json]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
roll = sorted((r for r in rows if "rollout_cost_kernel" in r["Kernel_Name"]), key=lambda r: r["s"])
wait = sorted((r for r in rows if "wait_published_kernel" in r["Kernel_Name"]), key=lambda r: r["s"])
post = sorted((r for r in rows if "post_gathered_kernel" in r["Kernel_Name"]), key=lambda r: r["s"])
side_stream = {r["Stream_Id"] for r in wait}
gather = sorted((r for r in rows if r["Stream_Id"] in side_stream and "wait_published" not in r["Kernel_Name"]
                 and "post_gathered" not in r["Kernel_Name"]), key=lambda r: r["s"])
n = min(len(roll), len(wait), len(post))
skip = max(0, n - 40)                                     # the last 40 steps (steady state)
rec = []
for i in range(skip, n - 1):
    k, k2, w, p = roll[i], roll[i + 1], wait[i], post[i]
    g = [x for x in gather if w["e"] <= x["s"] <= p["s"]]
    rec.append({
        "step": i,
        "rollout_us": (k["e"] - k["s"]) / 1e3,
        "gap_to_next_rollout_us": (k2["s"] - k["e"]) / 1e3,
        "waiter_ends_after_rollout_end_us": (w["e"] - k["e"]) / 1e3,
        "gather_kernel": g[0]["Kernel_Name"].split("(")[0] if g else None,
        "gather_start_after_next_rollout_start_us": (g[0]["s"] - k2["s"]) / 1e3 if g else None,
        "gather_us": (g[0]["e"] - g[0]["s"]) / 1e3 if g else None,
        "post_end_before_next_rollout_end_us": (k2["e"] - p["e"]) / 1e3,
        "gather_inside_next_rollout": bool(g) and g[0]["s"] >= k2["s"] - 2000 and p["e"] <= k2["e"],
    })
med = lambda key: sorted(x[key] for x in rec if x[key] is not None)[len(rec) // 2]  # noqa: E731
out = {"trace": sys.argv[1], "steps_analysed": len(rec), "launch_stream": sorted({r["Stream_Id"] for r in roll}),
       "side_stream": sorted(side_stream),
       "median": {k: med(k) for k in ("rollout_us", "gap_to_next_rollout_us", "waiter_ends_after_rollout_end_us",
                                      "gather_start_after_next_rollout_start_us", "gather_us", "post_end_before_next_rollout_end_us")},
       "gathers_completed_inside_the_next_rollout_kernel": sum(x["gather_inside_next_rollout"] for x in rec),
       "other_kernels_on_the_launch_stream_between_rollouts": sorted({r["Kernel_Name"].split("(")[0] for r in rows
                                                                       if r["Stream_Id"] in {x["Stream_Id"] for x in roll}
                                                                       and roll[skip]["s"] < r["s"] < roll[n - 1]["s"]
                                                                       and "rollout_cost_kernel" not in r["Kernel_Name"]}),
       "first_steps": rec[:6]}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
