#!/usr/bin/env python3
"""Development tool (GPU): where a small launch of the rollout kernel spends its wall time — per-wave time stamps
(s_memrealtime, 100 MHz, one clock for the whole chip) at kernel entry, end of the horizon loop, partials written and
exit, from a -DCPMPPI_DEBUG_COUNTERS build, next to the HIP-event duration of the same launch.

  python __graft_entry__.py --variant dbg -DCPMPPI_DEBUG_COUNTERS=1
  CPMPPI_LIB=build_variants/dbg.so python tools/dev/timeline.py --config C4 --steps 12

Per launch (microseconds after the first wave's entry):
  entry_last   the last wave to start (dispatch ramp)
  loop p50/p99/max  end of the horizon loop
  partials_max last block's partial sums written
  exit_max     last wave leaves (after the fused finalize of the env's last block)
  event_us     HIP events around the launch (what bench.py reports as kernel time)
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from cartpolesimulation_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--rpl", type=int, default=0)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--reader", default=None, help="cpmppi_debug_read | cpmppi_debug_read_mid | cpmppi_debug_read_latency (default: by size)")
args = ap.parse_args()
E, N, H = {"C1": (1, 1024, 50), "C2": (8192, 1024, 50), "C3": (64, 4096, 100), "C4": (64, 2048, 50)}[args.config]
dev = torch.device("cuda", 0)
eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=args.rpl), device=0)
lib = L.load()
total = E * N
reader = args.reader or ("cpmppi_debug_read_latency" if (args.rpl == 1 or (args.rpl == 0 and total < 131072)) else
                         ("cpmppi_debug_read_mid" if total <= 524288 else "cpmppi_debug_read"))
fn = getattr(lib, reader)
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int]
rpl = 1 if "latency" in reader else 2
n_waves = min(16384, E * ((N + 256 * rpl - 1) // (256 * rpl)) * 4)
s0, tp, te, Lt = synthetic_inputs(E, H, 2, dev)
u_nom = eng.zeros(E, H)
cold = np.zeros(n_waves, np.uint32)
stamps = np.zeros((n_waves, 4), np.uint64)
cyc = np.zeros(n_waves, np.uint64)
for i in range(args.steps):
    eng.set_profiling(True)
    eng.step(s0, u_nom, tp, te, L=Lt, seed=1234, offset=i)
    torch.cuda.synchronize()
    r, _ = eng.get_profile()
    eng.set_profiling(False)
    assert fn(cold.ctypes.data, cyc.ctypes.data, stamps.ctypes.data, n_waves, 1) == 0
    t = stamps.astype(np.float64)
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    life = us[:, 1] - us[:, 0]
    print(json.dumps({"step": i, "reader": reader, "waves": n_waves, "event_us": round(float(r[0]) * 1e3, 1),
                      "entry_last": round(float(us[:, 0].max()), 1),
                      "loop_p50": round(float(np.percentile(us[:, 1], 50)), 1), "loop_p99": round(float(np.percentile(us[:, 1], 99)), 1),
                      "loop_max": round(float(us[:, 1].max()), 1),
                      "life_p50": round(float(np.percentile(life, 50)), 1), "life_max": round(float(life.max()), 1),
                      "partials_max": round(float(us[:, 2].max()), 1), "exit_max": round(float(us[:, 3].max()), 1),
                      "cold_max": int(cold.max()),
                      "us_per_cold_entry": round(float(np.polyfit(cold.astype(np.float64), life, 1)[0]), 3) if cold.max() > 0 else None,
                      "slowest": [(round(float(life[j]), 1), int(cold[j])) for j in np.argsort(-life)[:6]],
                      "life_p90": round(float(np.percentile(life, 90)), 1), "life_cold0_p50": round(float(np.median(life[cold == 0])), 1) if np.any(cold == 0) else None,
                      "waves_cold0": int((cold == 0).sum()),
                      "shader_clock_ghz": round(float(np.median(cyc.astype(np.float64) / np.maximum(life, 1e-3))) / 1e3, 3)}))
