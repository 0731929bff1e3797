#!/usr/bin/env python3
"""Development tool (GPU): per-launch duration of the rollout kernel over a LONG loop at a small config (HIP events
recorded by the library on the launch stream), to separate clock ramp / DVFS from kernel properties.

  python tools/dev/step_series.py --config C4 --rpl 2 --steps 3000 [--preheat 1.0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--rpl", type=int, default=0)
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--preheat", type=float, default=0.0, help="seconds of the 8192-env headline workload run right before")
ap.add_argument("--frozen", action="store_true", help="reset u_nom to zero before every step (identical work per launch)")
args = ap.parse_args()
E, N, H = {"C2": (8192, 1024, 50), "C3": (64, 4096, 100), "C4": (64, 2048, 50), "C1": (1, 1024, 50)}[args.config]
dev = torch.device("cuda", 0)
eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=args.rpl), device=0)
s0, tp, te, L = synthetic_inputs(E, H, 2, dev)
u_nom = eng.zeros(E, H)
if args.preheat > 0:
    big = MPPIEngine(8192, MPPIConfig(num_rollouts=1024, mpc_horizon=50), device=0)
    bs0, btp, bte, bL = synthetic_inputs(8192, 50, 3, dev)
    bu = big.zeros(8192, 50)
    t0 = time.perf_counter()
    i = 0
    while time.perf_counter() - t0 < args.preheat:
        big.step(bs0, bu, btp, bte, L=bL, seed=1, offset=i)
        i += 1
        if i % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
for i in range(3):
    eng.step(s0, u_nom, tp, te, L=L, seed=1234, offset=i)
torch.cuda.synchronize()
eng.set_profiling(True)
t0 = time.perf_counter()
for i in range(args.steps):
    if args.frozen:
        u_nom.zero_()
    eng.step(s0, u_nom, tp, te, L=L, seed=1234, offset=3 + i)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
r, f = eng.get_profile()
r = np.asarray(r) * 1e3
chunks = [r[i:i + max(1, len(r) // 10)] for i in range(0, len(r), max(1, len(r) // 10))]
print(json.dumps({"config": args.config, "rpl": args.rpl, "steps": args.steps, "preheat_s": args.preheat, "frozen": args.frozen,
                  "wall_us_per_step": wall / args.steps * 1e6,
                  "kernel_us": {"min": float(r.min()), "p10": float(np.percentile(r, 10)), "median": float(np.median(r)),
                                "p90": float(np.percentile(r, 90)), "max": float(r.max()), "mean": float(r.mean())},
                  "first_40": [round(float(x), 1) for x in r[:40]],
                  "chunk_medians": [round(float(np.median(c)), 1) for c in chunks],
                  "rollouts_per_s_at_median": E * N / (float(np.median(r)) * 1e-6)}))
