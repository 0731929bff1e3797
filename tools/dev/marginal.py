#!/usr/bin/env python3
"""Development tool (GPU): marginal cost of a substep and of a control step at a small launch shape.

Times the rollout kernel (HIP events, median of `--steps` launches from the same state, `u_nom` reset each time) for
intermediate_steps S in {4, 10, 16, 22} at fixed horizon and for two horizons at S = 10, and fits
    t = t0 + H * (a + S * b)
so that b = time per substep, a = per-control-step work (noise, cost, interpolation), t0 = launch + prologue + finalize.

  python tools/dev/marginal.py --envs 64 --rollouts 2048 --horizon 50 [--rpl 1]
"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=64)
ap.add_argument("--rollouts", type=int, default=2048)
ap.add_argument("--horizon", type=int, default=50)
ap.add_argument("--rpl", type=int, default=0)
ap.add_argument("--steps", type=int, default=15)
ap.add_argument("--stat", default="min", choices=["min", "median"])
args = ap.parse_args()
dev = torch.device("cuda", 0)
E, N = args.envs, args.rollouts


def timed(H, S):
    # mpc_timestep scales with S so that the substep length (and with it the dynamics per substep) is unchanged
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, intermediate_steps=S, mpc_timestep=0.002 * S,
                                   rollouts_per_lane=args.rpl), device=0)
    s0, tp, te, Lt = synthetic_inputs(E, H, 2, dev)
    u_nom = eng.zeros(E, H)
    ts = []
    for i in range(args.steps + 3):
        u_nom.zero_()
        eng.set_profiling(True)
        eng.step(s0, u_nom, tp, te, L=Lt, seed=99, offset=i)
        torch.cuda.synchronize()
        r, _ = eng.get_profile()
        eng.set_profiling(False)
        if i >= 3:
            ts.append(float(r[0]) * 1e3)
    return float(np.min(ts) if args.stat == "min" else np.median(ts))


H = args.horizon
rows = []
for S in (4, 10, 16, 22):
    rows.append((H, S, timed(H, S)))
rows.append((2 * H, 10, timed(2 * H, 10)))
A = np.array([[1.0, h, h * s] for h, s, _ in rows])
y = np.array([t for _, _, t in rows])
(t0, a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
print(json.dumps({"envs": E, "rollouts": N, "horizon": H, "rpl": args.rpl, "stat": args.stat,
                  "points_us": [(h, s, round(t, 1)) for h, s, t in rows],
                  "t0_us": round(float(t0), 1), "per_step_ns": round(float(a) * 1e3, 1), "per_substep_ns": round(float(b) * 1e3, 1)}))
