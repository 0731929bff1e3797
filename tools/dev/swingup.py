#!/usr/bin/env python3
"""Development experiment (GPU): closed-loop swing-up from the hanging position on the device harness, several configs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cartpolesimulation_amd.engine import MPPIEngine
from cartpolesimulation_amd.configs import MPPIConfig
from cartpolesimulation_amd.harness import BatchedCartPoleExperiment

E = 64
rng = np.random.Generator(np.random.SFC64(1))
s0 = np.zeros((E, 6), np.float32)
ang = np.pi + rng.uniform(-0.2, 0.2, E)
s0[:, 0] = ang; s0[:, 2] = np.cos(ang); s0[:, 3] = np.sin(ang)
import itertools
cases = [dict(num_rollouts=3500, mpc_horizon=35, horizon_reduce=r, LBD=l, cost_function_specification=c)
         for c in ("quadratic_boundary_grad_minimal", "default", "quadratic_boundary_grad") for r in ("sum", "mean") for l in (100.0,)]
cases += [dict(num_rollouts=3500, mpc_horizon=35, horizon_reduce="mean", LBD=1.0), dict(num_rollouts=3500, mpc_horizon=35, horizon_reduce="sum", LBD=1000.0),
          dict(num_rollouts=3500, mpc_horizon=40), dict(num_rollouts=3500, mpc_horizon=45)]
for kw in cases:
    cost, N, H, LBD = kw.get("cost_function_specification", "quadratic_boundary_grad_minimal"), kw["num_rollouts"], kw["mpc_horizon"], (kw.get("horizon_reduce", "sum"), kw.get("LBD", 100.0))
    eng = MPPIEngine(E, MPPIConfig(**kw))
    ex = BatchedCartPoleExperiment(eng, seed=3)
    t0 = time.perf_counter()
    out = ex.run(s0, 500, record=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = out["states"].cpu().numpy()
    up = np.abs(st[:, :, 0]) < 0.2
    first = [int(np.argmax(up[:, e])) if up[:, e].any() else -1 for e in range(E)]
    final_up = up[-50:].all(axis=0).mean()
    print(f"{cost:32s} N={N} H={H} LBD={LBD}: upright for the last second in {100*final_up:.0f} % of {E} envs; "
          f"median first-upright step {np.median([f for f in first if f >= 0]) if any(f >= 0 for f in first) else -1}; "
          f"max|x| {np.abs(st[:, :, 4]).max():.3f}; {dt:.2f} s for 500 control steps")
