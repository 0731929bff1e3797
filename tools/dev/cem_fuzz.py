#!/usr/bin/env python3
"""Development tool (GPU): the CEM kernels over random shapes — the top-k refit (stable, ties by index) against the numpy
oracle on random and heavily tied costs, populations from 1 to 16384 (not powers of two), any K <= N; the cost-only launch
against the fused step's costs for the same plans.   python tools/dev/cem_fuzz.py --n 100 --seed 1"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from oracle import oracle_np as O  # noqa: E402

f32 = np.float32
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.Generator(np.random.SFC64(args.seed))
fails = done = 0
for it in range(args.n):
    E = int(rng.integers(1, 5))
    N = int(rng.choice([1, 2, 3, 7, 31, 64, 65, 100, 200, 255, 256, 257, 1000, 1024, 1025, 4097, 8192, 9000, 16384]))
    H = int(rng.choice([1, 2, 5, 17, 35, 64]))
    K = int(rng.integers(1, min(N, 300) + 1))
    kind = str(rng.choice(["random", "tied", "constant", "with_inf"]))
    desc = dict(E=E, N=N, H=H, K=K, costs=kind)
    try:
        eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, shift_mode="none"))
        Q = eng.tensor(rng.uniform(-1, 1, (E, N, H)).astype(f32))
        if kind == "random":
            S = rng.uniform(0, 1000, (E, N)).astype(f32)
        elif kind == "tied":
            S = rng.integers(0, 5, (E, N)).astype(f32)
        elif kind == "constant":
            S = np.full((E, N), 3.5, f32)
        else:
            S = rng.uniform(0, 1000, (E, N)).astype(f32)
            S[rng.uniform(size=(E, N)) < 0.2] = np.inf
        St = eng.tensor(S)
        m, sd, el = eng.cem_update(St, Q, K, 0.01, return_elites=True)
        Qh = Q.cpu().numpy()
        for e in range(E):
            mr, sr, idx = O.cem_update(S[e], Qh[e], K, 0.01)
            assert np.array_equal(el.cpu().numpy()[e], idx), f"env {e}: elite set / order differs"
            assert np.abs(m.cpu().numpy()[e] - mr).max() <= 2e-6 and np.abs(sd.cpu().numpy()[e] - sr).max() <= 4e-6, f"env {e}: mean / stdev"
        if N <= 1100 and it % 3 == 0:
            # cost-only launch == the fused step's per-rollout costs for the same plans (u_nom = 0, delta_u = plans, no shift)
            s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.1, 0.1), 0.0) for _ in range(E)])
            tp, te = np.zeros(E, f32), np.ones(E, f32)
            Sc = eng.rollout_cost(s0, Q, tp, te).cpu().numpy()
            Sf = eng.empty(E, N)
            eng.step(s0, eng.zeros(E, H), tp, te, delta_u=Q, S_out=Sf)
            # (the fused step adds the MPPI correction term cc (0.5 (1 - 1/NU) R du^2 + R u du + 0.5 R u^2) with u = u_run; with
            # u_nom = 0 and plans inside the limits u = du = the plan: (0.5 (1 - 1/NU) + 1.5) plan^2 summed over the horizon)
            corr = ((0.5 * (1 - 1 / 1000.0) + 1.5) * (Qh.astype(np.float64) ** 2)).sum(axis=2)
            assert np.abs(Sf.cpu().numpy() - (Sc + corr)).max() <= 2e-4 * np.abs(Sc).max() + 1e-3, "cost-only launch vs fused step"
        eng.close()
        done += 1
    except AssertionError as ex:
        fails += 1
        print("FAIL", json.dumps(desc), str(ex)[:300], flush=True)
    except Exception as ex:  # noqa: BLE001
        fails += 1
        print("ERROR", json.dumps(desc), type(ex).__name__, str(ex)[:300], flush=True)
print(json.dumps({"configurations": args.n, "passed": done, "failed": fails, "seed": args.seed}))
