#!/usr/bin/env python3
"""Randomised differential test (GPU box): the fused HIP step against the C oracle on many random problem instances.

Beyond the 8 golden regimes: `--batches` x 64 random envs (states over the whole state space, targets, pole lengths,
warm nominal sequences), 1024 rollouts x 50 steps each, the SAME perturbations on both sides.  Reports the distribution
of the relative per-rollout cost deviation and of the control-update deviation, per math mode and lane mapping, as one
JSON object (stored by the round as profiles/<tag>/fuzz_parity.json).  The oracle runs on the host cores (OpenMP).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from oracle import oracle_np as O  # noqa: E402
from oracle import oracle_c as OC  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as PU  # noqa: E402  (the tests' own allowance rules: 1e-4 + the oracle's A/B gap / soft-min conditioning)

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, default=10)
ap.add_argument("--envs", type=int, default=64)
ap.add_argument("--costs", nargs="+", default=["qbgm"], choices=["qbgm", "default", "legacy"],
                help="cost plugins to sweep (quadratic_boundary_grad_minimal, default, the legacy mppi-cartpole cost with its glue)")
args = ap.parse_args()
E, N, H = args.envs, 1024, 50
THL = 0.198
from cartpolesimulation_amd.configs import legacy_mppi_config  # noqa: E402

out = {"workload": f"{args.batches} batches x {E} random envs x {N} rollouts x {H} steps per variant, perturbations from the device sampler",
       "variants": {}}
ENGINE_CFG = {"qbgm": lambda **kw: MPPIConfig(**kw),
              "default": lambda **kw: MPPIConfig(cost_function_specification="default", **kw),
              "legacy": lambda **kw: legacy_mppi_config(**kw)}
ORACLE_CFG = {"qbgm": lambda: O.MPPIConfig(N=N, H=H),
              "default": lambda: O.MPPIConfig(N=N, H=H, cost_id=O.COST_DEFAULT),
              "legacy": lambda: O.MPPIConfig(N=N, H=H, cost_id=O.COST_LEGACY, SQRTRHOINV=0.02, control_mode="penalise",
                                             shift_mode="append_zero", correction_u="u_nom")}
for cost, math, rpl in [(c, m, r) for c in args.costs for m, r in (("fast", 1), ("fast", 2), ("precise", 1))]:
    eng = MPPIEngine(E, ENGINE_CFG[cost](num_rollouts=N, mpc_horizon=H, math_mode=math, rollouts_per_lane=rpl))
    cfg_c = OC.make_config(ORACLE_CFG[cost]())
    cfg_b = OC.make_config(ORACLE_CFG[cost](), mode="f64sub")
    rels, du_max, excess = [], [], []
    rng = np.random.Generator(np.random.SFC64(77))
    for b in range(args.batches):
        ang = rng.uniform(-np.pi, np.pi, E)
        s0 = np.zeros((E, 6), np.float32)
        s0[:, 0], s0[:, 1] = ang, rng.uniform(-12, 12, E)
        s0[:, 2], s0[:, 3] = np.cos(ang), np.sin(ang)
        s0[:, 4], s0[:, 5] = rng.uniform(-0.9, 0.9, E) * THL, rng.uniform(-0.6, 0.6, E)
        tp = (rng.uniform(-0.8, 0.8, E) * THL).astype(np.float32)
        te = np.where(rng.uniform(size=E) < 0.8, 1.0, -1.0).astype(np.float32)
        Lv = rng.uniform(0.2, 0.5, E).astype(np.float32)
        u0 = np.clip(0.3 * rng.standard_normal((E, H)), -1, 1).astype(np.float32)
        _, du = eng.sample(seed=1000 + b, offset=b, knots=False, delta_u=True)
        un = eng.tensor(u0.copy())
        S = eng.empty(E, N)
        eng.step(s0, un, tp, te, L=Lv, delta_u=du, S_out=S)
        u_ref, _, S_ref = OC.step(cfg_c, s0, u0, du.cpu().numpy(), tp, te, L=Lv)
        rel = np.abs(S.cpu().numpy() - S_ref) / np.abs(S_ref)
        rels.append(rel.reshape(-1))
        du_max.append(np.abs(un.cpu().numpy() - u_ref).max(axis=1))
        # the same deviation against what the tests allow: 1e-4 + max(oracle A/B gap, soft-min conditioning bound)
        du_h = du.cpu().numpy()
        u_b, _, S_b = OC.step(cfg_b, s0, u0, du_h, tp, te, L=Lv)
        for e in range(E):
            allow = 1e-4 + max(float(np.abs(u_ref[e] - u_b[e]).max()),
                               float(np.max(PU.softmin_allowance(S_ref[e], S_b[e], du_h[e], LBD=100.0))))
            excess.append(float(np.abs(un.cpu().numpy()[e] - u_ref[e]).max()) / allow)
    r, d = np.concatenate(rels), np.concatenate(du_max)
    eng.close()
    out["variants"][f"{cost}/{math}/rollouts_per_lane={rpl}"] = {
        "rollouts": int(r.size), "envs": int(d.size),
        "cost_rel_dev": {"median": float(np.median(r)), "p90": float(np.percentile(r, 90)), "p99": float(np.percentile(r, 99)),
                         "p999": float(np.percentile(r, 99.9)), "max": float(r.max()), "frac_below_1e-4": float(np.mean(r < 1e-4)),
                         "frac_below_1e-3": float(np.mean(r < 1e-3))},
        "control_update_abs_dev_per_env": {"median": float(np.median(d)), "p90": float(np.percentile(d, 90)), "max": float(d.max()),
                                           "frac_below_1e-4": float(np.mean(d < 1e-4)),
                                           "frac_within_test_allowance": float(np.mean(np.asarray(excess) <= 1.0)),
                                           "max_over_test_allowance": float(np.max(excess))}}
print(json.dumps(out))
