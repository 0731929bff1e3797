#!/usr/bin/env python3
"""Development tool (GPU): how far the HIP path sits from the reference's golden rollouts, bucketed as SURVEY.md H2 asks
(rollouts clear of the edge-bounce / angle-wrap / cost-indicator discontinuities vs the rest).  Used to set the fixed
tolerances of tests/test_gpu_parity.py — run it, read the numbers, never derive a test bound from the kernel's error."""
import json
import os
import sys

import numpy as np
from numpy.random import SFC64, Generator

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_np as O  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402

f32 = np.float32
g = np.load(os.path.join(ROOT, "tests", "golden", "rollouts_c2.npz"))
N, H = int(g["N"]), int(g["H"])
THL = float(O.DEFAULT_PARAMS.TrackHalfLength)


def regen(seed, stdev):
    rng = Generator(SFC64(int(seed)))
    for _ in range(5):
        rng.uniform(-1.0, 1.0)
    kn = O.sample_knots(rng, N, H, np.float64(stdev))
    return kn, O.interpolate_knots(kn, H)


for name in ["upright", "hanging", "near_edge", "fast", "random0", "random1", "random2", "random3"]:
    kn, du = regen(g[f"{name}/seed"], g["stdev"])
    s0, u_nom, target = g[f"{name}/s0"], g[f"{name}/u_nom"], float(g[f"{name}/target"])
    u_run = (u_nom + du).astype(f32)
    A = O.predict_core(s0, u_run)
    B = O.predict_core(s0, u_run, mode="f64sub")
    gap = np.abs(A[:, -1] - B[:, -1])
    near_edge = (np.abs(A[:, :, O.POSITION_IDX]) > THL - 2e-3).any(axis=1)
    near_wrap = (np.abs(np.abs(A[:, :, O.ANGLE_IDX]) - np.pi) < 2e-3).any(axis=1)
    clear = ~(near_edge | near_wrap)
    band = 1e-4 + 1e-4 * np.abs(A[:, -1])
    for math in ("precise", "fast"):
        eng = MPPIEngine(1, MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode=math, shift_mode="none",
                                       control_mode="penalise", correction_u="u_nom"))
        traj = eng.predict(s0, u_run).cpu().numpy()
        d = np.abs(traj[:, -1] - A[:, -1])
        inband = (d <= band).all(axis=1)
        inband_gap = (d <= band + gap).all(axis=1)
        inband_2gap = (d <= band + 2 * gap).all(axis=1)
        excess = (d / band).max(axis=1)
        rec = {"regime": name, "math": math, "clear": int(clear.sum()), "bucketed": int((~clear).sum()),
               "clear_fail_band": int((~inband & clear).sum()), "clear_fail_band_plus_gap": int((~inband_gap & clear).sum()),
               "clear_fail_band_plus_2gap": int((~inband_2gap & clear).sum()),
               "clear_max_excess_bands": round(float(excess[clear].max()), 3) if clear.any() else None,
               "clear_gap_max_bands": round(float((gap / band).max(axis=1)[clear].max()), 3) if clear.any() else None,
               "bucketed_fail_band_plus_gap": int((~inband_gap & ~clear).sum())}
        # fused costs on both lane mappings
        for rpl in ((1, 2) if math == "fast" else (1,)):
            e2 = MPPIEngine(1, MPPIConfig(num_rollouts=N, mpc_horizon=H, math_mode=math, shift_mode="none", control_mode="clip",
                                          correction_u="u_nom", cc_weight=0.0, rollouts_per_lane=rpl))
            un = e2.tensor(u_nom[None].copy())
            S = e2.empty(1, N)
            e2.step(s0[None], un, target, 1.0, delta_u=du[None], S_out=S)
            S = S.cpu().numpy()[0]
            S_ref = g[f"{name}/clip/S_qbgm"]
            rel = np.abs(S - S_ref) / np.abs(S_ref)
            # the reference's own ambiguity on costs: mode B trajectories through the oracle's cost
            uc = np.clip(u_run, -1, 1)
            SB = O.trajectory_cost(O.COST_QBGM, O.predict_core(s0, uc, mode="f64sub"), uc, f32(target), f32(1.0))
            SA = O.trajectory_cost(O.COST_QBGM, O.predict_core(s0, uc), uc, f32(target), f32(1.0))
            relgap = np.abs(SA - SB) / np.abs(SA)
            rec[f"cost_rpl{rpl}"] = {"median": float(np.median(rel)), "p99": float(np.percentile(rel, 99)), "max_clear": float(rel[clear].max()),
                                     "max_all": float(rel.max()), "clear_over_1e-4": int((rel[clear] > 1e-4).sum()),
                                     "clear_over_1e-4_plus_gap": int((rel[clear] > 1e-4 + relgap[clear]).sum()),
                                     "all_over_1e-4_plus_gap": int((rel > 1e-4 + relgap).sum()),
                                     "oracle_AB_relgap_max_clear": float(relgap[clear].max()),
                                     "u_new_maxdiff": float(np.abs(un.cpu().numpy()[0] - np.clip(u_nom + O.reward_weighted_average(S_ref, du), -1, 1)).max())}
        print(json.dumps(rec))
