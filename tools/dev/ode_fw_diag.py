#!/usr/bin/env python3
"""Development tool (GPU): which rollouts of a full-width predictor_ODE launch sit outside band + envelope, and what they look like."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_np as O
from oracle import oracle_c as OC
import parity_util as PU
import test_gpu_configs as TC
from cartpolesimulation_amd.engine import MPPIEngine
from cartpolesimulation_amd.configs import MPPIConfig
f32 = np.float32
name, E, N, H, seed = ("C4", 64, 2048, 50, 13) if (len(sys.argv) < 2 or sys.argv[1] == "C4") else ("C3", 64, 4096, 100, 12)
s0, tp, te, Lv = TC.inputs(E, H, seed=seed)
rng = np.random.Generator(np.random.SFC64(9))
u0 = (0.1 * rng.standard_normal((E, H))).astype(f32)
outs = {}
kn = None
for mode in ("fast", "precise"):
    eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, predictor_type="ODE", math_mode=mode))
    if kn is None:
        kn, _ = eng.sample(seed=2, offset=0)
    un, S = eng.tensor(u0.copy()), eng.empty(E, N)
    eng.step(s0, un, tp, te, L=Lv, knots=kn, S_out=S)
    outs[mode] = S.cpu().numpy()
kn_h = kn.cpu().numpy()
ocfg = O.MPPIConfig(N=N, H=H, integrator="ODE")
for e0 in range(0, E, 8):
    sl = slice(e0, e0 + 8)
    du = np.stack([O.interpolate_knots(kn_h[e], H) for e in range(e0, e0 + 8)])
    ref = PU.c_oracle_step_with_flags(ocfg, s0[sl], u0[sl], du, tp[sl], te[sl], L=Lv[sl], probes=True)
    gap = PU.envelope(ref["S_a"], ref["S_b"], *ref["S_alt"])
    for mode in ("fast", "precise"):
        d = np.abs(outs[mode][sl].astype(np.float64) - ref["S_a"])
        off = d > 1e-4 * np.abs(ref["S_a"]) + gap
        sens = gap > 0.25e-4 * np.abs(ref["S_a"])
        for i in range(8):
            n_off = int((off[i] & ~sens[i]).sum())
            if n_off:
                idx = np.nonzero(off[i] & ~sens[i])[0][:3]
                e = e0 + i
                u_shift = np.concatenate([u0[e, 1:], u0[e, -1:]])
                print(f"{name} {mode} env {e}: {n_off} clear rollouts off; s0 = {np.round(s0[e], 3).tolist()} L {Lv[e]:.3f} tp {tp[e]:.3f}")
                for n in idx:
                    ur = np.clip(u_shift + du[i, n], -1, 1).astype(f32)
                    tr = OC.predict(OC.make_config(ocfg), s0[e:e + 1], ur[None], L=Lv[e:e + 1])[0]
                    print(f"    rollout {n}: S {outs[mode][e, n]:.6g} S_a {ref['S_a'][i, n]:.6g} rel {d[i, n] / abs(ref['S_a'][i, n]):.2e} gap/|S| {gap[i, n] / abs(ref['S_a'][i, n]):.2e}"
                          f" max|x| {np.abs(tr[:, 4]).max():.3f} max|w| {np.abs(tr[:, 1]).max():.1f} max|v| {np.abs(tr[:, 5]).max():.2f}")
