#!/usr/bin/env python3
"""Development tool (GPU): every env of the C3 / C4 launch against the C oracle (both arithmetic modes), per math mode
and lane mapping: how many rollouts sit outside band + A/B gap, flagged or clear, and what the offenders look like
(the oracle's own trajectory: top angular speed, closest approach to the edge, angle seam crossings, the A/B gap).
  python tools/dev/cfg_parity_diag.py [C3|C4]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_np as O  # noqa: E402
from oracle import oracle_c as OC  # noqa: E402
import parity_util as PU  # noqa: E402
from test_gpu_configs import make, inputs  # noqa: E402

f32 = np.float32
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
E, N, H = {"C3": (64, 4096, 100), "C4": (64, 2048, 50)}[name]
s0, tp, te, Lv = inputs(E, H, seed=2 if name == "C3" else 3)
rng = np.random.Generator(np.random.SFC64(9))
u0 = (0.1 * rng.standard_normal((E, H))).astype(f32)
eng = make(E, N, H)
kn, _ = eng.sample(seed=2, offset=0)
kn_h = kn.cpu().numpy()
outs = {}
for tag, kw in (("fast_r2", dict(rollouts_per_lane=2)), ("fast_r1", dict(rollouts_per_lane=1)), ("precise", dict(math_mode="precise"))):
    e = make(E, N, H, **kw)
    un, S = e.tensor(u0.copy()), e.empty(E, N)
    e.step(s0, un, tp, te, L=Lv, knots=kn, S_out=S)
    outs[tag] = (un.cpu().numpy(), S.cpu().numpy())
    e.close()
ocfg = O.MPPIConfig(N=N, H=H)
THL = PU.THL
SENS = float(os.environ.get("DIAG_SENS", "0.25e-4"))
CH = 8
tot = {t: dict(clear_off=0, flagged_off=0, clear=0, flagged=0, sens_off=0, worst_clear=0.0, u_worst=0.0,
               env_clear=0, env_sens=0, env_clear_off=0, env_sens_off=0, env_worst_clear=0.0, u_off_envs=0, u_off_envs_env=0) for t in outs}
for e0 in range(0, E, CH):
    sl = slice(e0, e0 + CH)
    du = np.stack([O.interpolate_knots(kn_h[e], H) for e in range(e0, e0 + CH)])
    ref = PU.c_oracle_step_with_flags(ocfg, s0[sl], u0[sl], du, tp[sl], te[sl], L=Lv[sl], probes=True)
    # the oracle's trajectories once more for the offenders' features
    u_shift = np.concatenate([u0[sl, 1:], u0[sl, -1:]], axis=1)
    u_run = np.clip(u_shift[:, None, :] + du, -1, 1).astype(f32).reshape(CH * N, H)
    traj = OC.predict(OC.make_config(ocfg), np.repeat(s0[sl], N, axis=0), u_run, L=np.repeat(Lv[sl], N)).reshape(CH, N, H + 1, 6)
    for t, (u_m, S_m) in outs.items():
        for i, e in enumerate(range(e0, e0 + CH)):
            Sa, Sb = ref["S_a"][i].astype(np.float64), ref["S_b"][i].astype(np.float64)
            gap = np.abs(Sa - Sb)
            dev = np.abs(S_m[e] - Sa)
            off = dev > 1e-4 * np.abs(Sa) + gap
            fl = ref["flags"][i]
            sens = gap > 1e-4 * np.abs(Sa)
            T = tot[t]
            T["clear"] += int((~fl).sum()); T["flagged"] += int(fl.sum())
            T["clear_off"] += int((off & ~fl).sum()); T["flagged_off"] += int((off & fl).sum())
            T["sens_off"] += int((off & ~fl & sens).sum())
            exc = dev / (1e-4 * np.abs(Sa) + gap)
            if (~fl).any():
                T["worst_clear"] = max(T["worst_clear"], float(exc[~fl].max()))
            T["u_worst"] = max(T["u_worst"], float(np.abs(u_m[e] - ref["u_a"][i]).max()))
            # rule with the envelope of the reference's own realisations (B, C = fma float32, P = one ulp away)
            genv = PU.envelope(Sa, Sb, *[a[i] for a in ref["S_alt"]])
            sens_e = genv > SENS * np.abs(Sa)
            off_e = dev > 1e-4 * np.abs(Sa) + genv
            bucket = fl | sens_e
            T["env_clear"] += int((~bucket).sum()); T["env_sens"] += int((sens_e & ~fl).sum())
            T["env_clear_off"] += int((off_e & ~bucket).sum()); T["env_sens_off"] += int((off_e & bucket).sum())
            if (~bucket).any():
                T["env_worst_clear"] = max(T["env_worst_clear"], float((dev / (1e-4 * np.abs(Sa) + genv))[~bucket].max()))
            allow = PU.softmin_allowance(Sa, Sb, du[i])
            ugap = float(PU.envelope(ref["u_a"][i], ref["u_b"][i]).max())
            ugap_e = float(PU.envelope(ref["u_a"][i], ref["u_b"][i], *[a[i] for a in ref["u_alt"]]).max())
            du_ = np.abs(u_m[e] - ref["u_a"][i])
            T["u_off_envs"] += int((du_ > 1e-4 + np.maximum(ugap, allow)).any())
            T["u_off_envs_env"] += int((du_ > 1e-4 + np.maximum(ugap_e, allow)).any())
            for n in np.nonzero(off_e & ~bucket)[0][:6]:
                print(json.dumps({"rule": "envelope", "tag": t, "env": e, "rollout": int(n), "rel_dev": float(dev[n] / abs(Sa[n])),
                                  "rel_env": float(genv[n] / abs(Sa[n])), "rel_gapAB": float(gap[n] / abs(Sa[n])), "S_a": float(Sa[n])}))
            for n in np.nonzero(off & ~fl)[0][:6]:
                tr = traj[i, n]
                print(json.dumps({"tag": t, "env": e, "rollout": int(n), "S": float(S_m[e][n]), "S_a": float(Sa[n]), "S_b": float(Sb[n]),
                                  "rel_dev": float(dev[n] / abs(Sa[n])), "rel_gap": float(gap[n] / abs(Sa[n])),
                                  "max_w": float(np.abs(tr[:, 1]).max()), "edge_margin": float(THL - (np.abs(tr[:, 4]) + 0.02 * np.abs(tr[:, 5])).max()),
                                  "min_dist_pi": float(np.abs(np.abs(tr[:, 0]) - np.pi).min()), "s0": s0[e].tolist(), "L": float(Lv[e]),
                                  "other_tags_rel_dev": {t2: float(abs(outs[t2][1][e][n] - Sa[n]) / abs(Sa[n])) for t2 in outs if t2 != t}}))
print(json.dumps({"config": name, "totals": tot}))
