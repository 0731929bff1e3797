#!/usr/bin/env python3
"""C3 / C4 at full size: the DISTRIBUTION over the 64 envs of |u - u_A| / max(1e-4, |u_A - u_B|) (how far the kernel's updated
control sequence sits from the reference's float32 result, in units of the reference's own two-mode spread) for several input
seeds, FAST and PRECISE - the evidence behind the bound of 3 asserted in tests/test_gpu_configs.py (task 6).

Development / evidence tool (uses the oracle: test infrastructure).  Usage: python tools/c3_spread.py [--config C3] [--seeds 2 21 22 23] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_np as O  # noqa: E402
from oracle import parity as PU  # noqa: E402
from test_gpu_configs import inputs, make  # noqa: E402

f32 = np.float32


def one(name, E, N, H, seed):
    s0, tp, te, Lv = inputs(E, H, seed=seed)
    rng = np.random.Generator(np.random.SFC64(9))          # (seed 2 = the launch of tests/test_gpu_configs.py::test_config_full_size[C3])
    u0 = (0.1 * rng.standard_normal((E, H))).astype(f32)
    out = {}
    eng = make(E, N, H)
    kn, _ = eng.sample(seed=seed, offset=0)
    kn_h = kn.cpu().numpy()
    res = {}
    S_dev = {}
    for mode in ("fast", "precise"):
        e = make(E, N, H, math_mode=mode)
        un, S = e.tensor(u0.copy()), e.empty(E, N)
        e.step(s0, un, tp, te, L=Lv, knots=kn, S_out=S)
        res[mode], S_dev[mode] = un.cpu().numpy(), S.cpu().numpy()
        e.close()
    clear = {m: 0 for m in res}
    ocfg = O.MPPIConfig(N=N, H=H)
    ratios = {m: [] for m in res}
    ratios_env = {m: [] for m in res}
    spreads, envelopes = [], []
    for e0 in range(0, E, 8):
        sl = slice(e0, e0 + 8)
        du = np.stack([O.interpolate_knots(kn_h[e], H) for e in range(e0, e0 + 8)])
        ref = PU.c_oracle_step_with_flags(ocfg, s0[sl], u0[sl], du, tp[sl], te[sl], L=Lv[sl], probes=True)
        for i, e in enumerate(range(e0, e0 + 8)):
            ua = ref["u_a"][i].astype(np.float64)
            spreads.append(float(np.abs(ua - ref["u_b"][i]).max()))
            # the envelope of ALL the oracle's realisations of the reference on this env: modes A / B and the probes one rounding
            # away from mode A (FMA build, initial state / perturbations / pole length one ulp up)
            env = max([spreads[-1]] + [float(np.abs(ua - a[i]).max()) for a in ref["u_alt"]])
            envelopes.append(env)
            for m in res:
                # the rollouts compared at full strength (rule ODE_V0: clear of the oracle's discontinuity and rounding-sensitivity flags)
                b = PU.cost_buckets(S_dev[m][e], ref["S_a"][i], ref["S_b"][i], ref["flags"][i], [a[i] for a in ref["S_alt"]], flag_sensitive=True)
                clear[m] += int((~b["flagged"]).sum())
                ratios[m].append(PU.reference_spread_ratio(res[m][e], ref["u_a"][i], ref["u_b"][i]))
                ratios_env[m].append(float(np.abs(res[m][e] - ua).max() / max(1e-4, env)))
    for m, r in ratios.items():
        r = np.array(r)
        out[m] = dict(clear_fraction=clear[m] / float(E * N), worst=float(r.max()), worst_env=int(r.argmax()), p90=float(np.percentile(r, 90)), median=float(np.median(r)),
                      over_1=int((r > 1).sum()), over_2=int((r > 2).sum()), over_3=int((r > 3).sum()),
                      ratios=[round(float(x), 3) for x in r])
        q = np.array(ratios_env[m])
        out[m]["vs_envelope"] = dict(worst=float(q.max()), worst_env=int(q.argmax()), p90=float(np.percentile(q, 90)), over_1=int((q > 1).sum()),
                                     over_2=int((q > 2).sum()), over_3=int((q > 3).sum()), ratios=[round(float(x), 3) for x in q])
    out["reference_AB_spread"] = dict(worst=float(np.max(spreads)), median=float(np.median(spreads)), per_env=[float(f"{x:.3e}") for x in spreads])
    out["reference_envelope"] = dict(worst=float(np.max(envelopes)), median=float(np.median(envelopes)), per_env=[float(f"{x:.3e}") for x in envelopes])
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3", choices=["C3", "C4"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[2, 21, 22, 23, 24])
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    E, N, H = (64, 4096, 100) if args.config == "C3" else (64, 2048, 50)
    rec = dict(config=args.config, E=E, N=N, H=H, seeds={})
    for seed in args.seeds:
        rec["seeds"][str(seed)] = r = one(args.config, E, N, H, seed)
        print(f"[{args.config} seed {seed}] " + "  ".join(
            f"{m}: worst {r[m]['worst']:.2f} (env {r[m]['worst_env']}) p90 {r[m]['p90']:.2f} median {r[m]['median']:.2f} >1:{r[m]['over_1']} >2:{r[m]['over_2']} >3:{r[m]['over_3']}"
            for m in ("fast", "precise")) + f"  clear fraction {r['fast']['clear_fraction']:.3f} / {r['precise']['clear_fraction']:.3f}"
            + f"  reference |u_A-u_B| worst {r['reference_AB_spread']['worst']:.2e}" +
            "  | vs the envelope of all oracle realisations: " + "  ".join(f"{m} worst {r[m]['vs_envelope']['worst']:.2f} >3:{r[m]['vs_envelope']['over_3']}" for m in ("fast", "precise")), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
