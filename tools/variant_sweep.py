#!/usr/bin/env python3
"""Which build of the rollout kernel should a launch of a given size get?  Development tool (VERDICT r4 task 2): times one MPPI step
(wall clock over K back-to-back steps, one stream) for every (rollouts per lane, build variant) the library can be steered to through
CPMPPI_LONE_FORM_MAX_WAVES / CPMPPI_LATENCY_MAX_ROLLOUTS, over a range of launch sizes.

Since round 6 the two overrides exist in a -DCPMPPI_DEV_KNOBS build only (the shipped library has the constants):
  python __graft_entry__.py --variant devknobs -DCPMPPI_DEV_KNOBS
Usage: CPMPPI_LIB=build_variants/devknobs.so python tools/variant_sweep.py [--shapes 2048x50 4096x100 1024x50] [--envs 16 32 64 ...] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_inputs  # noqa: E402
from cartpolesimulation_amd.configs import MPPIConfig  # noqa: E402
from cartpolesimulation_amd.engine import MPPIEngine  # noqa: E402

BIG = str(2 ** 40)
SETTINGS = {   # name -> (rollouts_per_lane, env overrides)
    "R1 latency (v0)": (1, {"CPMPPI_LATENCY_MAX_ROLLOUTS": BIG}),
    "R1 throughput (v1)": (1, {"CPMPPI_LATENCY_MAX_ROLLOUTS": "0"}),
    "R2 lone form (v3)": (2, {"CPMPPI_LONE_FORM_MAX_WAVES": BIG}),
    "R2 phased / throughput (v2/v1)": (2, {"CPMPPI_LONE_FORM_MAX_WAVES": "0"}),
    "library default": (0, {}),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["2048x50", "4096x100", "1024x50"])
    ap.add_argument("--envs", type=int, nargs="+", default=[16, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024])
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    out = []
    for shape in args.shapes:
        N, H = (int(x) for x in shape.split("x"))
        for E in args.envs:
            if E * N * H > 2 ** 28:
                continue
            s0, tp, te, Lt = synthetic_inputs(E, H, 3, dev)
            row = dict(E=E, N=N, H=H, rollouts=E * N, results={})
            for name, (rpl, env) in SETTINGS.items():
                for k in ("CPMPPI_LATENCY_MAX_ROLLOUTS", "CPMPPI_LONE_FORM_MAX_WAVES"):
                    os.environ.pop(k, None)
                os.environ.update(env)
                eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, rollouts_per_lane=rpl))
                u, Q = eng.zeros(E, H), eng.empty(E)
                prep = eng.prepare_step(s0, u, tp, te, L=Lt, seed=1234, offset=0, Q_out=Q)
                ts = []
                for rnd in range(args.rounds):
                    u.zero_()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(args.steps):
                        prep.run(offset=i)
                    torch.cuda.synchronize()
                    if rnd:
                        ts.append((time.perf_counter() - t0) / args.steps * 1e6)
                info = eng.last_launch()
                row["results"][name] = dict(us=float(np.median(ts)), min_us=float(np.min(ts)), kernel=info["kernel"], blocks=info["blocks"],
                                            checksum=float(u.double().abs().sum()))
                eng.close()
            best = min((v["us"], k) for k, v in row["results"].items() if k != "library default")
            d = row["results"]["library default"]
            print(f"{E:5d} x {N} x {H} ({E * N:8d} rollouts): " + "  ".join(f"{k.split(' (')[0]} {v['us']:7.1f}" for k, v in row["results"].items())
                  + f"   | best {best[1]} ({best[0]:.1f} us); default runs {d['kernel']} = {d['us'] / best[0]:.2f} x best", flush=True)
            out.append(row)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
