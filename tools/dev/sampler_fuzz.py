#!/usr/bin/env python3
"""Development tool (GPU): the perturbation path over random shapes — the device interpolation of knots is bit for bit
the oracle's (scipy interp1d as the reference calls it), the tiled layout is a lossless permutation of the reference
layout (tile / untile), the tiled sampler equals tile(sampler), and the fused step gives identical results from the three
representations of the SAME perturbations (knots, delta_u, tiled delta_u) per lane mapping.
  python tools/dev/sampler_fuzz.py --n 100 --seed 1"""
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

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
rng = np.random.Generator(np.random.SFC64(args.seed))
fails = done = 0
for it in range(args.n):
    E = int(rng.integers(1, 5))
    N = int(rng.choice([1, 3, 63, 64, 65, 128, 130, 255, 256, 257, 512, 700]))
    H = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 19, 20, 21, 35, 50, 51, 99, 100]))
    period = int(rng.choice([1, 2, 3, 4, 5, 7, 10, 12]))
    rpl = int(rng.choice([1, 2]))
    desc = dict(E=E, N=N, H=H, period=period, rpl=rpl)
    try:
        eng = MPPIEngine(E, MPPIConfig(num_rollouts=N, mpc_horizon=H, period_interpolation_inducing_points=period, rollouts_per_lane=rpl))
        seed, off = int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1000))
        kn, du = eng.sample(seed=seed, offset=off, knots=True, delta_u=True)
        knh, duh = kn.cpu().numpy(), du.cpu().numpy()
        # two interpolation forms by design: cpmppi_interpolate / the knots noise source use the reference's (float64 slope,
        # scipy interp1d as controller_mppi_cartpole.py:434-446 calls it) - for knots that come from the reference's own
        # stream; the device sampler and the in-kernel Philox path interpolate their OWN knots with one float32 FMA
        du_i = eng.interpolate(kn)
        for e in range(E):
            assert np.array_equal(O.interpolate_knots(knh[e], H, period), du_i.cpu().numpy()[e]), "cpmppi_interpolate != oracle interpolation"
        assert np.abs(duh - du_i.cpu().numpy()).max() <= 1.2e-7 * max(1.0, np.abs(duh).max()), "sampler's delta_u more than an ulp from the exact interpolation"
        tiled = eng.tile_delta_u(du)
        assert np.array_equal(eng.untile(tiled).cpu().numpy(), duh), "untile(tile(du)) != du"
        assert torch.equal(eng.sample_tiled(seed, off), tiled), "sample_tiled != tile(sample)"
        s0 = np.stack([O.create_cartpole_state(rng.uniform(-1, 1), rng.uniform(-2, 2), rng.uniform(-0.15, 0.15), rng.uniform(-0.3, 0.3))
                       for _ in range(E)])
        tp, te = (0.05 * rng.uniform(-1, 1, E)).astype(np.float32), np.ones(E, np.float32)
        u0 = (0.2 * rng.standard_normal((E, H))).astype(np.float32)

        def run(**kw):
            un = eng.tensor(u0.copy())
            S = eng.empty(E, N)
            eng.step(s0, un, tp, te, S_out=S, **kw)
            return S.cpu().numpy(), un.cpu().numpy()

        for base, others in (((dict(delta_u=du), "sampler's delta_u"), ((dict(delta_u_tiled=tiled), "tiled"), (dict(seed=seed, offset=off), "philox"))),
                             ((dict(delta_u=du_i), "interpolated knots"), ((dict(knots=kn), "knots"),))):
            S0, u0n = run(**base[0])
            for kw, name in others:
                S1, u1 = run(**kw)
                assert np.array_equal(S1, S0), f"costs differ between {base[1]} and {name}"
                assert np.abs(u1 - u0n).max() <= 2e-6, f"updated controls differ between {base[1]} and {name}: {np.abs(u1 - u0n).max():.2e}"
        eng.close()
        done += 1
    except AssertionError as ex:
        fails += 1
        print("FAIL", json.dumps(desc), str(ex)[:300], flush=True)
    except Exception as ex:  # noqa: BLE001
        fails += 1
        print("ERROR", json.dumps(desc), type(ex).__name__, str(ex)[:300], flush=True)
print(json.dumps({"configurations": args.n, "passed": done, "failed": fails, "seed": args.seed}))
